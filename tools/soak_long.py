#!/usr/bin/env python3
"""dev tool: long chained runs (pipelined twice + step by step) must be bit-identical -- race screen for the
deferred publication, the bound hand-over events and the two-buffer noise lookahead."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util
spec = importlib.util.spec_from_file_location("soak", os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak_determinism.py"))
src = open(spec.origin).read().split("bad = 0\nbad += chain_screen")[0]
ns = {"__name__": "soak_defs", "__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
bad = 0
bad += ns["chain_screen"](256, 256, 65536, 12000)
bad += ns["chain_screen"](64, 50, 8192, 20000)
bad += ns["chain_screen"](512, 512, 32768, 600, "float64")
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
