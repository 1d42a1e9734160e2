#!/usr/bin/env python3
"""dev tool: bench.py's extra.variants block alone (engine-only legs at C2).   python tools/variants_only.py [key ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ces_amd import engine
keys = set(sys.argv[1:])
if keys:
    bench.VARIANTS = [kv for kv in bench.VARIANTS if kv[0] in keys or kv[0] == "aldi_default"]
out = bench.variants_leg(engine, 256, 256, 65536, "float32", 0)
for k, v in out.items():
    if isinstance(v, dict) and k != "in_a_run":
        print("%-28s %s  %s" % (k, v.get("ms_per_step"), v.get("ratio_to_default", v.get("error", ""))))
print("in a run:", json.dumps(out.get("in_a_run"), indent=1))
