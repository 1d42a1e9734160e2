#!/usr/bin/env python3
"""dev tool: per step of a rocprofv3 --kernel-trace run of bench.py: when the second Gram launch and the Cholesky start
relative to the end of the first reduce, and how long they take (looks for the HIP-event-sampled step's distortion).
    python tools/trace_steps_summary.py DIR [last_n]"""
import csv, glob, sys
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fn = sorted(glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void ", "").replace("cesx::", "").split("<")[0].split("(")[0]
ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
k3 = [i for i, e in enumerate(ev) if e[0] in ("update2_kernel", "update3_kernel")]
out = []
for a, b in zip(k3[:-1], k3[1:]):
    seg = ev[a + 1:b + 1]
    names = [e[0] for e in seg]
    if names.count("gram2_kernel") != 2 or "potrf_reg_kernel" not in names:
        continue
    g = [e for e in seg if e[0] == "gram2_kernel"]
    red = [e for e in seg if e[0] == "gram_reduce_kernel"]
    po = [e for e in seg if e[0] == "potrf_reg_kernel"][0]
    t0 = red[0][2]
    out.append("gram_a %6.1f | gram_b starts %+6.1f lasts %6.1f | potrf starts %+6.1f lasts %6.1f | K3 starts %+6.1f after gram_b end, lasts %6.1f | step %6.1f"
               % ((g[0][2] - g[0][1]) / 1e3, (g[1][1] - t0) / 1e3, (g[1][2] - g[1][1]) / 1e3, (po[1] - t0) / 1e3, (po[2] - po[1]) / 1e3,
                  (seg[-1][1] - g[1][2]) / 1e3, (seg[-1][2] - seg[-1][1]) / 1e3, (seg[-1][2] - g[0][1]) / 1e3))
print("\n".join(out[-last:]))
