#!/usr/bin/env python3
"""Print the kernel time line of the last complete step(s) from a rocprofv3 --kernel-trace CSV directory (dev tool).
    python tools/trace_step.py gpurun_out/prof_x [nsteps]"""
import csv, glob, sys
d = sys.argv[1]; nst = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fn = sorted(glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void ", "").replace("cesx::", "").split("<")[0].split("(")[0]
k3 = [i for i, r in enumerate(rows) if "update2_kernel" in r["Kernel_Name"] or "update2s_kernel" in r["Kernel_Name"] or "update3_kernel" in r["Kernel_Name"]]
if not k3:
    k3 = [i for i, r in enumerate(rows) if "update_kernel" in r["Kernel_Name"]]
i0, i1 = k3[-2 - nst] + 1, k3[-2] + 3
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f -> %9.1f  dur %8.1f  %s  [q%s]" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
