#!/usr/bin/env python3
"""Condense rocprofv3 output directories (copied back under gpurun_out/) into the small summaries
that are committed under profiles/.

    python tools/prof_summary.py TAG gpurun_out/prof_k [gpurun_out/prof_FETCH_SIZE ...]

  * <prof_k>   : rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
                 -> profiles/<TAG>_kernel_stats.csv (the tool's own stats table) and
                    profiles/<TAG>_kernel_trace_step.txt (time line of the last complete step)
  * <prof_XXX> : rocprofv3 --pmc <one counter> --output-format csv -- python3 bench.py ...
                 (one pass per counter, MI355X_MICROARCH.md) -> per-kernel mean of the counter per
                 launch in profiles/<TAG>_pmc_summary.json, and profiles/traffic.json =
                 2 * FETCH_SIZE + WRITE_SIZE in bytes per launch (gfx950 FETCH_SIZE tallies the
                 128-byte requests of wide streaming reads as 64 bytes; both counters are in KB).
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "").replace("cesx::", "")
    return name.split("<")[0].split("(")[0]


def main():
    tag, kdir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(ROOT, "profiles")
    stats = sorted(glob.glob(os.path.join(kdir, "*", "*_kernel_stats.csv")))
    if stats:
        shutil.copy(stats[-1], os.path.join(out, tag + "_kernel_stats.csv"))
    trace = sorted(glob.glob(os.path.join(kdir, "*", "*_kernel_trace.csv")))
    if trace:
        rows = list(csv.DictReader(open(trace[-1])))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        # a step ends with metric_final_kernel; show the last complete one plus the next step's first launch
        last = [i for i, r in enumerate(rows) if "metric_final_kernel" in r["Kernel_Name"]]
        i0, i1 = last[-3] + 1, min(last[-2] + 1, len(rows) - 1)
        t0 = int(rows[i0]["Start_Timestamp"])
        with open(os.path.join(out, tag + "_kernel_trace_step.txt"), "w") as f:
            f.write("# one step of `python3 bench.py` under rocprofv3 --kernel-trace (us from the step's first launch;\n"
                    "# the profiler widens the gaps between launches: unprofiled step time is in the bench JSON)\n")
            for r in rows[i0:i1 + 1]:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                f.write("%9.1f -> %9.1f  dur %8.1f  %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"])))
    summary = {}
    for d in pmc_dirs:
        for fn in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            acc = {}
            for r in csv.DictReader(open(fn)):
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                disp = r.get("Dispatch_Id", "")
                acc.setdefault(k, {}).setdefault(disp, 0.0)
                acc[k][disp] += float(r["Counter_Value"])
            for (kern, ctr), per in acc.items():
                vals = sorted(per.values())
                summary.setdefault(kern, {})[ctr] = dict(mean_per_launch=sum(vals) / len(vals), launches=len(vals),
                                                         median_per_launch=vals[len(vals) // 2])
    if summary:
        json.dump(summary, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
        traffic = {"_note": "HBM bytes per launch at C2 = (2*FETCH_SIZE + WRITE_SIZE) KB from separate rocprofv3 --pmc "
                            "passes (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md); "
                            "gram_kernel = sum of a step's two launches; see profiles/%s_pmc_summary.json" % tag}
        for kern, c in summary.items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c and kern in ("gram_kernel", "update2_kernel", "update_kernel"):
                per = (2 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024
                traffic[kern] = int(per * (2 if kern == "gram_kernel" else 1))
        if "update2_kernel" in traffic:
            traffic["update_kernel"] = traffic["update2_kernel"]
        json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
