#!/usr/bin/env python3
"""Condense rocprofv3 output directories (copied back under gpurun_out/) into the small summaries
that are committed under profiles/.

    python tools/prof_summary.py TAG[:CONFIG] gpurun_out/prof_k [gpurun_out/prof_FETCH_SIZE ...]

CONFIG (default C2) is the key under which the HBM bytes go into profiles/traffic.json (bench.py --config).

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


def kernel_sources_sha():
    """sha1 (16 hex digits) of the kernel sources the profiled library was built from: bench.py compares it with the
    tree it runs in and refuses HBM counters of another build (profiles/traffic.json `_src_sha16`)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "ces_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def traffic_per_step(summary):
    """HBM bytes per step and kernel from the per-counter summaries.  Every --pmc pass is a run of its own and the
    benchmark's pre-warm is time-based, so the passes hold DIFFERENT numbers of steps: each counter is normalised by
    the steps of ITS pass (= the launches of the one-per-step update kernel in that pass)."""
    steps = {}
    for k3 in ("update4_kernel", "update2_kernel", "update3_kernel", "metric_final_kernel"):      # one launch per step
        if k3 in summary and "FETCH_SIZE" in summary[k3] and "WRITE_SIZE" in summary[k3]:
            steps = {c: summary[k3][c]["launches"] for c in ("FETCH_SIZE", "WRITE_SIZE")}
            break
    per_step = {}
    for kern, c in summary.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and steps:
            f = c["FETCH_SIZE"]["mean_per_launch"] * c["FETCH_SIZE"]["launches"] / steps["FETCH_SIZE"]
            w = c["WRITE_SIZE"]["mean_per_launch"] * c["WRITE_SIZE"]["launches"] / steps["WRITE_SIZE"]
            per_step[kern] = int((2 * f + w) * 1024)
    return per_step


def mfma_util_per_kernel(summary):
    """MFMA utilisation of each kernel from the counter passes: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's SIMDs) /
    (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024 SIMDs), totals over all launches of the kernel in their passes
    normalised per step like the traffic (a step's two Gram launches count together)."""
    steps = {}
    for k3 in ("update4_kernel", "update2_kernel", "update3_kernel", "metric_final_kernel"):
        if k3 in summary and "SQ_VALU_MFMA_BUSY_CYCLES" in summary[k3] and "GRBM_GUI_ACTIVE" in summary[k3]:
            steps = {c: summary[k3][c]["launches"] for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")}
            break
    out = {}
    for kern, c in summary.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and steps:
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] * c["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"] / steps["SQ_VALU_MFMA_BUSY_CYCLES"]
            act = c["GRBM_GUI_ACTIVE"]["mean_per_launch"] * c["GRBM_GUI_ACTIVE"]["launches"] / steps["GRBM_GUI_ACTIVE"]
            if busy > 0 and act > 0:
                out[kern] = round(busy / (act / 8.0 * 1024.0), 4)
    return out


def main():
    if sys.argv[1] == "--retraffic":          # python tools/prof_summary.py --retraffic TAG[:CONFIG]  (from profiles/TAG_pmc_summary.json)
        tag, _, config = sys.argv[2].partition(":")
        config = config or "C2"
        out = os.path.join(ROOT, "profiles")
        summary = json.load(open(os.path.join(out, tag + "_pmc_summary.json")))
        tpath = os.path.join(out, "traffic.json")
        traffic = json.load(open(tpath))
        per_step = traffic_per_step(summary)
        per_step["gram_kernel"] = per_step.get("gram2_kernel", 0) + per_step.get("gram_kernel", 0)
        per_step["_source"] = tag + "_pmc_summary.json"
        per_step["_src_sha16"] = summary.get("_src_sha16") or kernel_sources_sha()
        mu = mfma_util_per_kernel(summary)
        if "gram2_kernel" in mu:
            mu["gram_kernel"] = mu["gram2_kernel"]
        per_step["_mfma_util"] = mu
        traffic[config] = per_step
        json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
        print({k: v for k, v in per_step.items() if isinstance(v, int) and v > 1e6})
        return
    tag, kdir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    tag, _, config = tag.partition(":")
    config = config or "C2"
    out = os.path.join(ROOT, "profiles")
    stats = sorted(glob.glob(os.path.join(kdir, "*", "*_kernel_stats.csv")))
    if stats:
        shutil.copy(stats[-1], os.path.join(out, tag + "_kernel_stats.csv"))
    trace = sorted(glob.glob(os.path.join(kdir, "*", "*_kernel_trace.csv")))
    if trace:
        rows = list(csv.DictReader(open(trace[-1])))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        # a step ends with its update kernel (the metric finalisation of a single-device run rides on the next step's
        # U x U reduce launch); show the last complete one plus the next step's first launches
        last = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("update4_kernel", "update2_kernel", "update3_kernel"))]
        if len(last) < 4:
            last = [i for i, r in enumerate(rows) if "metric_final_kernel" in r["Kernel_Name"]]
        i0, i1 = last[-4] + 1, min(last[-3] + 2, len(rows) - 1)
        t0 = int(rows[i0]["Start_Timestamp"])
        with open(os.path.join(out, tag + "_kernel_trace_step.txt"), "w") as f:
            f.write("# one step of `python3 bench.py` under rocprofv3 --kernel-trace (us from the step's first launch;\n"
                    "# the profiler widens the gaps between launches: unprofiled step time is in the bench JSON)\n")
            for r in rows[i0:i1 + 1]:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                f.write("%9.1f -> %9.1f  dur %8.1f  %-26s queue %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3,
                                                                          short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
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
        summary = {k: v for k, v in summary.items() if not k.startswith("__amd") and "at::" not in k and "elementwise" not in k}
        json.dump(summary, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
        tpath = os.path.join(out, "traffic.json")
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = {}
        if "gram_kernel" in traffic:                       # round-1 flat layout -> keyed by config
            traffic = {"C2_r01": {k: v for k, v in traffic.items() if not k.startswith("_")}}
        traffic["_note"] = ("HBM bytes per STEP of each kernel = (2*FETCH_SIZE + WRITE_SIZE) KB summed over the kernel's launches "
                            "of a step, from separate rocprofv3 --pmc passes (gfx950 FETCH_SIZE tallies the 128-B requests of "
                            "wide streaming reads as 64 B, MI355X_MICROARCH.md section HBM); keyed by bench.py --config; "
                            "gram_kernel = both Gram launches of a step (gram2_kernel = the LDS-DMA form)")
        per_step = traffic_per_step(summary)
        if per_step:
            per_step["gram_kernel"] = per_step.get("gram2_kernel", 0) + per_step.get("gram_kernel", 0)
            # (update_kernel also counts the forward-map launches of the bench set-up: the step's K3 is
            #  update2_kernel (fp32) / update3_kernel (fp64) when the LDS-DMA kernels ran)
            per_step["_source"] = tag + "_pmc_summary.json"
            per_step["_src_sha16"] = kernel_sources_sha()      # (run right behind the profile passes, on the tree they ran from)
            mu = mfma_util_per_kernel(summary)
            if "gram2_kernel" in mu:
                mu["gram_kernel"] = mu["gram2_kernel"]
            per_step["_mfma_util"] = mu
            traffic[config] = per_step
        json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
