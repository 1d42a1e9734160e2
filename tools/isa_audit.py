"""Static audit of the gfx950 ISA of the hot kernels (no GPU needed).

    python tools/isa_audit.py [--write profiles/r05_resource_usage.txt]

Compiles every translation unit of libcesx.so with
``-Rpass-analysis=kernel-resource-usage`` (registers, spills, scratch, occupancy per
kernel) and with ``-S`` (device ISA), then walks the ISA of each kernel: a LOOP is a label
that a later branch jumps back to; inside loops it counts ``scratch_*`` instructions and
``v_readlane`` / ``v_writelane`` against the kernel's SGPR-spill VGPR (the reloads the
compiler puts where it ran out of scalar registers).  ``tests/test_isa_audit.py`` fails
when a hot kernel gains scratch or in-loop spill traffic.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ces_amd import build as B      # noqa: E402

FIELDS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill",
          "VGPRs Spill", "LDS Size [bytes/block]")


def demangle(names):
    if not names:
        return {}
    filt = "c++filt"
    out = subprocess.run([filt] + list(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def remarks(src):
    """{mangled kernel: {field: int}} of one translation unit."""
    cmd = [B._hipcc()] + B.FLAGS + ["-c", os.path.join(B.CSRC, src), "-o", "/dev/null", "--cuda-device-only",
                                    "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    res, cur = {}, None
    for line in err.split("\n"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            v = m.group(2)
            cur[m.group(1).strip()] = int(v) if v.isdigit() else v
    return res


def isa(src):
    cmd = [B._hipcc()] + B.FLAGS + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", "-"]
    return subprocess.run(cmd, capture_output=True, text=True).stdout


def loops_of(body):
    """[(first line, last line)] of the loops of one function body (list of lines)."""
    label_at = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
    spans = []
    for i, ln in enumerate(body):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            spans.append((label_at[m.group(1)], i))
    return spans


def audit_isa(text):
    """{mangled kernel: dict(scratch_total, scratch_in_loop, spill_in_loop, mfma, mfma_in_loop)}"""
    out = {}
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        while j < len(lines) and not lines[j].startswith("\t.end_amdhsa_kernel") and not re.match(r"^\.Lfunc_end", lines[j]):
            j += 1
        body = lines[i:j]
        spans = loops_of(body)
        inloop = [False] * len(body)
        for a, b in spans:
            for k in range(a, b + 1):
                inloop[k] = True
        # the SGPR-spill VGPR: the register v_writelane stores into most often
        wl = {}
        for ln in body:
            mm = re.match(r"^\s+v_writelane_b32 (v\d+), s", ln)
            if mm:
                wl[mm.group(1)] = wl.get(mm.group(1), 0) + 1
        spill_regs = {r for r, c in wl.items() if c >= 2}
        d = dict(scratch_total=0, scratch_in_loop=0, spill_in_loop=0, spill_total=0, mfma=0, mfma_in_loop=0, loops=len(spans))
        for k, ln in enumerate(body):
            s = ln.strip()
            if s.startswith("scratch_"):
                d["scratch_total"] += 1
                d["scratch_in_loop"] += inloop[k]
            mm = re.match(r"v_(readlane|writelane)_b32 (\w+), (\w+),", s)
            if mm:
                reg = mm.group(3) if mm.group(1) == "readlane" else mm.group(2)
                if reg in spill_regs:
                    d["spill_total"] += 1
                    d["spill_in_loop"] += inloop[k]
            if s.startswith("v_mfma"):
                d["mfma"] += 1
                d["mfma_in_loop"] += inloop[k]
        out[name] = d
        i = j
    return out


def collect(sources=None):
    sources = sources or B.SOURCES
    with ThreadPoolExecutor(max_workers=4) as pool:
        rem = list(pool.map(remarks, sources))
        asm = list(pool.map(isa, sources))
    table = {}
    for src, r, a in zip(sources, rem, asm):
        au = audit_isa(a)
        for k, v in r.items():
            row = dict(v)
            row.update(au.get(k, {}))
            row["source"] = src
            table[k] = row
    return table


def render(table):
    names = demangle(sorted(table))
    rows = []
    hdr = ("kernel", "src", "SGPR", "VGPR", "AGPR", "scratch B/lane", "occ", "SGPR spill", "VGPR spill",
           "in-loop spill insns", "in-loop scratch insns", "MFMA in loop / all")
    for k in sorted(table, key=lambda q: (table[q]["source"], names.get(q, q))):
        r = table[k]
        nm = re.sub(r"\(.*", "", names.get(k, k)).replace("cesx::", "").replace("void ", "")
        rows.append((nm, r["source"].replace("kernels_", "").replace(".hip", ""), r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
                     r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"),
                     r.get("VGPRs Spill"), r.get("spill_in_loop"), r.get("scratch_in_loop"),
                     "%s / %s" % (r.get("mfma_in_loop"), r.get("mfma"))))
    w = [max(len(str(x[i])) for x in rows + [hdr]) for i in range(len(hdr))]
    fmt = "  ".join("%%-%ds" % x for x in w)
    return "\n".join([fmt % hdr, fmt % tuple("-" * x for x in w)] + [fmt % tuple(str(c) for c in r) for r in rows])


if __name__ == "__main__":
    t = collect()
    txt = ("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage + a walk of the -S output (tools/isa_audit.py)\n"
           "# in-loop spill insns: v_readlane / v_writelane against the kernel's SGPR-spill VGPR between a loop label and its back edge\n"
           + render(t) + "\n")
    if "--write" in sys.argv:
        path = sys.argv[sys.argv.index("--write") + 1]
        with open(os.path.join(ROOT, path), "w") as f:
            f.write(txt)
    print(txt)
