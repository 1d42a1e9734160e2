"""Build ``libcesx.so`` (the C-ABI engine) in-tree with hipcc for gfx950.

    python -m ces_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is written next to this
file so that it travels with the source tree; it is git-ignored.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcesx.so")
SOURCES = ["engine.hip", "kernels_stats.hip", "kernels_gram.hip", "kernels_gram2.hip", "kernels_dense.hip", "kernels_update.hip", "kernels_update2.hip", "kernels_update3.hip", "kernels_update4.hip", "kernels_calib.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "cesx_internal.h"), os.path.join(os.path.dirname(HERE), "include", "cesx.h")]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libcesx.so."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(obj, [spath] + HEADERS):
            jobs.append([hipcc] + FLAGS + ["-c", spath, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), res.stderr))
        return res.stderr

    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
