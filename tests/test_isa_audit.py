"""The hot kernels' register budget, checked on the ISA hipcc emits for gfx950 (no GPU needed).

Round 4's Gram kernel kept its row sums in scratch (a scratch_load / s_waitcnt vmcnt(0) / scratch_store round trip
per piece pair and tile) and reloaded 10 - 74 spilled SGPRs through v_readlane inside its tile loop; nobody had looked.
tools/isa_audit.py compiles the translation units with -Rpass-analysis=kernel-resource-usage and walks the -S output;
this test fails when a hot kernel gains scratch, spills, or spill traffic inside a loop.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def table():
    import isa_audit
    t = isa_audit.collect(["kernels_gram2.hip", "kernels_update2.hip", "kernels_update3.hip", "kernels_update4.hip", "kernels_dense.hip"])
    names = isa_audit.demangle(sorted(t))
    return {re.sub(r"\(.*", "", names[k]).replace("cesx::", "").replace("void ", ""): v for k, v in t.items()}


def test_gram_kernel_has_no_scratch_and_no_spills(table):
    rows = {k: v for k, v in table.items() if k.startswith("gram2_kernel<")}
    assert len(rows) == 8                                  # {float, double} x {scalar, per-lane DMA addressing} x {60-, 64-KiB slots}
    for name, r in rows.items():
        assert r["ScratchSize [bytes/lane]"] == 0, name
        assert r["SGPRs Spill"] == 0 and r["VGPRs Spill"] == 0, name
        assert r["scratch_total"] == 0 and r["spill_in_loop"] == 0, name
        assert r["Occupancy [waves/SIMD]"] == 4, name      # 16 waves per workgroup, one workgroup per CU
        assert r["mfma_in_loop"] == r["mfma"] > 0, name


def test_update_kernels_keep_their_loops_free_of_scratch(table):
    for name in ("update2_kernel<false, true>", "update2_kernel<false, false>", "update3_kernel"):
        r = table[name]
        assert r["scratch_in_loop"] == 0 and r["VGPRs Spill"] == 0, name
        assert r["Occupancy [waves/SIMD]"] == 2, name      # two workgroups of four waves per CU
    # the benchmark's instantiation (every K segment from memory, hk applied at run time): the few SGPR reloads its
    # k-tile loop still has (round 5 count; the Philox instantiations, which nothing on the hot path launches, have 36 - 40)
    assert table["update2_kernel<false, true>"]["spill_in_loop"] <= 6
    assert table["update2_kernel<false, true>"]["ScratchSize [bytes/lane]"] == 0
    # fp64 (the drop-in class's default dtype; C5): round 5 left it with a nominal 36-byte frame and 15 spilled SGPRs -- the epilogue's
    # arguments, held across the K loop; round 6 re-reads them from the kernarg segment behind the loop
    r3 = table["update3_kernel"]
    assert r3["ScratchSize [bytes/lane]"] == 0 and r3["SGPRs Spill"] == 0 and r3["VGPRs Spill"] == 0 and r3["scratch_total"] == 0


def test_the_tail_launch_of_the_benchmark_keeps_its_registers(table):
    """tail_aldi_kernel<false> sits between the second reduce and K3 on the benchmark's critical path (12 - 16 us, latency
    bound).  Round 5's first dense-Sigma version of it shared the instantiation: 157 VGPRs, 98 spilled SGPRs, occupancy 3 --
    and the step lost 4 us before anybody looked.  The dense path is its own instantiation now."""
    for name in ("tail_aldi_kernel<false, false>", "tail_aldi_kernel<false, true>"):      # (..., true: the chained image of update4_kernel)
        r = table[name]
        assert r["Occupancy [waves/SIMD]"] == 4 and r["SGPRs Spill"] <= 4 and r["ScratchSize [bytes/lane]"] == 0, name


def test_the_chained_update_kernel_keeps_ten_blocks_in_registers(table):
    """update4_kernel (round 6: K3 through the Cholesky factor) holds 10 accumulator blocks (160 VGPRs) + fragments per wave at
    two workgroups per CU; every accumulator index must fold to a constant in its unrolled tiles -- an array that does not
    ends up in scratch.  18 triangular tiles + one G tile in the loop, 64 MFMAs each."""
    r = table["update4_kernel"]
    assert r["ScratchSize [bytes/lane]"] == 0 and r["SGPRs Spill"] == 0 and r["VGPRs Spill"] == 0
    assert r["Occupancy [waves/SIMD]"] == 2 and r["scratch_total"] == 0 and r["spill_in_loop"] == 0
    assert r["mfma"] == 19 * 64 and r["mfma_in_loop"] == 64


def test_the_factorisation_that_writes_the_chained_image_costs_no_more_spills(table):
    """potrf_reg_kernel<17, 1 / 2> store every panel twice (L, and transposed and scaled by -1 / Sigma; 2: without the fp64
    factor): instantiations of their own, so that the plain one keeps its registers; none may grow past the round-5 counts."""
    for name in ("potrf_reg_kernel<17, 0>", "potrf_reg_kernel<17, 1>", "potrf_reg_kernel<17, 2>"):
        r = table[name]
        # (1 / 2: waves 4..7 write the images from LDS one panel behind -- a second code path in the panel loop; 1 is a problem's first step only)
        lim = {"0>": (52, 71), "1>": (62, 98), "2>": (55, 84)}[name[-2:]]
        assert r["ScratchSize [bytes/lane]"] == 0 and r["SGPRs Spill"] <= lim[0] and r["spill_in_loop"] <= lim[1], name


def test_small_update_kernels_have_no_scratch_and_no_spills(table):
    """update2s_kernel (fp32, four instantiations) / update3s_kernel (fp64): one accumulator block (two in fp64) per wave and,
    in fp64, all 24 A-fragment pairs of the row block in registers -- 96 of its 140 VGPRs, which is why this is pinned."""
    rows = {k: v for k, v in table.items() if k.startswith("update2s_kernel<") or k == "update3s_kernel"}
    assert len(rows) == 5
    for name, r in rows.items():
        assert r["ScratchSize [bytes/lane]"] == 0 and r["SGPRs Spill"] == 0 and r["VGPRs Spill"] == 0, name
        assert r["mfma"] > 0, name
