"""numpy restatement of the engine's on-device noise generator.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws xi from
numpy's global legacy RNG (ces/calibrate.py:447, 488, 527); a GPU engine
cannot share that stream, so parity runs inject xi and production runs use the
counter-based generator restated here:

  Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; constants of Random123),
  counter = (global particle index lo, hi, row quad, step), key = seed (lo, hi);
  the four 32-bit outputs feed two Box-Muller pairs that give the noise of rows
  4q..4q+3 of that particle.  fp32 engines use 24-bit uniforms, fp64 engines
  32-bit uniforms, both offset by half a unit so that u is never 0.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def noise_block(p, J, seed, step, j_offset=0, dtype=np.float32):
    """The (p, J) block the engine draws for (seed, step) on the shard that
    starts at global particle ``j_offset``.  Evaluated in float64."""
    nq = (p + 3) // 4
    gj = (np.arange(J, dtype=np.uint64) + np.uint64(j_offset))[None, :].repeat(nq, axis=0)
    q = np.arange(nq, dtype=np.uint32)[:, None].repeat(J, axis=1)
    lo = (gj & MASK).astype(np.uint32)
    hi = (gj >> np.uint64(32)).astype(np.uint32)
    step_arr = np.full_like(lo, np.uint32(step & 0xFFFFFFFF))
    x = philox4x32_10(lo, hi, q, step_arr, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    if np.dtype(dtype) == np.dtype(np.float32):
        u = [((xi >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24 for xi in x]
    else:
        u = [(xi.astype(np.float64) + 0.5) * 2.0 ** -32 for xi in x]
    ra, rb = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    z = np.stack([ra * np.cos(2 * np.pi * u[1]), ra * np.sin(2 * np.pi * u[1]),
                  rb * np.cos(2 * np.pi * u[3]), rb * np.sin(2 * np.pi * u[3])], axis=1)   # (nq, 4, J)
    return z.reshape(nq * 4, J)[:p]
