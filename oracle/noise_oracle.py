"""CPU restatement of the in-kernel tie-break noise (include/mal_hip.h MAL_STEP_NOISE_PHILOX).

TEST INFRASTRUCTURE ONLY (see oracle/mal_oracle.py).  The reference draws ``torch.randn`` on the host
(manydepth/loss_utils.py:105-106); the HIP step can instead draw N(0,1) inside its first kernel.  The generator is the
published Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; also what
torch's device generator uses) followed by Box-Muller.  Pinned by the known-answer vectors of the Random123
distribution (``kat_vectors``), checked in tests/test_noise_oracle.py.
"""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ctr: (...,4) uint32 array-like, key: (2,) -> (...,4) uint32"""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0, p1 = c[0] * np.uint64(M0), c[2] * np.uint64(M1)
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack(c, -1).astype(np.uint32)


def tiebreak_noise(seed, step, B, H, W):
    """(B,1,H,W) float32: pixel (b,y,x) takes value y&3 of the group keyed by the pixel index of row y&~3
    (mal_march.h tie_noise4); counter = (group, step lo, step hi, 'MAL'), key = seed."""
    b, y, x = np.meshgrid(np.arange(B), np.arange(H), np.arange(W), indexing="ij")
    group = (b * H * W + (y & ~3) * W + x).astype(np.uint32)
    ctr = np.stack([group, np.full_like(group, step & MASK), np.full_like(group, (step >> 32) & MASK),
                    np.full_like(group, 0x4D414C)], -1)
    o = philox4x32_10(ctr, (seed & MASK, (seed >> 32) & MASK)).astype(np.float64)
    out = np.empty((B, H, W, 4))
    for h in range(2):
        u1 = (o[..., 2 * h] + 1.0) * 2.0 ** -32
        u2 = o[..., 2 * h + 1] * 2.0 ** -32
        r = np.sqrt(-2.0 * np.log(u1))
        out[..., 2 * h] = r * np.cos(2 * np.pi * u2)
        out[..., 2 * h + 1] = r * np.sin(2 * np.pi * u2)
    sel = np.take_along_axis(out, (y & 3)[..., None], -1)[..., 0]
    return sel[:, None].astype(np.float32)
