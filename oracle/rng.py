"""TEST INFRASTRUCTURE (oracle): CPU restatement of the noise generator of csrc/rng.hip.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference draws its fresh NoiseInjection noise and ray jitter from torch's global CUDA generator
(models/model_v3.py:334-336 `image.new_empty(batch, 1, height, width).normal_()`, nerf_utils.py:110 `torch.rand`); a
particular stream is not part of its contract (nothing in the reference pins one), the distributions are.  The HIP path
draws them from Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- the generator behind
curand / torch's own CUDA streams) keyed by torch's seed and offset.  This file restates that generator in numpy: the
integer stream is pinned by the published known-answer vectors of the Random123 distribution (kat_vectors: philox4x32 10),
and the device kernel is compared with it bit for bit (integers) / to rounding (Box-Muller in fp32 with hardware log2 / sin).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
STREAM = 0x43495053       # third counter word of every draw of this package ("CIPS")


def philox4x32_10(ctr, key):
    """ctr: uint32 [n, 4], key: uint32 [2] or [n, 2] -> uint32 [n, 4]."""
    c = np.array(ctr, dtype=np.uint32, copy=True).reshape(-1, 4)
    k = np.broadcast_to(np.array(key, dtype=np.uint32).reshape(-1, 2), (c.shape[0], 2)).copy()
    for _ in range(10):
        p0 = M0 * c[:, 0].astype(np.uint64)
        p1 = M1 * c[:, 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        c = np.stack([hi1 ^ c[:, 1] ^ k[:, 0], lo1, hi0 ^ c[:, 3] ^ k[:, 1], lo0], axis=1)
        with np.errstate(over="ignore"):
            k = np.stack([k[:, 0] + W0, k[:, 1] + W1], axis=1).astype(np.uint32)
    return c


def raw_words(seed, base, n_threads):
    """The four 32-bit words of threads 0 .. n_threads-1: counter (lo, hi of base + t, STREAM, 0), key (lo, hi of seed)."""
    idx = np.uint64(base) + np.arange(n_threads, dtype=np.uint64)
    ctr = np.stack([(idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                    np.full(n_threads, STREAM, np.uint32), np.zeros(n_threads, np.uint32)], axis=1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key)


def fill(seed, base, n_normal, n_uniform):
    """(normal float64 [n_normal], uniform float32 [n_uniform]) of one cips3d_rng_fill call: thread t < ceil(n_normal / 4) makes
    normals 4t .. 4t+3 (Box-Muller pairs (w0, w1), (w2, w3): r = sqrt(-2 ln u), u = ((w >> 8) + 0.5) 2^-24 evaluated in fp32;
    angle (w' >> 8) 2^-24 revolutions; sine first), the following ceil(n_uniform / 4) threads make uniforms (w >> 8) 2^-24."""
    qn, qu = (n_normal + 3) // 4, (n_uniform + 3) // 4
    w = raw_words(seed, base, qn + qu)
    wn, wu = w[:qn], w[qn:]
    u = ((wn[:, [0, 2]] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    t = (wn[:, [1, 3]] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    r = np.sqrt(-2.0 * np.log(u.astype(np.float64)))
    ang = 2.0 * np.pi * t.astype(np.float64)
    normal = np.stack([r[:, 0] * np.sin(ang[:, 0]), r[:, 0] * np.cos(ang[:, 0]), r[:, 1] * np.sin(ang[:, 1]),
                       r[:, 1] * np.cos(ang[:, 1])], axis=1).reshape(-1)[:n_normal]
    uniform = ((wu >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(-1)[:n_uniform]
    return normal, uniform
