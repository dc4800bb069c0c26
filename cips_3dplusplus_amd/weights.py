"""Deterministic synthetic weights (and test inputs) for a generator `state_dict` -- closed form, no RNG.

No checkpoint is obtainable offline, so benchmarks and full-size parity fixtures run on
random-init-like weights.  The values are keyed by PARAMETER NAME (not by construction order), so
the same tensors can be loaded into the reference generator (tests/golden/make_golden.py,
this container only) and into this package's generator on the GPU box.

Every element is a pure function of (name, seed, element index): a 64-bit integer mix (splitmix64
finaliser) of the counter, turned into a uniform (top 24 bits) or a bell-shaped variate (sum of the
four 16-bit fields, unit variance) by exactly rounded IEEE double operations.  No library random
number generator and no transcendental function is involved, so the tensors are bit-identical on
every machine, numpy / torch version and thread count: the golden fixtures
assert the checksum instead of skipping on a mismatch.

Distributions follow the reference constructors (SURVEY.md Appendix E;
/root/reference/exp/cips3d/volume_renderer.py:16-29,56-67 and models/model_v3.py:48-52,188-191,
250-254): they set the range of the SIREN sine arguments and of the decoder activations.
Deviations, made so every term of the path is numerically live: NoiseInjection.weight = 0.1
(init 0), activation / toRGB biases ~ U(-0.1, 0.1) (init 0); "normal" draws are the 4-fold
Irwin-Hall bell (variance 1, support +-3.46) instead of a Gaussian.
"""
import math
import zlib

import numpy as np
import torch

_M64 = (1 << 64) - 1
_IH4_STD = math.sqrt((65536.0 ** 2 - 1.0) / 3.0)      # std of the sum of four uniform 16-bit integers


def _mix(name, seed, n):
    """n 64-bit hashes of the counters 0..n-1 under the key (name, seed): splitmix64 finaliser, wrapping uint64."""
    key = ((zlib.crc32(name.encode()) << 32) ^ (int(seed) * 0x9E3779B97F4A7C15)) & _M64
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(key)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def det_uniform(name, shape, bound=1.0, seed=0):
    """U(-bound, bound) on a 2^24 grid (cell centres), float32; closed form in (name, seed, index)."""
    shape = tuple(shape)
    k = (_mix(name, seed, _numel(shape)) >> np.uint64(40)).astype(np.float64)         # 24 bits
    v = ((k + 0.5) * (2.0 / 16777216.0) - 1.0) * float(bound)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def det_normal(name, shape, std=1.0, seed=0):
    """Zero-mean bell with standard deviation `std` (sum of four 16-bit uniforms), float32; closed form."""
    shape = tuple(shape)
    z = _mix(name, seed, _numel(shape))
    m = np.uint64(0xFFFF)
    s = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))).astype(np.float64)
    v = (s - 131070.0) / _IH4_STD * float(std)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def det_unit_uniform(name, shape, seed=0):
    """U[0, 1) on a 2^24 grid, float32 (the per-ray jitter of `get_z_vals`)."""
    shape = tuple(shape)
    k = (_mix(name, seed, _numel(shape)) >> np.uint64(40)).astype(np.float64)
    return torch.from_numpy((k / 16777216.0).astype(np.float32).reshape(shape))


def synth_tensor(name, shape, seed=0, lr_mul_mapping=0.01):
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    uni = lambda bound: det_uniform(name, shape, bound, seed)      # noqa: E731
    nrm = lambda std: det_normal(name, shape, std, seed)           # noqa: E731
    if name.endswith("sigmoid_beta"):
        return torch.full(shape, 0.1)
    if name.endswith("blur.kernel") or name.endswith("upsample.kernel"):
        k = torch.tensor([1.0, 3.0, 3.0, 1.0])
        k = k[None, :] * k[:, None]
        return k / k.sum() * 4.0
    if ".gamma." in name or ".beta." in name:                      # FiLM style heads (LinearLayer)
        if leaf == "weight":
            return nrm(0.25 * math.sqrt(2.0 / (1.04 * shape[1])))
        return uni(math.sqrt(1.0 / shape[0]) if len(shape) == 1 else 1.0)
    if name.startswith("renderer.network.") or name.startswith("network."):
        if leaf == "weight":
            if ".pts_linears.0." in name:
                return uni(1.0 / 3.0)
            return uni(math.sqrt(6.0 / shape[1]) / 25.0)
        # bias: U(+-sqrt(1/in)); `in` is not recoverable from a 1-D shape for the heads, so use
        # the layer width (hidden) for FiLM layers and a fixed small range for the two heads.
        return uni(1.0 / 16.0)
    if name.startswith("style_decoder."):
        if leaf == "weight":
            return nrm(1.0 / lr_mul_mapping)
        return uni(0.1 / lr_mul_mapping)                            # runtime bias = bias * lr_mul
    if name.startswith("style."):
        if leaf == "weight":
            return nrm(math.sqrt(2.0 / (1.04 * shape[1])))
        return uni(math.sqrt(1.0 / shape[0]))
    if ".modulation." in name:
        if leaf == "weight":
            return nrm(1.0)
        return torch.ones(shape) + uni(0.05)
    if name.endswith("conv.weight"):
        return nrm(1.0)
    if name.endswith("noise.weight"):
        return torch.full(shape, 0.1)
    if name.endswith("activate.bias") or leaf == "bias":
        return uni(0.1)
    raise KeyError(f"no synthetic rule for parameter {name!r} {shape}")


def synth_state_dict(shapes, seed=0, lr_mul_mapping=0.01):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    return {k: synth_tensor(k, s, seed, lr_mul_mapping) for k, s in shapes.items()}


def state_dict_checksum(sd):
    """Exact integer checksum of a float32 state dict (wrapping 64-bit sum of position-weighted bit patterns, keyed
    by parameter name): equal on every machine iff the tensors are bit-identical."""
    tot = 0
    for k in sorted(sd):
        bits = sd[k].detach().cpu().contiguous().float().numpy().reshape(-1).view(np.uint32).astype(np.uint64)
        with np.errstate(over="ignore"):
            w = (np.arange(bits.size, dtype=np.uint64) % np.uint64(8191)) + np.uint64(1)
            part = int((bits * w).sum(dtype=np.uint64))
        tot = (tot + part * ((zlib.crc32(k.encode()) % 65521) + 1)) & _M64
    return tot


def synth_inputs(cfg, batch=1, seed=12345, img_size=64):
    """Closed-form latent codes, decoder noise buffers and truncation means for a G_cfg (the inputs of the full-size
    fixtures and of the bench's deterministic mode): (zs, noise_bufs, (mean_r, mean_d))."""
    zdim = cfg["mapping_renderer_cfg"]["z_dim"]
    zs = [det_normal("input.z_render", (batch, zdim), 1.0, seed), det_normal("input.z_decoder", (batch, zdim), 1.0, seed)]
    up = set(cfg["decoder_cfg"].get("upsample_list", []))
    sizes, res, size = [img_size], img_size, 2 ** (int(math.log2(cfg["decoder_cfg"]["size_start"])) + 1)
    while size <= cfg["decoder_cfg"]["size_end"]:     # two StyledConvs per nominal size; resolution doubles at `up` sizes
        if size in up:
            res *= 2
        sizes += [res, res]
        size *= 2
    nb = [det_normal(f"input.noise{i}", (1, 1, s, s), 1.0, seed) for i, s in enumerate(sizes)]
    means = (det_normal("input.mean_render", (1, cfg["mapping_renderer_cfg"]["style_dim"]), 0.2, seed),
             det_normal("input.mean_decoder", (1, cfg["mapping_decoder_cfg"]["style_dim"]), 0.2, seed))
    return zs, nb, means


def synth_inversion_inputs(cfg, res, seed=5):
    """Closed-form leaves and targets of one flip-inversion step (BASELINE config 5; tests/golden/config5.npz):
    (locations [2,2] = image + mirrored view, w_render [2,D+1,S], w_decoder [2,n_latent,S'], noise_bufs, target rgb,
    target thumb)."""
    D = cfg["renderer_cfg"]["N_layers_renderer"]
    dc = cfg["decoder_cfg"]
    n_latent = (int(math.log2(dc["size_end"])) - int(math.log2(dc["size_start"]))) * 2 + 2
    locs = torch.tensor([[0.8, 0.1], [-0.8, 0.1]])
    w_r = det_normal("c5.w_render", (2, D + 1, cfg["mapping_renderer_cfg"]["style_dim"]), 0.5, seed)
    w_d = det_normal("c5.w_decoder", (2, n_latent, cfg["mapping_decoder_cfg"]["style_dim"]), 0.5, seed)
    _, nb, _ = synth_inputs(cfg, batch=1, seed=seed)
    t_rgb = det_uniform("c5.t_rgb", (2, 3, res, res), 1.0, seed)
    t_thumb = det_uniform("c5.t_thumb", (2, 3, 64, 64), 1.0, seed)
    return locs, w_r, w_d, nb, t_rgb, t_thumb
