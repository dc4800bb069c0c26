"""Deterministic synthetic weights for a generator `state_dict`.

No checkpoint is obtainable offline, so benchmarks and full-size parity fixtures run on
random-init weights.  The values are keyed by PARAMETER NAME (not by construction order), so
the same tensors can be loaded into the reference generator (tests/golden/make_golden.py,
this container only) and into this package's generator on the GPU box.

Distributions follow the reference constructors (SURVEY.md Appendix E;
/root/reference/exp/cips3d/volume_renderer.py:16-29,56-67 and models/model_v3.py:48-52,188-191,
250-254): they set the range of the SIREN sine arguments and of the decoder activations.
Deviations, made so every term of the path is numerically live: NoiseInjection.weight = 0.1
(init 0), activation / toRGB biases ~ U(-0.1, 0.1) (init 0).
"""
import math
import zlib

import torch


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _uniform(shape, bound, g):
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).mul_(bound).float()


def _normal(shape, std, g):
    return torch.randn(shape, generator=g, dtype=torch.float64).mul_(std).float()


def synth_tensor(name, shape, seed=0, lr_mul_mapping=0.01):
    shape = tuple(shape)
    g = _gen(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("sigmoid_beta"):
        return torch.full(shape, 0.1)
    if name.endswith("blur.kernel") or name.endswith("upsample.kernel"):
        k = torch.tensor([1.0, 3.0, 3.0, 1.0])
        k = k[None, :] * k[:, None]
        return k / k.sum() * 4.0
    if ".gamma." in name or ".beta." in name:                      # FiLM style heads (LinearLayer)
        if leaf == "weight":
            return _normal(shape, 0.25 * math.sqrt(2.0 / (1.04 * shape[1])), g)
        return _uniform(shape, math.sqrt(1.0 / shape[0]) if len(shape) == 1 else 1.0, g)
    if name.startswith("renderer.network.") or name.startswith("network."):
        if leaf == "weight":
            if ".pts_linears.0." in name:
                return _uniform(shape, 1.0 / 3.0, g)
            return _uniform(shape, math.sqrt(6.0 / shape[1]) / 25.0, g)
        # bias: U(+-sqrt(1/in)); `in` is not recoverable from a 1-D shape for the heads, so use
        # the layer width (hidden) for FiLM layers and a fixed small range for the two heads.
        return _uniform(shape, 1.0 / 16.0, g)
    if name.startswith("style_decoder."):
        if leaf == "weight":
            return _normal(shape, 1.0 / lr_mul_mapping, g)
        return _uniform(shape, 0.1 / lr_mul_mapping, g)             # runtime bias = bias * lr_mul
    if name.startswith("style."):
        if leaf == "weight":
            return _normal(shape, math.sqrt(2.0 / (1.04 * shape[1])), g)
        return _uniform(shape, math.sqrt(1.0 / shape[0]), g)
    if ".modulation." in name:
        if leaf == "weight":
            return _normal(shape, 1.0, g)
        return torch.ones(shape) + _uniform(shape, 0.05, g)
    if name.endswith("conv.weight"):
        return _normal(shape, 1.0, g)
    if name.endswith("noise.weight"):
        return torch.full(shape, 0.1)
    if name.endswith("activate.bias") or leaf == "bias":
        return _uniform(shape, 0.1, g)
    raise KeyError(f"no synthetic rule for parameter {name!r} {shape}")


def synth_state_dict(shapes, seed=0, lr_mul_mapping=0.01):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    return {k: synth_tensor(k, s, seed, lr_mul_mapping) for k, s in shapes.items()}


def state_dict_checksum(sd):
    """Order-independent fp64 checksum used by fixtures to detect RNG drift between machines."""
    tot = 0.0
    for k in sorted(sd):
        t = sd[k].double().flatten()
        w = torch.arange(1, t.numel() + 1, dtype=torch.float64) % 97 + 1
        tot += float((t * w).sum()) * ((zlib.crc32(k.encode()) % 89) + 1)
    return tot
