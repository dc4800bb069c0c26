"""cips_3dplusplus_amd: MI355X (gfx950) implementation of the CIPS-3D++ generator-forward hot path.

Python mirrors of the reference interface (Generator, Decoder, VolumeFeatureRenderer, Camera,
fused_leaky_relu, upfirdn2d, ...) that launch hand-written HIP kernels through the C ABI of
libcips3d_hip.so (include/cips3d_hip.h).  See DESIGN.md and INTEGRATION.md.
"""
from . import configs, weights  # pure-python helpers, importable without the HIP library

__all__ = ["configs", "weights", "Generator", "Decoder", "VolumeFeatureRenderer", "Camera",
           "fused_leaky_relu", "FusedLeakyReLU", "upfirdn2d", "build_generator"]

_LAZY = {
    "Generator": ("generator", "Generator"),
    "Decoder": ("decoder", "Decoder"),
    "VolumeFeatureRenderer": ("renderer", "VolumeFeatureRenderer"),
    "Camera": ("camera", "Camera"),
    "fused_leaky_relu": ("op", "fused_leaky_relu"),
    "FusedLeakyReLU": ("op", "FusedLeakyReLU"),
    "upfirdn2d": ("op", "upfirdn2d"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f".{mod}", __name__), attr)
    raise AttributeError(name)


def build_generator(G_cfg, device="cuda", state_dict=None, seed=0):
    """`build_model(G_cfg)` of the reference harness (tl2 registry lookup, test_cips3dpp.py:706):
    construct, load either a checkpoint `state_dict` or deterministic synthetic weights, move to device.  The handle is
    an inference / inversion model: `eval()` with every parameter frozen (`requires_grad_(False)`); the inversion loop
    re-enables the decoder's (`G.decoder.requires_grad_(True)`, projector_v10.py:117-123)."""
    from .generator import Generator
    cfg = {k: v for k, v in G_cfg.items() if k not in ("register_modules", "name")}
    G = Generator(**cfg).eval().requires_grad_(False)
    if state_dict is None:
        shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        state_dict = weights.synth_state_dict(shapes, seed=seed,
                                              lr_mul_mapping=cfg["mapping_decoder_cfg"]["lr_mul_mapping"])
    G.load_state_dict(state_dict, strict=True)
    return G.to(device)
