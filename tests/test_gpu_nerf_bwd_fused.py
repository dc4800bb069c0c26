"""The fused NeRF backward (csrc/nerf_bwd_fused.hip: forward recompute with stash + register-resident MFMA backward) against
the materialised sequence (csrc/nerf_bwd.hip + the decoder GEMM, fp32 MFMA data gradients), which test_gpu_backward.py pins to
the oracle's autograd and to the reference's gradients (tests/golden/backward.npz, config5.npz).  Both are fp32-accurate
evaluations of the same sums in a different order: they agree to ~1e-5 of each gradient's max-abs."""
import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import _lib, autograd as AG, configs, hip, weights
from cips_3dplusplus_amd.camera import Camera

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(hidden, depth, B, S, seed=3):
    cfg = configs.tiny_G_cfg(hidden, depth, 1) if hidden < 256 else configs.ffhq_G_cfg(256, depth)
    G = pkg.build_generator(cfg, DEV, seed=seed)
    r = G.renderer
    locs = torch.tensor([[0.25, 0.1], [-0.4, -0.05], [0.1, 0.2], [0.0, 0.0]])[:B].to(DEV)
    cam, focal, near, far = Camera.generate_camera_params(locations=locs, img_size=S, device=DEV, fov_ang=6,
                                                          dist_radius=0.12)[:4]
    styles = (0.5 * weights.det_normal("nbf.styles", (B, depth + 1, r.style_dim), 1.0, seed)).to(DEV)
    film = AG.film_table(r, styles).detach()
    return r, cam, focal, near, far, film


@pytest.mark.parametrize("hidden,depth,B,S,N,static,perturb,scale", [
    (32, 2, 2, 8, 6, False, True, 1.0),        # the tiny generators of the backward goldens
    (32, 1, 1, 8, 5, True, False, 1.0),        # no hidden MFMA layer: the sigma head hangs on layer 0
    (64, 3, 3, 12, 7, False, True, 1e-6),      # 144 rays = 9 groups, odd chunking, gradients at the scale of a mean loss
    (128, 2, 1, 16, 8, True, True, 1e3),
    (256, 6, 2, 64, 24, False, True, 1e-5),    # BASELINE config 5 (CompCars 256^2, D=6, batch 2)
    (256, 9, 1, 16, 4, False, False, 1.0),     # deep network: the two-tile slab steps
])
def test_fused_backward_agrees_with_the_materialised_sequence(hidden, depth, B, S, N, static, perturb, scale):
    assert hip.nerf_backward_fused_supported(hidden, depth, S, N)
    r, cam, focal, near, far, film = _setup(hidden, depth, B, S)
    H = r.hidden_dim
    u = weights.det_unit_uniform("nbf.u", (B, S, S, 1), 2).to(DEV) if perturb else None
    dF = (scale * weights.det_normal("nbf.dF", (B, H, S, S), 1.0, 3)).to(DEV)
    dT = (10.0 * scale * weights.det_normal("nbf.dT", (B, 3, S, S), 1.0, 4)).to(DEV)
    packed, layer_bias = r._derived_buffers()
    args = (r.network, r.sigmoid_beta.detach(), cam, focal, near, far, u, film, layer_bias)
    f0, c0 = hip.nerf_backward(*args, S, N, static, dF, dT)
    f1, c1 = hip.nerf_backward_fused(*args, packed, r._packed_transposed(), S, N, static, dF, dT)
    assert torch.isfinite(f1).all() and torch.isfinite(c1).all()
    for l in range(depth + 1):
        for k, nm in ((0, "gamma"), (1, "beta")):
            ref = f0[:, l, k]
            d = float((ref - f1[:, l, k]).abs().max() / ref.abs().max())
            assert d < 3e-5, (l, nm, d)
    d = float((c0 - c1).abs().max() / c0.abs().max())
    assert d < 3e-5, ("dcam", d)


def test_fused_backward_is_linear_in_the_upstream_gradient():
    """The per-point power-of-two operand scaling must not leak into the result: scaling both upstream gradients by 2^k scales
    every output by exactly 2^k, bit for bit (all scales are powers of two and no intermediate leaves fp16's range)."""
    r, cam, focal, near, far, film = _setup(64, 2, 1, 8)
    S, N, H = 8, 6, 64
    dF = (1e-3 * weights.det_normal("nbf.lin.dF", (1, H, S, S), 1.0, 3)).to(DEV)
    dT = (1e-3 * weights.det_normal("nbf.lin.dT", (1, 3, S, S), 1.0, 4)).to(DEV)
    packed, layer_bias = r._derived_buffers()
    args = (r.network, r.sigmoid_beta.detach(), cam, focal, near, far, None, film, layer_bias)
    run = lambda k: hip.nerf_backward_fused(*args, packed, r._packed_transposed(), S, N, False, dF * k, dT * k)
    f1, c1 = run(1.0)
    for k in (2.0 ** 10, 2.0 ** -12):
        fk, ck = run(k)
        # the sums are atomics in run-dependent order: exactness holds up to that reordering
        assert float((fk / k - f1).abs().max() / f1.abs().max()) < 2e-6
        assert float((ck / k - c1).abs().max() / c1.abs().max()) < 2e-6


def test_unsupported_shapes_fall_back():
    assert not hip.nerf_backward_fused_supported(48, 2, 8, 6)      # hidden width the MFMA tiles do not cover
    assert not hip.nerf_backward_fused_supported(32, 2, 6, 6)      # 36 rays: not whole 16-ray groups
    lib = _lib.load()
    assert lib.cips3d_nerf_bwd_fused_stash_floats(2, 64, 24, 256, 6, 4) == 2 * 1024 * 6 * 6 * 16 * 256


@pytest.mark.parametrize("hidden,depth,B,S,N", [(32, 2, 2, 8, 6), (256, 6, 2, 64, 24), (256, 9, 1, 16, 4)])
def test_stash_filled_by_the_forward_gives_the_same_gradients(hidden, depth, B, S, N):
    """Differentiable forward: cips3d_nerf_render writes the accumulator stash and the per-point sdf / rgb logits itself
    (cips3d_nerf_params.stash / bwd_sdf / bwd_crgb), the backward only rebuilds g from the view layer's stash.  Same gradients
    as when the backward recomputes the forward (the stash holds the same accumulators bit for bit; g is summed in a different
    order), and the forward's own outputs do not change."""
    r, cam, focal, near, far, film = _setup(hidden, depth, B, S)
    H = r.hidden_dim
    u = weights.det_unit_uniform("nbf.u2", (B, S, S, 1), 2).to(DEV)
    dF = (1e-4 * weights.det_normal("nbf.dF2", (B, H, S, S), 1.0, 3)).to(DEV)
    dT = (1e-3 * weights.det_normal("nbf.dT2", (B, 3, S, S), 1.0, 4)).to(DEV)
    packed, layer_bias = r._derived_buffers()
    args = (r.network, r.sigmoid_beta.detach(), cam, focal, near, far, u, film, layer_bias)
    plain = r.render(cam, focal, near, far, None, S, N, perturb_u=u, film=film)
    fwd = hip.nerf_forward_stash(B, S, N, H, depth, DEV)
    kept = r.render(cam, focal, near, far, None, S, N, perturb_u=u, film=film, stash=fwd)
    for a, b in zip(plain, kept):
        if a is not None:
            assert torch.equal(a, b)
    f0, c0 = hip.nerf_backward_fused(*args, packed, r._packed_transposed(), S, N, False, dF, dT)
    f1, c1 = hip.nerf_backward_fused(*args, packed, r._packed_transposed(), S, N, False, dF, dT, fwd=fwd)
    assert float((f0 - f1).abs().max() / f0.abs().max()) < 1e-5
    assert float((c0 - c1).abs().max() / c0.abs().max()) < 1e-5
