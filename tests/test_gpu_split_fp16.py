"""Split-fp16 arithmetic (x = fp16 hi + fp16 lo, three exact fp16 products accumulated in fp32 on the fp16 matrix
instruction) is the default way this package evaluates fp32 GEMMs: the NeRF point MLP (csrc/nerf.hip) and the stand-alone
decoder GEMMs (CIPS3D_GEMM_SPLIT).  These tests pin its accuracy claim: against an fp64 evaluation it errs no more than
plain fp32 does (the reference's precision), and it agrees with the fp32-MFMA kernels to ~1e-6 of the value range."""
import math

import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import _lib, configs, hip, weights
from cips_3dplusplus_amd.camera import Camera
from conftest import maxdiff
from oracle import path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(t):
    return t.to(DEV).contiguous()


@pytest.mark.parametrize("cin,cout,hw,B", [(512, 512, 4096, 1), (256, 512, 4096, 2), (512, 256, 1024, 1), (64, 32, 256, 2)])
def test_split_gemm_is_as_accurate_as_fp32(cin, cout, hw, B):
    x = weights.det_normal("sg.x", (B, cin, hw), 1.5, cin)
    x = x * (1.0 + 3.0 * weights.det_unit_uniform("sg.m", (B, cin, 1), cin))       # a spread of activation magnitudes
    W = weights.det_normal("sg.W", (cout, cin), 1.0, cout)
    s = 1.0 + weights.det_uniform("sg.s", (B, cin), 0.5, 3)
    scale = 1.0 / math.sqrt(cin)
    w64 = (scale * W.double())[None] * s.double()[:, None, :]
    w64 = w64 * torch.rsqrt((w64 ** 2).sum(-1, keepdim=True) + 1e-8)
    ref64 = torch.bmm(w64, x.double())
    w32 = (scale * W)[None] * s[:, None, :]
    w32 = w32 * torch.rsqrt((w32 ** 2).sum(-1, keepdim=True) + 1e-8)
    ref32 = torch.bmm(w32, x)
    Wd, sd_, xd = cu(W.view(1, cout, cin, 1, 1)), cu(s), cu(x.view(B, cin, hw, 1))
    out = {}
    for split in (False, True):
        wm = hip.modulate_weights(Wd, sd_, cin, B, cout, cin, 1, scale, True, True, split=split)
        out[split] = hip.modconv1x1(xd, wm, cout, epilogue=0, split=split).view(B, cout, hw).cpu()
    rng = float(ref64.abs().max())
    e_exact, e_split, e_ref32 = (float((t.double() - ref64).abs().max()) for t in (out[False], out[True], ref32))
    print(f"{cin}->{cout}: |fp32 MFMA - fp64| {e_exact:.2e}  |split - fp64| {e_split:.2e}  |torch fp32 - fp64| {e_ref32:.2e}  range {rng:.1f}")
    assert e_split <= 1.5 * max(e_exact, e_ref32) + 1e-7 * rng
    assert maxdiff(out[True], out[False]) < 3e-6 * rng


def test_split_nerf_is_as_accurate_as_fp32():
    """Deep renderer (D = 8, hidden 256): the HIP renderer (split-fp16 MLP) against the fp64 oracle errs no more than the fp32
    oracle (= the reference's arithmetic) does."""
    cfg = configs.ffhq_G_cfg(256, 8)
    G = pkg.build_generator(cfg, DEV, seed=5)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    S, N, B, D = 16, 12, 1, 8
    cam = O.camera_params(torch.tensor([[0.3, 0.1]]), S, 6, 0.12)
    styles = weights.det_normal("sn.styles", (B, D + 1, 256), 0.5, 2)
    thumb, feats, sdf, mask, xyz = G.renderer.render(cu(cam[0]), cu(cam[1]), cu(cam[2]), cu(cam[3]), cu(styles), S, N, return_sdf=True)

    def oracle(dt):
        sdd = {k: v.to(dt) if v.is_floating_point() else v for k, v in sd.items()}
        c = [t.to(dt) for t in cam[:4]]
        ro, rd, vd = O.rays_in_world(c[1], S, c[0], False)
        z = O.z_vals(c[2], c[3], B, S, S, N)
        pts = O.ray_points(ro, rd, z)
        R = S * S
        return O.renderer_forward(sdd, "renderer", pts.reshape(B, R, N, 3), rd.reshape(B, R, 3), vd.reshape(B, R, 3),
                                  z.reshape(B, R, N), c[2], c[3], styles.to(dt), D)
    r32, r64 = oracle(torch.float32), oracle(torch.float64)
    to_img = lambda t: t.reshape(B, S, S, -1).permute(0, 3, 1, 2)       # noqa: E731
    for name, hipv, i in (("features", feats, 1), ("thumb", thumb, 0), ("sdf", sdf.reshape(B, S * S, N, 1), 2)):
        ref64 = r64[i] if name == "sdf" else to_img(r64[i])
        ref32 = r32[i] if name == "sdf" else to_img(r32[i])
        e_hip = float((hipv.cpu().double().reshape(ref64.shape) - ref64).abs().max())
        e_32 = float((ref32.double() - ref64).abs().max())
        print(f"D=8 {name}: |hip - fp64| {e_hip:.2e}   |fp32 oracle - fp64| {e_32:.2e}")
        assert e_hip <= 2.0 * e_32 + 2e-7, name


def test_fp32_exact_mode_agrees_with_the_default():
    cfg = configs.ffhq_G_cfg(256, 2)
    G = pkg.build_generator(cfg, DEV, seed=1)
    zs, nb, _ = weights.synth_inputs(cfg, seed=4)
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=torch.tensor([[0.2, 0.05]], device=DEV))
    kw = dict(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[cu(b) for b in nb],
              nerf_cfg=dict(N_samples=12, perturb=False, static_viewdirs=False))
    a = G(**kw)["rgb"].clone()
    plan = list(G._plans.values())[0].plan
    assert any(plan.layers[i].flags & 2 for i in range(plan.n_dec_layers))          # split-packed layers in the default plan
    G.set_decoder_precision("fp32_exact")
    b = G(**kw)["rgb"].clone()
    plan = list(G._plans.values())[0].plan
    assert not any(plan.layers[i].flags & 2 for i in range(plan.n_dec_layers))
    rng = float(b.abs().max())
    d = maxdiff(a, b)
    print(f"default (split-fp16 GEMMs) vs fp32_exact decoder: max-abs {d:.2e} on range {rng:.2f}")
    assert 0 < d < 2e-5 * rng
    G.set_decoder_precision("fp32")
    assert torch.equal(G(**kw)["rgb"], a)
