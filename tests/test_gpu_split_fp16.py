"""Split-fp16 arithmetic (x = fp16 hi + fp16 lo, three exact fp16 products accumulated in fp32 on the fp16 matrix
instruction) is the default way this package evaluates fp32 GEMMs: the NeRF point MLP (csrc/nerf.hip) and the stand-alone
decoder GEMMs (CIPS3D_GEMM_SPLIT).  These tests pin its accuracy claim: against an fp64 evaluation it errs no more than
plain fp32 does (the reference's precision), and it agrees with the fp32-MFMA kernels to ~1e-6 of the value range."""
import math

import os

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


@pytest.mark.parametrize("D,N,B", [(2, 24, 1), (8, 13, 2)])
def test_exact_fp32_render_kernel_and_split_kernel_bracket_fp64_equally(D, N, B):
    """VolumeFeatureRenderer.set_precision("fp32_exact") (csrc/nerf.hip's F32 instantiation on v_mfma_f32_16x16x4_f32: IEEE fp32 products, the
    reference's F.linear arithmetic, cips3d/volume_renderer.py:15-35, 74-85) and the default split-fp16 kernel against the fp64
    oracle: both err like the fp32 oracle does; they differ from each other at the fp32 noise level; odd sample counts and
    batches take the padded / multi-group paths; switching back restores the default bit for bit."""
    cfg = configs.ffhq_G_cfg(256, D)
    G = pkg.build_generator(cfg, DEV, seed=5)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    S = 64
    cam = O.camera_params(torch.tensor([[0.3, 0.1], [-0.4, 0.05]][:B]), S, 6, 0.12)
    styles = weights.det_normal("sn.styles", (B, D + 1, 256), 0.5, 2)
    u = weights.det_unit_uniform("sn.u", (B, S * S), 3)
    args = (cu(cam[0]), cu(cam[1]), cu(cam[2]), cu(cam[3]), cu(styles), S, N)
    out = {}
    for prec in ("fp32", "fp32_exact", "fp32"):
        G.renderer.set_precision(prec)
        r = G.renderer.render(*args, perturb_u=cu(u), return_sdf=True)
        if prec in out:
            for a, b in zip(out[prec], r):
                assert torch.equal(a, b)              # back in the default precision: the same bits as before
        out[prec] = [t.clone() for t in r]

    def oracle(dt):
        sdd = {k: v.to(dt) if v.is_floating_point() else v for k, v in sd.items()}
        c = [t.to(dt) for t in cam[:4]]
        ro, rd, vd = O.rays_in_world(c[1], S, c[0], False)
        z = O.z_vals(c[2], c[3], B, S, S, N, perturb_u=u.reshape(B, S, S, 1).to(dt))
        pts = O.ray_points(ro, rd, z)
        R = S * S
        return O.renderer_forward(sdd, "renderer", pts.reshape(B, R, N, 3), rd.reshape(B, R, 3), vd.reshape(B, R, 3),
                                  z.reshape(B, R, N), c[2], c[3], styles.to(dt), D)
    r32, r64 = oracle(torch.float32), oracle(torch.float64)
    to_img = lambda t: t.reshape(B, S, S, -1).permute(0, 3, 1, 2)       # noqa: E731
    for name, idx, i in (("features", 1, 1), ("thumb", 0, 0), ("sdf", 2, 2)):
        ref64 = r64[i] if name == "sdf" else to_img(r64[i])
        ref32 = r32[i] if name == "sdf" else to_img(r32[i])
        e_32 = float((ref32.double() - ref64).abs().max())
        e_split = float((out["fp32"][idx].cpu().double().reshape(ref64.shape) - ref64).abs().max())
        e_exact = float((out["fp32_exact"][idx].cpu().double().reshape(ref64.shape) - ref64).abs().max())
        d = float((out["fp32"][idx] - out["fp32_exact"][idx]).abs().max())
        print(f"D={D} {name}: |split - fp64| {e_split:.2e}  |exact - fp64| {e_exact:.2e}  |fp32 oracle - fp64| {e_32:.2e}  |split - exact| {d:.2e}")
        assert e_exact <= 2.0 * e_32 + 2e-7 and e_split <= 2.0 * e_32 + 2e-7, name
        assert 0 < d <= 4.0 * e_32 + 1e-6, name
    with pytest.raises(NotImplementedError):
        pkg.build_generator(configs.tiny_G_cfg(32, 2, 1), DEV, seed=0).renderer.set_precision("fp32_exact")


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


def test_planes_roundtrip_and_chain_gemm():
    """Split-fp16 "planes" storage (csrc/chain.hip): x -> (hi, lo) -> x to 2^-22 relative; a chain of 1x1 modulated convs whose
    activations stay in planes against the fp32-MFMA kernel layer by layer; fp32 / bf16 exits; folded ToRGB partial sums."""
    B, C, H = 2, 512, 64
    HW = H * H
    x = cu(weights.det_normal("pl.x", (B, C, H, H), 2.0, 1) * (1.0 + 5.0 * weights.det_unit_uniform("pl.m", (B, C, 1, 1), 2)))
    p = hip.to_planes(x)
    assert p.shape == (B, C // 8, 2, HW, 8) and p.dtype == torch.float16
    back = hip.from_planes(p, H, H)
    # The planes hold x * 2^-e, max|x| 2^-e in [2^14, 2^15) per sample (cips3d_range): 22 significant bits, and an absolute floor
    # of half an fp16 subnormal step (2^-25) in the SCALED domain, i.e. 2^-39 of the sample's maximum
    mx = x.abs().amax(dim=(1, 2, 3), keepdim=True)
    assert bool(((back - x).abs() <= 2.5e-7 * x.abs() + mx * 2.0 ** -39).all())
    # planes layout: channel 8 cb + e of pixel n lives at [b, cb, plane, n, e]
    e = p.cips3d_exp[:, 0].to(torch.float32).view(B, 1, 1, 1)       # (one exponent per sample, written to every pixel block)
    hi = (x * torch.exp2(-e)).to(torch.float16)
    assert torch.equal(p[:, :, 0].permute(0, 1, 3, 2).reshape(B, C, HW), hi.reshape(B, C, HW))
    raw = hip.to_planes(x, ranged=False)                 # e = 0: the halves of x itself
    assert torch.equal(raw[:, :, 0].permute(0, 1, 3, 2).reshape(B, C, HW), x.to(torch.float16).reshape(B, C, HW))
    assert bool(((hip.from_planes(raw, H, H) - x).abs() <= 2.5e-7 * x.abs() + 3.1e-8).all())
    scale = 1.0 / math.sqrt(C)
    nw = torch.full((1,), 0.2, device=DEV)
    cur_p, cur_x = p, x
    for layer in range(3):
        Cout = 512 if layer < 2 else 256
        W = cu(weights.det_normal(f"pl.W{layer}", (1, Cout, C, 1, 1), 1.0, 3))
        s = cu(1.0 + weights.det_uniform(f"pl.s{layer}", (B, C), 0.4, 4))
        bias = cu(weights.det_uniform(f"pl.b{layer}", (Cout,), 0.3, 5))
        nz = cu(weights.det_normal(f"pl.n{layer}", (B if layer == 1 else 1, 1, H, H), 1.0, 6))
        wm_x = hip.modulate_weights(W, s, C, B, Cout, C, 1, scale, True, True)
        wm_s = hip.modulate_weights(W, s, C, B, Cout, C, 1, scale, True, True, split=True)
        if layer < 2:
            Wr = cu(weights.det_normal(f"pl.Wr{layer}", (1, 3, Cout, 1, 1), 1.0, 7))
            wr = hip.modulate_weights(Wr, cu(1.0 + weights.det_uniform("pl.sr", (B, Cout), 0.3, 8)), Cout, B, 3, Cout, 1,
                                      1.0 / math.sqrt(Cout), False, False)
            part_ref = torch.zeros(Cout // 64, B, 3, HW, device=DEV)
            lib = _lib.load()
            ref = torch.empty(B, Cout, H, H, device=DEV)
            nb = HW if nz.shape[0] == B else 0
            _lib.check(lib.cips3d_modconv1x1_torgb(cur_x.data_ptr(), wm_x.data_ptr(), ref.data_ptr(), B, C, Cout, HW, 1, nz.data_ptr(),
                                                   nb, nw.data_ptr(), bias.data_ptr(), wr.data_ptr(), part_ref.data_ptr(), None,
                                                   None, torch.cuda.current_stream().cuda_stream), "ref")
            part = torch.zeros_like(part_ref)
            out_p = hip.modconv1x1_planes(cur_p, wm_s, Cout, HW, "planes", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr,
                                          rgb_part=part)
            got = hip.from_planes(out_p, H, H)
            rng = float(ref.abs().max())
            assert maxdiff(got, ref) < 4e-6 * rng, layer
            # (the two kernels may cut the rows into different blocks -- 64 or 128 rows per slot: what is defined is the slots' sum)
            assert maxdiff(part.sum(0), part_ref.sum(0)) < 4e-6 * float(part_ref.sum(0).abs().max()), layer
            cur_p, cur_x = out_p, ref
        else:       # the low-resolution GEMM that leaves the run: fp32 and bf16 exits, no epilogue
            ref = hip.modconv1x1(cur_x, wm_x, Cout, epilogue=0)
            o32 = hip.modconv1x1_planes(cur_p, wm_s, Cout, HW, "fp32")
            o16 = hip.modconv1x1_planes(cur_p, wm_s, Cout, HW, "bf16")
            rng = float(ref.abs().max())
            assert maxdiff(o32.view_as(ref), ref) < 4e-6 * rng
            assert torch.equal(o16, o32.to(torch.bfloat16))
    assert hip.planes_supported(512, 512, 4096) and not hip.planes_supported(32, 512, 4096)


def test_planes_run_in_the_one_call_forward():
    """The default plan runs conv1 + convs.0-7 and the first low-resolution GEMM on split-fp16 planes (flags bits 2 / 3),
    fed by the render kernel's planes output; the result agrees with the same plan without planes (fp32 storage, split in
    registers) to the rounding of the stored activations (2^-22 relative per value)."""
    from cips_3dplusplus_amd import plan as _plan
    cfg = configs.ffhq_G_cfg(1024, 2)
    G = pkg.build_generator(cfg, DEV, seed=2)
    zs, nb, _ = weights.synth_inputs(cfg, seed=6)
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=torch.tensor([[-0.3, 0.1]], device=DEV))
    kw = dict(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[cu(b) for b in nb],
              nerf_cfg=dict(N_samples=24, perturb=False, static_viewdirs=False))
    old = _plan.PLANES_RUN
    try:
        _plan.PLANES_RUN = True
        G._plans = {}
        a = G(**kw)["rgb"].clone()
        L = list(G._plans.values())[0].plan.layers
        kinds = [(L[i].kind, L[i].flags) for i in range(list(G._plans.values())[0].plan.n_dec_layers)]
        convs64 = [fl for k, fl in kinds if k == 0][:9]
        assert all(fl & 4 and fl & 8 for fl in convs64), kinds            # planes in and out for the nine 64^2 convs
        first_up = [fl for k, fl in kinds if k == 1][0]
        assert first_up & 4 and not first_up & 8
        _plan.PLANES_RUN = False
        G._plans = {}
        b = G(**kw)["rgb"].clone()
        L = list(G._plans.values())[0].plan.layers
        assert not any(L[i].flags & 12 for i in range(list(G._plans.values())[0].plan.n_dec_layers))
    finally:
        _plan.PLANES_RUN = old
        G._plans = {}
    rng = float(b.abs().max())
    d = maxdiff(a, b)
    print(f"planes run vs fp32-storage split run, 1024^2: max-abs {d:.2e} on range {rng:.2f}")
    assert d < 2e-5 * rng


def test_planes_run_ends_in_fp32_when_nothing_up_samples():
    """64^2 output (upsample_list = []): the last ToRGB is the image and cannot be folded, so the run's last conv leaves in
    fp32; tiny generators (channels the planes kernel does not tile) plan no run at all."""
    cfg = configs.ffhq_G_cfg(64, 2)
    G = pkg.build_generator(cfg, DEV, seed=3)
    zs, nb, _ = weights.synth_inputs(cfg, seed=7)
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=torch.tensor([[0.1, 0.0]], device=DEV))
    ncfg = dict(N_samples=6, perturb=False, static_viewdirs=False)
    r = G(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[cu(b) for b in nb], nerf_cfg=ncfg)
    plan = list(G._plans.values())[0].plan
    fl = [plan.layers[i].flags for i in range(plan.n_dec_layers) if plan.layers[i].kind == 0]
    # planes from the first conv on; the run ends (fp32 exit: bit 2 without bit 3) where a ToRGB can no longer be folded --
    # this configuration has nine ToRGBs at 64^2, one more than a fold holds -- and nothing after the exit reads planes
    end = [i for i, x in enumerate(fl) if x & 4 and not x & 8]
    assert fl[0] & 4 and len(end) == 1 and all(x & 8 for x in fl[:end[0]]) and not any(x & 4 for x in fl[end[0] + 1:]), fl
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    cam = O.camera_params(torch.tensor([[0.1, 0.0]]), 64, 6, 0.12)
    ref = O.generator_forward(sd, cfg, zs, cam[0], cam[1], 64, cam[2], cam[3], ncfg, nb)
    assert maxdiff(r["rgb"].cpu(), ref["rgb"]) < 1e-3
    Gt = pkg.build_generator(configs.tiny_G_cfg(32, 2, 1), DEV, seed=1)
    et, ft, nt, fat, _ = Camera.generate_camera_params(8, DEV, locations=torch.zeros(1, 2, device=DEV))
    Gt(zs=[torch.randn(1, 32, device=DEV)] * 2, cam_poses=et, focals=ft, img_size=8, near=nt, far=fat, nerf_cfg=ncfg)
    pt = list(Gt._plans.values())[0].plan
    assert not any(pt.layers[i].flags & 12 for i in range(pt.n_dec_layers))


@pytest.mark.parametrize("C,H,B", [(256, 32, 1), (128, 32, 2), (64, 64, 1), (32, 64, 2)])
def test_fused_stage_split_products_agree_with_fp32(C, H, B):
    """cips3d_fused_up_conv with CIPS3D_GEMM_SPLIT (weights packed CIPS3D_MOD_SPLIT16; the FIR + noise + bias + leaky-ReLU
    epilogue of conv1 splits its outputs into fp16 halves before the LDS hand-off) against the fp32-MFMA instantiation of the
    same kernel on the same inputs: out2, the rgb skip sum and the chained next-stage y agree to ~1e-6 of their range.  (This
    is the test that caught hipcc folding the sqrt(2) gain into the fp16 conversion of `hi`: 2.5e-4 before cips3d_split16
    made the value opaque.)"""
    chains = hip.fused_up_conv_chains(C)
    lib = _lib.load()

    def mod(W, s, flags):
        out = torch.empty(B * W.shape[0] * C, device=DEV)
        _lib.check(lib.cips3d_modulate_weights(W.data_ptr(), s.data_ptr(), C, out.data_ptr(), B, W.shape[0], C, 1,
                                               1.0 / math.sqrt(C), flags, torch.cuda.current_stream().cuda_stream), "mod")
        return out

    y = cu(weights.det_normal("f.y", (B, C, H, H), 1.0, C))
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    fir = cu(k1.outer(k1) / 16.0)
    n1 = cu(weights.det_normal("f.n1", (1, 1, 2 * H, 2 * H), 1.0, 2))
    n2 = cu(weights.det_normal("f.n2", (1, 1, 2 * H, 2 * H), 1.0, 3))
    nw1, nw2 = torch.full((1,), 0.3, device=DEV), torch.full((1,), -0.2, device=DEV)
    b1, b2 = cu(weights.det_uniform("f.b1", (C,), 0.2, 4)), cu(weights.det_uniform("f.b2", (C,), 0.2, 5))
    W2, Wn, Wr = (cu(weights.det_normal(f"f.{k}", shp, 1.0, 6)) for k, shp in (("W2", (C, C)), ("Wn", (C // 2, C)), ("Wr", (3, C))))
    s2, sn, sr = (cu(1.0 + weights.det_uniform(f"f.s{k}", (B, C), 0.3, 7)) for k in range(3))
    wmr = mod(Wr, sr, 0)
    brgb = cu(weights.det_uniform("f.brgb", (3,), 0.1, 8))
    skip = cu(weights.det_normal("f.skip", (B, 3, H, H), 1.0, 9))
    res = {}
    for split in (False, True):
        f16 = hip.MOD_SPLIT16 if split else 0
        wm2 = mod(W2, s2, hip.MOD_DEMODULATE | hip.MOD_PACKED | f16)
        wmn = mod(Wn, sn, hip.MOD_DEMODULATE | hip.MOD_PACKED | hip.MOD_CHAINED | f16) if chains else None
        res[split] = hip.fused_up_conv(y, fir, n1, nw1, b1, wm2, n2, nw2, b2, wmr, brgb, skip, skip_up=True, wm_next=wmn,
                                       split=split)
    assert len(res[True]) == (3 if chains else 2)
    for name, a, b in zip(("out2", "rgb", "y_next"), res[False], res[True]):
        d, rng = float((a - b).abs().max()), float(a.abs().max())
        assert d < 2e-6 * max(rng, 1.0), (name, d, rng)


@pytest.mark.parametrize("Cout,Cin,B,demod", [(512, 512, 2, True), (64, 256, 1, True), (32, 1024, 2, False), (64, 96, 3, True),
                                                (32, 32, 1, True), (32, 2048, 1, True)])
def test_split_and_bf16_weight_fragments_are_the_plain_matrix_rearranged(Cout, Cin, B, demod):
    """CIPS3D_MOD_SPLIT / CIPS3D_MOD_BF16 fragments (8 consecutive channels per 16-byte piece) hold exactly the plain
    modulated matrix: hi / lo halves of
    2^8 wm, resp. its RNE bf16 rounding, at [ot][kb][plane][(q << 4) | (o & 15)][j] with channel = 32 kb + 8 q + j."""
    W = cu(weights.det_normal("mf.W", (1, Cout, Cin, 1, 1), 1.0, Cout + Cin))
    s = cu(1.0 + weights.det_uniform("mf.s", (B, Cin), 0.5, 3))
    scale = 1.0 / math.sqrt(Cin)
    plain = hip.modulate_weights(W, s, Cin, B, Cout, Cin, 1, scale, demod, False).view(B, Cout, Cin)
    # [B, ot, o16, kb, q, j] -> [B, ot, kb, q, o16, j]
    frag = plain.view(B, Cout // 16, 16, Cin // 32, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous()
    sp = hip.modulate_weights(W, s, Cin, B, Cout, Cin, 1, scale, demod, True, split=True)
    got = sp.view(torch.float16).view(B, Cout // 16, Cin // 32, 2, 4, 16, 8)
    x = frag * 256.0
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    assert torch.equal(got[:, :, :, 0], hi) and torch.equal(got[:, :, :, 1], lo)
    bf = hip.modulate_weights(W, s, Cin, B, Cout, Cin, 1, scale, demod, True, bf16=True)
    got16 = bf.view(torch.bfloat16)[: B * Cout * Cin].view(B, Cout // 16, Cin // 32, 4, 16, 8)
    assert torch.equal(got16, frag.to(torch.bfloat16))


def test_the_two_instruction_split_is_the_split():
    """cips3d_split_word / cips3d_split_pair (v_fma_mix{lo,hi}_f16, csrc/common.h): hi = fp16(x), lo = fp16(x - hi), round to
    nearest even, fp16 subnormals kept; the scaled form splits the EXACT product t k (one rounding, not two).  (Compared as
    values: fma(-0, 1, +0) is +0 where a conversion keeps -0 -- the one bit pattern that can differ.)"""
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    mags = torch.tensor([1e-9, 1e-7, 6e-5, 1e-3, 0.3, 1.0, 37.0, 2049.0, 3.0e4])
    x = torch.cat([torch.randn(40000, generator=g)[:, None] * mags[None, :], torch.tensor([[0.0, -0.0, 65504.0, -65504.0, 2.0 ** -24, 2.0 ** -25,
                                                                                           1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -12, -5.5]])]).reshape(-1)
    x = x[x.abs() < 65519.0]
    x = x[: x.numel() // 2 * 2].contiguous()
    for k in (1.0, 1.41421356237309515 * 2.0 ** -3, 0.7853981):
        kk = float(torch.tensor(k, dtype=torch.float32))
        xs = x if k == 1.0 else x[(x * kk).abs() < 65000.0]
        xs = xs[: xs.numel() // 2 * 2].contiguous()
        xd = cu(xs)
        words = torch.empty(xs.numel(), dtype=torch.int32, device=DEV)
        pairs = torch.empty(xs.numel(), dtype=torch.int32, device=DEV)
        _lib.check(lib.cips3d_split_words(xd.data_ptr(), kk, words.data_ptr(), pairs.data_ptr(), xs.numel(), hip.stream_ptr()), "split_words")
        w16 = words.cpu().view(torch.float16).view(-1, 2)            # [hi, lo] per value
        import numpy as np
        r16 = lambda d: torch.from_numpy(d.numpy().astype(np.float16))   # noqa: E731  (numpy rounds a double to half ONCE; torch goes through float)
        p64 = xs.double() * kk                                        # exact
        hi_exact = r16(p64)
        lo_exact = r16(p64 - hi_exact.double())
        p32 = (xs * kk)                                               # the fp32 product, then the plain split
        hi_plain = r16(p32.double())
        lo_plain = r16(p32.double() - hi_plain.double())
        even = torch.arange(xs.numel()) % 2 == 0
        assert torch.equal(w16[even, 0].float(), hi_exact[even].float())
        assert torch.equal(w16[even, 1].float(), lo_exact[even].float())
        assert torch.equal(w16[~even, 0].float(), hi_plain[~even].float())
        assert torch.equal(w16[~even, 1].float(), lo_plain[~even].float())
        if k == 1.0:
            pr = pairs.cpu().view(torch.float16).view(-1, 2, 2)      # [pair][hi|lo][element]
            xe = xs.view(-1, 2)
            hi = r16(xe.double())
            lo = r16(xe.double() - hi.double())
            assert torch.equal(pr[:, 0].float(), hi.float()) and torch.equal(pr[:, 1].float(), lo.float())


def test_half_chip_tiles_give_the_same_bits():
    """cips3d_range.half_chip (a second view in flight: pipeline.ViewPipeline): the 64^2 chain layer as 128 tiles of 128 x 128 on
    half the CUs -- same planes, same exponents and patch maxima, and the SAME ToRGB slots (64-row slots: four 16-row tiles summed
    in tile order whatever the workgroup's height) as the 256 tiles of 64 x 128."""
    B, C, H = 1, 512, 64
    HW = H * H
    x = cu(weights.det_normal("hc.x", (B, C, H, H), 1.5, 1))
    W = cu(weights.det_normal("hc.W", (1, C, C, 1, 1), 1.0, 2))
    s = cu(1.0 + weights.det_uniform("hc.s", (B, C), 0.3, 3))
    bias = cu(weights.det_uniform("hc.b", (C,), 0.2, 4))
    nw = cu(torch.tensor([0.3]))
    nz = cu(weights.det_normal("hc.n", (1, 1, H, H), 1.0, 5))
    wm_s = hip.modulate_weights(W, s, C, B, C, C, 1, 1.0 / math.sqrt(C), True, True, split=True)
    Wr = cu(weights.det_normal("hc.Wr", (1, 3, C, 1, 1), 1.0, 6))
    wr = hip.modulate_weights(Wr, s, C, B, 3, C, 1, 1.0 / math.sqrt(C), False, False)
    xp = hip.to_planes(x)
    outs = []
    for half in (False, True):
        part = torch.full((C // 64, B, 3, HW), float("nan"), device=DEV)
        o = hip.modconv1x1_planes(xp, wm_s, C, HW, "planes", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr, rgb_part=part,
                                  half_chip=half)
        outs.append((o, o.cips3d_exp, o.cips3d_pmax, part))
    (o0, e0, p0, part0), (o1, e1, p1, part1) = outs
    assert torch.equal(o0, o1) and torch.equal(e0, e1) and torch.equal(part0, part1) and bool(torch.isfinite(part1).all())
    # patch maxima: a 128-row workgroup's wave leaves ONE maximum for its two 16-channel patches -- a bound of each, and the same
    # maximum over the channels of a pixel block (what the consuming layer takes)
    assert bool((p1 >= p0).all()) and torch.equal(p0.amax(-1), p1.amax(-1))
