"""bf16 STORAGE of the pre-FIR activations of the fused up-sampling stages (CIPS3D_Y_BF16; Generator.set_decoder_precision
("bf16_storage"), BASELINE config 3): kernel-level identities (a bf16 y_lo gives exactly what the fp32 kernel gives on the
rounded values; y_next is the RNE rounding of the fp32 kernel's y_next) and the generator against the oracle with the same
roundings (oracle/path.py: bf16_decoder="storage") and against exact fp32 (PSNR)."""
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


def _mod(W, s, C, flags):
    B = s.shape[0]
    out = torch.empty(B * W.shape[0] * C, device=DEV)
    _lib.check(_lib.load().cips3d_modulate_weights(W.data_ptr(), s.data_ptr(), C, out.data_ptr(), B, W.shape[0], C, 1,
                                                   1.0 / math.sqrt(C), flags, torch.cuda.current_stream().cuda_stream), "mod")
    return out


def test_gemm_bf16_output_is_the_rounded_fp32_output():
    B, Cin, Cout, H = 2, 512, 256, 64
    x = cu(weights.det_normal("ys.x", (B, Cin, H, H), 1.0, 1))
    W = cu(weights.det_normal("ys.W", (Cout, Cin), 1.0, 1))
    s = cu(1.0 + weights.det_uniform("ys.s", (B, Cin), 0.3, 1))
    wm = _mod(W, s, Cin, hip.MOD_DEMODULATE | hip.MOD_PACKED)
    y32 = hip.modconv1x1(x, wm, Cout, epilogue=0, bf16=True)
    y16 = hip.modconv1x1(x, wm, Cout, epilogue=0, bf16=True, out_bf16=True)
    assert y16.dtype == torch.bfloat16 and y16.shape == y32.shape
    assert torch.equal(y16, y32.to(torch.bfloat16))               # same accumulators, RNE on the store
    with pytest.raises(RuntimeError):
        hip.modconv1x1(x, wm, Cout, epilogue=1, bias=torch.zeros(Cout, device=DEV), bf16=True, out_bf16=True)


@pytest.mark.parametrize("C,H,B", [(256, 32, 1), (128, 32, 2), (64, 32, 1), (32, 64, 1)])
def test_fused_stage_with_bf16_y(C, H, B):
    """fused_up_conv on a bf16 y_lo == the fp32-storage kernel on the same (already rounded) values, bit for bit; its y_next
    is the RNE rounding of that kernel's y_next."""
    chains = hip.fused_up_conv_chains(C)
    y32 = cu(weights.det_normal("ysf.y", (B, C, H, H), 1.0, C))
    y16 = y32.to(torch.bfloat16)
    y_rounded = y16.float()
    fir = cu(torch.tensor([1.0, 3.0, 3.0, 1.0]).outer(torch.tensor([1.0, 3.0, 3.0, 1.0])) / 16.0)
    n1 = cu(weights.det_normal("ysf.n1", (1, 1, 2 * H, 2 * H), 1.0, 2))
    n2 = cu(weights.det_normal("ysf.n2", (B, 1, 2 * H, 2 * H), 1.0, 3))
    nw1, nw2 = torch.full((1,), 0.3, device=DEV), torch.full((1,), -0.2, device=DEV)
    b1, b2 = cu(weights.det_uniform("ysf.b1", (C,), 0.2, 4)), cu(weights.det_uniform("ysf.b2", (C,), 0.2, 5))
    W2, Wn, Wr = (cu(weights.det_normal(f"ysf.{k}", shp, 1.0, 6)) for k, shp in (("W2", (C, C)), ("Wn", (C // 2, C)), ("Wr", (3, C))))
    s2, sn, sr = (cu(1.0 + weights.det_uniform(f"ysf.s{k}", (B, C), 0.3, 7)) for k in range(3))
    wm2 = _mod(W2, s2, C, hip.MOD_DEMODULATE | hip.MOD_PACKED)
    wmn = _mod(Wn, sn, C, hip.MOD_DEMODULATE | hip.MOD_PACKED | hip.MOD_CHAINED) if chains else None
    wmr = _mod(Wr, sr, C, 0)
    brgb = cu(weights.det_uniform("ysf.brgb", (3,), 0.1, 8))
    skip = cu(weights.det_normal("ysf.skip", (B, 3, H, H), 1.0, 9))
    kw = dict(skip_up=True, bf16=True)
    if chains:
        o_a, rgb_a, yn_a = hip.fused_up_conv(y_rounded, fir, n1, nw1, b1, wm2, n2, nw2, b2, wmr, brgb, skip, wm_next=wmn, **kw)
        o_b, rgb_b, yn_b = hip.fused_up_conv(y16, fir, n1, nw1, b1, wm2, n2, nw2, b2, wmr, brgb, skip, wm_next=wmn, **kw)
        assert yn_b.dtype == torch.bfloat16 and torch.equal(yn_b, yn_a.to(torch.bfloat16))
    else:
        o_a, rgb_a = hip.fused_up_conv(y_rounded, fir, n1, nw1, b1, wm2, n2, nw2, b2, wmr, brgb, skip, **kw)
        o_b, rgb_b = hip.fused_up_conv(y16, fir, n1, nw1, b1, wm2, n2, nw2, b2, wmr, brgb, skip, **kw)
    assert torch.equal(o_a, o_b) and torch.equal(rgb_a, rgb_b)
    with pytest.raises(RuntimeError):                                 # bf16 storage needs the bf16 GEMM mode
        hip.fused_up_conv(y16, fir, n1, nw1, b1, wm2, n2, nw2, b2, wmr, brgb, skip, skip_up=True, bf16=False)


def _psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10 * math.log10(float(b.abs().max()) ** 2 / mse)


def test_generator_bf16_storage_vs_oracle_and_fp32():
    """256^2 generator (both up-sampling stages and the two flat blocks behind them fused and chained): the storage mode against the oracle with the same
    operand / storage roundings (bounded like the compute mode: accumulation order + rare rounding flips), against exact
    fp32 (PSNR), and the plan really switches the flag."""
    cfg = configs.ffhq_G_cfg(256, 2)
    G = pkg.build_generator(cfg, DEV, seed=1)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    zs, nb, _ = weights.synth_inputs(cfg, batch=1, seed=21)
    locs = torch.tensor([[0.25, -0.05]])
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=cu(locs))
    ncfg = dict(N_samples=12, perturb=False, static_viewdirs=False)
    kw = dict(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[cu(b) for b in nb],
              nerf_cfg=ncfg)
    r32 = G(**kw)["rgb"].clone()
    G.set_decoder_precision("bf16")
    r16 = G(**kw)["rgb"].clone()
    G.set_decoder_precision("bf16_storage")
    r16s = G(**kw)["rgb"].clone()
    plan = list(G._plans.values())[0]
    assert plan.plan.decoder_bf16 == 2
    assert not torch.equal(r16s, r16)
    cam = O.camera_params(locs, 64, 6, 0.12)
    ref = O.generator_forward(sd, cfg, zs, cam[0], cam[1], 64, cam[2], cam[3], ncfg, nb, bf16_decoder="storage")["rgb"]
    scale = float(ref.abs().max())
    d = maxdiff(r16s.cpu(), ref)
    p_s, p_c = _psnr(r16s, r32), _psnr(r16, r32)
    print(f"bf16 storage mode, 256^2: vs oracle(storage) max-abs {d:.3e} on range {scale:.2f}; PSNR vs fp32 {p_s:.1f} dB "
          f"(compute-only mode {p_c:.1f} dB)")
    # (four tensors are stored as bf16 here -- the conv results of the two up-sampling blocks and of the two flat blocks behind
    # them: a value that the two accumulation orders round to different bf16 neighbours is a 2^-8 relative step)
    assert d < 2e-2 * scale and _psnr(r16s.cpu(), ref) > 53.0
    assert p_s > 35.0
    G.set_decoder_precision("fp32")
    assert torch.equal(G(**kw)["rgb"], r32)


# ------------------------------------------------------------------------------ planes16: the 64^2 run on one bf16 plane
def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("B,C,Cout,H,W", [(1, 512, 512, 64, 64), (4, 512, 512, 64, 64), (2, 256, 512, 16, 16), (3, 64, 128, 8, 10),
                                           (1, 128, 64, 4, 5)])
def test_planes16_layer_vs_rounded_operands(B, C, Cout, H, W):
    """cips3d_modconv1x1_planes16 (csrc/chain.hip, NP = 1) against its definition: both operands rounded to bf16 (round to
    nearest even), exact products, fp32 accumulation; NoiseInjection + bias + leaky ReLU and the folded ToRGB on the unrounded
    fp32 result; planes16 / bf16 exits are the RNE rounding of the fp32 exit.  Covers batches, ragged pixel counts and both
    noise strides."""
    HW = H * W
    x = cu(weights.det_normal("p16.x", (B, C, H, W), 2.0, 1) * (1.0 + 5.0 * weights.det_unit_uniform("p16.m", (B, C, 1, 1), 2)))
    p = hip.to_planes16(x)
    assert p.shape == (B, C // 8, HW, 8) and p.dtype == torch.bfloat16
    # layout: channel 8 cb + e of pixel n lives at [b, cb, n, e]; the values are the RNE roundings
    assert torch.equal(p.permute(0, 1, 3, 2).reshape(B, C, HW), x.to(torch.bfloat16).reshape(B, C, HW))
    assert torch.equal(hip.from_planes16(p, H, W), _bf(x))
    scale = 1.0 / math.sqrt(C)
    Wt = cu(weights.det_normal("p16.W", (1, Cout, C, 1, 1), 1.0, 3))
    s = cu(1.0 + weights.det_uniform("p16.s", (B, C), 0.4, 4))
    bias = cu(weights.det_uniform("p16.b", (Cout,), 0.3, 5))
    nw = torch.full((1,), 0.2, device=DEV)
    wm = hip.modulate_weights(Wt, s, C, B, Cout, C, 1, scale, True, False).view(B, Cout, C)
    wm16 = hip.modulate_weights(Wt, s, C, B, Cout, C, 1, scale, True, True, bf16=True)
    ref = torch.bmm(_bf(wm).double(), _bf(x).reshape(B, C, HW).double())            # exact products, fp64 sum
    rng = float(ref.abs().max())
    o32, _ = hip.modconv1x1_planes16(p, wm16, Cout, HW, "fp32")
    assert maxdiff(o32.double(), ref) < 3e-6 * rng
    o16, _ = hip.modconv1x1_planes16(p, wm16, Cout, HW, "bf16")
    assert torch.equal(o16, o32.to(torch.bfloat16))
    for per_sample_noise in (False, True):
        nz = cu(weights.det_normal("p16.n", (B if per_sample_noise else 1, 1, H, W), 1.0, 6))
        Wr = cu(weights.det_normal("p16.Wr", (1, 3, Cout, 1, 1), 1.0, 7))
        wr = hip.modulate_weights(Wr, cu(1.0 + weights.det_uniform("p16.sr", (B, Cout), 0.3, 8)), Cout, B, 3, Cout, 1,
                                  1.0 / math.sqrt(Cout), False, False)
        part = torch.full((Cout // 64, B, 3, HW), float("nan"), device=DEV)
        out_p, nblk = hip.modconv1x1_planes16(p, wm16, Cout, HW, "planes16", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr,
                                              rgb_part=part)
        assert nblk == Cout // 64
        act = torch.nn.functional.leaky_relu(ref + (0.2 * nz.double()).reshape(-1, 1, HW) + bias.double().view(1, -1, 1), 0.2) * math.sqrt(2.0)
        got = hip.from_planes16(out_p, H, W).reshape(B, Cout, HW)
        arng = float(act.abs().max())
        # the stored value is the bf16 rounding of the fp32 activation: half a bf16 ulp (2^-9 relative) + the fp32 error
        assert bool(((got.double() - act).abs() <= 2.0 ** -8 * act.abs() + 3e-6 * arng).all())
        assert float((got.double() - _bf(act.float()).double()).abs().mean()) < 2e-6 * arng      # ... and nearly always THE rounding
        rgb_ref = torch.bmm(wr.view(B, 3, Cout).double(), act)
        assert maxdiff(part[:nblk].sum(0).double(), rgb_ref) < 3e-6 * float(rgb_ref.abs().max()) + 3e-6 * arng
    # a second layer on the stored planes16 == the definition applied to the stored (rounded) activation
    if Cout % 64 == 0 and C % 64 == 0:
        W2 = cu(weights.det_normal("p16.W2", (1, C, Cout, 1, 1), 1.0, 9))
        s2 = cu(1.0 + weights.det_uniform("p16.s2", (B, Cout), 0.4, 10))
        wm2 = hip.modulate_weights(W2, s2, Cout, B, C, Cout, 1, 1.0 / math.sqrt(Cout), True, False).view(B, C, Cout)
        wm2_16 = hip.modulate_weights(W2, s2, Cout, B, C, Cout, 1, 1.0 / math.sqrt(Cout), True, True, bf16=True)
        o2, _ = hip.modconv1x1_planes16(out_p, wm2_16, C, HW, "fp32")
        ref2 = torch.bmm(_bf(wm2).double(), hip.from_planes16(out_p, H, W).reshape(B, Cout, HW).double())
        assert maxdiff(o2.double(), ref2) < 3e-6 * float(ref2.abs().max())


def test_planes16_entry_point_refusals():
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(1, 8, 64, 8, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(64 * 64, device=DEV)
    o = torch.zeros(1, 64, 64, device=DEV)
    args = (1, 64, 64, 64, 0, None, 0, None, None, None, None, None, None, st)
    assert lib.cips3d_modconv1x1_planes16(x.data_ptr(), w.data_ptr(), o.data_ptr(), 1, *args) != 0          # split planes out: not here
    assert lib.cips3d_modconv1x1_planes16(None, w.data_ptr(), o.data_ptr(), 0, *args) != 0
    assert lib.cips3d_modconv1x1_planes16(x.data_ptr(), w.data_ptr(), o.data_ptr(), 0, 1, 32, 64, 64, 0, None, 0, None, None, None,
                                          None, None, None, st) != 0                                           # Cin % 64
    assert lib.cips3d_modconv1x1_planes16(x.data_ptr(), w.data_ptr(), o.data_ptr(), 0, 1, 64, 64, 64, 1, None, 0, None, None, None,
                                          None, None, None, st) != 0                                           # epilogue without bias
    assert lib.cips3d_to_planes16(o.data_ptr(), x.data_ptr(), 1, 60, 64, st) != 0                              # C % 8
    # CIPS3D_MOD_BF16 needs the plain packed layout and Cin % 32 == 0
    Wt = torch.zeros(64, 48, device=DEV)
    s = torch.ones(1, 48, device=DEV)
    assert lib.cips3d_modulate_weights(Wt.data_ptr(), s.data_ptr(), 48, w.data_ptr(), 1, 64, 48, 1, 1.0, 1 | 2 | 128, st) != 0
    assert lib.cips3d_modulate_weights(Wt.data_ptr(), s.data_ptr(), 48, w.data_ptr(), 1, 64, 64, 1, 1.0, 1 | 2 | 16 | 128, st) != 0


def test_bf16_modes_run_the_64sq_layers_on_planes16():
    """Both bf16 precisions plan conv1 + convs.0-7 and the first low-resolution GEMM on planes16 (flags bits 2 / 3 / 5).  The
    stored bf16 value IS the operand the fp32-stored form of the mode rounds to (test_planes16_layer_vs_rounded_operands), so
    the two forms differ by fp32 summation order only -- which in a bf16 network is not small: a sum that lands on the other
    side of a rounding boundary (p ~ 5e-4 per value) moves an activation by a whole bf16 ulp, and after a few layers the
    rounding noise of the two evaluations is decorrelated pixel by pixel.  Measured at 1024^2 (tools, round 2): planes16 and
    the fp32-stored form are 3.5e-3 apart (mean abs, range 11.2), each of them 7.3e-3 from the CPU oracle's bf16 evaluation
    (a third summation order), all three 1.24e-2 from exact fp32.  Bounds: the two forms are closer to each other than
    half the mode's own error, and equally far from fp32."""
    from cips_3dplusplus_amd import plan as _plan
    cfg = configs.ffhq_G_cfg(1024, 2)
    G = pkg.build_generator(cfg, DEV, seed=2)
    zs, nb, _ = weights.synth_inputs(cfg, seed=6)
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=torch.tensor([[-0.3, 0.1]], device=DEV))
    kw = dict(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[cu(b) for b in nb],
              nerf_cfg=dict(N_samples=24, perturb=False, static_viewdirs=False))
    r32 = G(**kw)["rgb"].clone()
    old = _plan.PLANES16_RUN
    try:
        for prec in ("bf16", "bf16_storage"):
            G.set_decoder_precision(prec)
            _plan.PLANES16_RUN = True
            G._plans = {}
            a = G(**kw)["rgb"].clone()
            pl = list(G._plans.values())[0].plan
            kinds = [(pl.layers[i].kind, pl.layers[i].flags) for i in range(pl.n_dec_layers)]
            convs64 = [fl for k, fl in kinds if k == 0][:9]
            assert all((fl & 4) and (fl & 8) and (fl & 32) for fl in convs64), kinds
            first_up = [fl for k, fl in kinds if k == 1][0]
            assert (first_up & 4) and (first_up & 32), kinds
            _plan.PLANES16_RUN = False
            G._plans = {}
            b = G(**kw)["rgb"].clone()
            pl2 = list(G._plans.values())[0].plan
            assert not any(pl2.layers[i].flags & 32 for i in range(pl2.n_dec_layers))
            d_ab, d_a, d_b = float((a - b).abs().mean()), float((a - r32).abs().mean()), float((b - r32).abs().mean())
            print(f"{prec}: planes16 vs fp32-stored {d_ab:.3e}; vs exact fp32 {d_a:.3e} / {d_b:.3e} (range {float(r32.abs().max()):.1f})")
            assert d_ab < 0.5 * d_b, prec
            assert abs(d_a - d_b) < 0.05 * d_b, prec
            assert maxdiff(a, b) < 2e-2 * float(r32.abs().max()), prec
    finally:
        _plan.PLANES16_RUN = old
        G._plans = {}
        G.set_decoder_precision("fp32")


def test_config3_forward_is_deterministic():
    """Batch 4 at 1024^2 in both bf16 precisions: repeated forwards on fixed inputs are bit-identical.  (Two workgroups of the
    bf16 chain kernel share a CU at this batch size -- the occupancy under which the packed-FMA form of its ToRGB fold returned
    run-to-run different sums; chain.hip is built without SLP vectorisation since.)"""
    G = pkg.build_generator(configs.ffhq_G_cfg(1024, 2), DEV, seed=0)
    B = 4
    g = torch.Generator(device=DEV).manual_seed(3)
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=0.2 * torch.randn(B, 2, device=DEV, generator=g))
    zs = [torch.randn(B, 256, device=DEV, generator=g), torch.randn(B, 256, device=DEV, generator=g)]
    kw = dict(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=G.create_noise_bufs(64, DEV),
              nerf_cfg=dict(N_samples=24, perturb=False, static_viewdirs=False))
    try:
        for prec in ("bf16", "bf16_storage"):
            G.set_decoder_precision(prec)
            ref = G(**kw)["rgb"].clone()
            for _ in range(25):
                assert torch.equal(G(**kw)["rgb"], ref), prec
    finally:
        G.set_decoder_precision("fp32")


@pytest.mark.parametrize("kind", ["planes16", "planes", "gemm_torgb", "fused_stage"])
def test_torgb_folds_repeat_bit_for_bit(kind):
    """The folded ToRGB partial sums (csrc/chain.hip both kernels, cips3d_modconv1x1_torgb, the fused up-sampling stage) at batch 4
    -- two and more workgroups per CU -- are the same bits on every run.  (With hipcc's SLP pairing of the fold's accumulations
    into v_pk_fma_f32 the planes16 kernel's sums differed on EVERY repeat: chain.hip's build note.  tools/fold_repeat.py is the
    longer form of this test.)"""
    import ctypes as C
    B, Cc, H = 4, 512, 64
    HW = H * H
    lib = _lib.load()
    x = cu(weights.det_normal("rp.x", (B, Cc, H, H), 2.0, 1))
    scale = 1.0 / math.sqrt(Cc)
    Wt = cu(weights.det_normal("rp.W", (1, Cc, Cc, 1, 1), 1.0, 3))
    s = cu(1.0 + weights.det_uniform("rp.s", (B, Cc), 0.4, 4))
    bias = cu(weights.det_uniform("rp.b", (Cc,), 0.3, 5))
    nw = torch.full((1,), 0.2, device=DEV)
    nz = cu(weights.det_normal("rp.n", (1, 1, H, H), 1.0, 6))
    Wr = cu(weights.det_normal("rp.Wr", (1, 3, Cc, 1, 1), 1.0, 7))
    wr = hip.modulate_weights(Wr, cu(1.0 + weights.det_uniform("rp.sr", (B, Cc), 0.3, 8)), Cc, B, 3, Cc, 1, 1.0 / math.sqrt(Cc), False, False)
    if kind == "planes16":
        p = hip.to_planes16(x)
        wm = hip.modulate_weights(Wt, s, Cc, B, Cc, Cc, 1, scale, True, True, bf16=True)

        def run():
            part = torch.full((Cc // 64, B, 3, HW), float("nan"), device=DEV)
            hip.modconv1x1_planes16(p, wm, Cc, HW, "planes16", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr, rgb_part=part)
            return part
    elif kind == "planes":
        p = hip.to_planes(x)
        wm = hip.modulate_weights(Wt, s, Cc, B, Cc, Cc, 1, scale, True, True, split=True)

        def run():
            part = torch.full((Cc // 64, B, 3, HW), float("nan"), device=DEV)
            hip.modconv1x1_planes(p, wm, Cc, HW, "planes", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr, rgb_part=part)
            return part
    elif kind == "gemm_torgb":
        wm = hip.modulate_weights(Wt, s, Cc, B, Cc, Cc, 1, scale, True, True)

        def run():
            part = torch.full((Cc // 64, B, 3, HW), float("nan"), device=DEV)
            out = torch.empty(B, Cc, H, H, device=DEV)
            nblk = C.c_int(0)
            _lib.check(lib.cips3d_modconv1x1_torgb(x.data_ptr(), wm.data_ptr(), out.data_ptr(), B, Cc, Cc, HW, 1 | hip.GEMM_BF16, nz.data_ptr(), 0,
                                                   nw.data_ptr(), bias.data_ptr(), wr.data_ptr(), part.data_ptr(), C.byref(nblk), None,
                                                   hip.stream_ptr()), "cips3d_modconv1x1_torgb")
            return part[:nblk.value]                    # (the library reports how many row blocks its tiling wrote)
    else:
        Cs, Hs = 64, 128
        y = cu(weights.det_normal("rp.y", (B, Cs, Hs, Hs), 1.0, 9))
        k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
        fir = cu(k1.outer(k1) / 16.0)
        n1 = cu(weights.det_normal("rp.n1", (1, 1, 2 * Hs, 2 * Hs), 1.0, 2))
        n2 = cu(weights.det_normal("rp.n2", (1, 1, 2 * Hs, 2 * Hs), 1.0, 3))
        b1, b2 = cu(weights.det_uniform("rp.b1", (Cs,), 0.2, 4)), cu(weights.det_uniform("rp.b2", (Cs,), 0.2, 5))
        W2 = cu(weights.det_normal("rp.W2", (1, Cs, Cs, 1, 1), 1.0, 6))
        s2 = cu(1.0 + weights.det_uniform("rp.s2", (B, Cs), 0.3, 7))
        wm2 = hip.modulate_weights(W2, s2, Cs, B, Cs, Cs, 1, 1.0 / math.sqrt(Cs), True, True)
        Wr2 = cu(weights.det_normal("rp.Wr2", (1, 3, Cs, 1, 1), 1.0, 8))
        wr2 = hip.modulate_weights(Wr2, s2, Cs, B, 3, Cs, 1, 1.0 / math.sqrt(Cs), False, False)
        brgb = cu(weights.det_uniform("rp.brgb", (3,), 0.1, 8))
        skip = cu(weights.det_normal("rp.skip", (B, 3, Hs, Hs), 1.0, 9))

        def run():
            return hip.fused_up_conv(y, fir, n1, nw, b1, wm2, n2, nw, b2, wr2, brgb, skip, skip_up=True, want_out2=False, bf16=True)[1]
    first = run().clone()
    assert bool(torch.isfinite(first).all())
    for _ in range(12):
        assert torch.equal(run(), first)
