"""Range safety of the decoder's default arithmetic (split-fp16 operands; cips3d_range in include/cips3d_hip.h).

fp16 has 5 exponent bits, the reference's fp32 convolution (models/model_v3.py:296-312) 8: an unscaled (hi, lo) pair turns
into inf / -inf above 65504 and loses fp32's relative accuracy below ~2^-3.  Every split of the decoder therefore happens on
x * 2^-e with one power of two per (tensor, sample) taken from a rigorous bound of max|x|.  These tests drive activations to
2^17 and 2^-14 times their usual size (and further) through every split site -- the stand-alone GEMM, the planes run, the fused
up-sampling stage, the one-call forward -- and hold the results to (a) the accuracy of plain fp32 against an fp64 evaluation
and (b) EXACT power-of-two equivariance: scaling the inputs by 2^k scales the outputs by 2^k bit for bit, which only holds
when no operand overflowed, lost low bits to fp16's subnormal range, or was scaled by anything but a power of two."""
import ctypes as C
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
SCALES = [17, -14, 40, -60]          # powers of two applied to the activations (the verdict's two and two far ones)


def cu(t):
    return t.to(DEV).contiguous()


def test_absmax_and_layer_constants():
    x = cu(weights.det_normal("rg.x", (3, 1000, 37), 1.0, 1))
    x[1] *= 1e-9
    x[2, 17, 5] = -7.5e8
    am = hip.absmax(x)
    assert am.shape == (3, hip.amax_floats())
    assert torch.equal(hip.amax_value(am), x.abs().amax(dim=(1, 2)))
    # unaligned rows take the scalar path
    y = x.view(-1)[1:1 + 2 * 999].view(2, 999)
    assert torch.equal(hip.amax_value(hip.absmax(y.clone())), y.abs().amax(dim=1))
    # lconst = {sqrt(2) (|nw| nb + max|bias|), sqrt(2) gain}; with a FIR the gain is its largest polyphase L1 norm
    bias = cu(weights.det_uniform("rg.b", (96,), 3.0, 2))
    nw = torch.full((1,), -0.25, device=DEV)
    noise = cu(weights.det_normal("rg.n", (1, 1, 32, 32), 1.0, 3))
    lc = hip.range_consts(2, bias, nw, 16.0, noise=noise, noise_bound=0.5).cpu()
    nb = max(float(noise.abs().max()), 0.5)
    want0 = math.sqrt(2.0) * (0.25 * nb + float(bias.abs().max()))
    assert torch.allclose(lc[:, 0], torch.full((2,), want0), rtol=1e-6) and torch.allclose(lc[:, 1], torch.full((2,), 16 * math.sqrt(2.0)))
    assert torch.equal(lc[:, 2:], torch.zeros(2, 2))
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    fir = cu(k1.outer(k1) / 16.0)
    lcf = hip.range_consts(1, bias, None, 99.0, fir=fir).cpu()
    assert abs(float(lcf[0, 1]) - math.sqrt(2.0)) < 1e-6 and abs(float(lcf[0, 0]) - math.sqrt(2.0) * float(bias.abs().max())) < 1e-5
    fir2 = cu(torch.arange(16, dtype=torch.float32).view(4, 4) - 6.0)
    gains = [sum(abs(float(fir2[py + 2 * a, px + 2 * b_])) for a in range(2) for b_ in range(2)) for py in range(2) for px in range(2)]
    assert abs(float(hip.range_consts(1, bias, None, 1.0, fir=fir2)[0, 1]) - math.sqrt(2.0) * max(gains)) < 1e-4


def test_modulate_table_writes_constants_and_row_l1_norms():
    """The one-call forward's source of lconst: the modulate table.  Row 0's wave writes c0 / c1; a NON-demodulated conv has no
    unit row norm to lean on, so every row raises l1 to its L1 norm."""
    lib = _lib.load()
    B, Cout, Cin = 2, 64, 96
    W = cu(weights.det_normal("mt.W", (Cout, Cin), 1.0, 1))
    s = cu(1.0 + weights.det_uniform("mt.s", (B, Cin), 0.5, 2))
    bias = cu(weights.det_uniform("mt.b", (Cout,), 2.0, 3))
    nw = torch.full((1,), 0.5, device=DEV)
    scale = 1.0 / math.sqrt(Cin)
    outs, lcs = [], []
    descs = (_lib.ModulateDesc * 2)()
    for i, demod in enumerate((True, False)):
        out = torch.empty(B * Cout * Cin, device=DEV)
        lc = torch.zeros(B, 4, device=DEV)
        d = descs[i]
        d.W, d.s, d.out, d.s_stride = W.data_ptr(), s.data_ptr(), out.data_ptr(), Cin
        d.Cout, d.Cin, d.ksq, d.flags, d.scale, d.row_begin = Cout, Cin, 1, (1 if demod else 0), scale, i * Cout
        d.lconst, d.bias, d.n_bias, d.noise_w = lc.data_ptr(), bias.data_ptr(), Cout, nw.data_ptr()
        outs.append(out)
        lcs.append(lc)
    tab = torch.frombuffer(bytearray(bytes(memoryview(descs))), dtype=torch.uint8).to(DEV)
    _lib.check(lib.cips3d_modulate_table(tab.data_ptr(), 2, 2 * Cout, B, 6.0, hip.stream_ptr()), "cips3d_modulate_table")
    c0 = math.sqrt(2.0) * (0.5 * 6.0 + float(bias.abs().max()))
    for lc in lcs:
        assert torch.allclose(lc[:, 0].cpu(), torch.full((B,), c0), rtol=1e-6)
        assert torch.allclose(lc[:, 1].cpu(), torch.full((B,), math.sqrt(2.0 * Cin)), rtol=1e-6)
    assert torch.equal(lcs[0][:, 2].cpu(), torch.zeros(B))                          # demodulated: sqrt(Cin) is the bound
    l1 = outs[1].view(B, Cout, Cin).abs().double().sum(-1).amax(-1)
    assert torch.allclose(lcs[1][:, 2].cpu().double(), l1.cpu(), rtol=1e-5)
    assert bool((lcs[1][:, 2].cpu().double() >= l1.cpu() * (1 - 1e-6)).all())


def _gemm_case(cin, cout, hw, B):
    x = weights.det_normal("rg.gx", (B, cin, hw), 1.5, cin) * (1.0 + 3.0 * weights.det_unit_uniform("rg.gm", (B, cin, 1), cin))
    W = weights.det_normal("rg.gW", (cout, cin), 1.0, cout)
    s = 1.0 + weights.det_uniform("rg.gs", (B, cin), 0.5, 3)
    scale = 1.0 / math.sqrt(cin)
    w64 = (scale * W.double())[None] * s.double()[:, None, :]
    w64 = w64 * torch.rsqrt((w64 ** 2).sum(-1, keepdim=True) + 1e-8)
    w32 = (scale * W)[None] * s[:, None, :]
    w32 = w32 * torch.rsqrt((w32 ** 2).sum(-1, keepdim=True) + 1e-8)
    return x, W, s, scale, w64, w32


@pytest.mark.parametrize("k", SCALES)
def test_split_gemm_keeps_fp32s_range(k):
    """cips3d_modconv1x1 in CIPS3D_GEMM_SPLIT mode on activations 2^k times their usual size: as accurate as fp32 against fp64
    (the bound of test_split_gemm_is_as_accurate_as_fp32), and bit-for-bit 2^k times the unscaled result."""
    cin, cout, hw, B = 512, 256, 1024, 2
    x, W, s, scale, w64, w32 = _gemm_case(cin, cout, hw, B)
    f = 2.0 ** k
    xs = x * f
    xs[1] *= 2.0 ** -9                                   # the samples of a batch get their own exponents
    ref64 = torch.bmm(w64, xs.double())
    ref32 = torch.bmm(w32, xs)
    Wd, sd_ = cu(W.view(1, cout, cin, 1, 1)), cu(s)
    wm_s = hip.modulate_weights(Wd, sd_, cin, B, cout, cin, 1, scale, True, True, split=True)
    wm_x = hip.modulate_weights(Wd, sd_, cin, B, cout, cin, 1, scale, True, True)
    got = hip.modconv1x1(cu(xs.view(B, cin, hw, 1)), wm_s, cout, epilogue=0, split=True)
    exact = hip.modconv1x1(cu(xs.view(B, cin, hw, 1)), wm_x, cout, epilogue=0).view(B, cout, hw).cpu()
    assert bool(torch.isfinite(got).all())
    # the epilogue recorded the output's maximum (the next split GEMM's scale)
    assert torch.equal(hip.amax_value(hip.amax_of(got, measure=False)), got.abs().amax(dim=(1, 2, 3)))
    got = got.view(B, cout, hw).cpu()
    for b in range(B):
        rng = float(ref64[b].abs().max())
        e_split, e_exact, e_32 = (float((t[b].double() - ref64[b]).abs().max()) for t in (got, exact, ref32))
        print(f"2^{k} sample {b}: |split - fp64| {e_split:.2e}  |fp32 MFMA - fp64| {e_exact:.2e}  |torch fp32 - fp64| {e_32:.2e}  range {rng:.2e}")
        assert e_split <= 1.5 * max(e_exact, e_32) + 1e-7 * rng
    base = hip.modconv1x1(cu(x.view(B, cin, hw, 1)), wm_s, cout, epilogue=0, split=True).view(B, cout, hw).cpu()
    assert torch.equal(got[0], base[0] * f) and torch.equal(got[1], base[1] * (f * 2.0 ** -9))


def test_split_gemm_of_zeros_and_of_one_spike():
    """Degenerate ranges: an all-zero sample (bound 0: the exponent clamps, the result is the epilogue's noise + bias) and a
    sample whose maximum is one spike 2^30 above the rest (the spike sets the exponent; the rest still gets >= 9 bits below
    the pair's floor of 2^-40 max, i.e. an absolute error far below fp32's own on that sum)."""
    cin, cout, hw, B = 64, 64, 256, 2
    x = weights.det_normal("rg.zx", (B, cin, hw), 1.0, 1)
    x[0] = 0.0
    x[1, 3, 7] = 2.0 ** 30
    W = cu(weights.det_normal("rg.zW", (1, cout, cin, 1, 1), 1.0, 2))
    s = cu(torch.ones(B, cin))
    bias = cu(weights.det_uniform("rg.zb", (cout,), 0.5, 3))
    nz = cu(weights.det_normal("rg.zn", (1, 1, 16, 16), 1.0, 4))
    nw = torch.full((1,), 0.3, device=DEV)
    scale = 1.0 / math.sqrt(cin)
    wm_s = hip.modulate_weights(W, s, cin, B, cout, cin, 1, scale, True, True, split=True)
    wm_x = hip.modulate_weights(W, s, cin, B, cout, cin, 1, scale, True, True)
    xd = cu(x.view(B, cin, 16, 16))
    got = hip.modconv1x1(xd, wm_s, cout, epilogue=1, noise=nz, noise_w=nw, bias=bias, split=True)
    ref = hip.modconv1x1(xd, wm_x, cout, epilogue=1, noise=nz, noise_w=nw, bias=bias)
    assert bool(torch.isfinite(got).all())
    assert torch.equal(got[0], ref[0])                                  # zeros: nothing but the epilogue
    assert maxdiff(got[1], ref[1]) < 4e-6 * float(ref[1].abs().max())


@pytest.mark.parametrize("k", SCALES)
def test_planes_run_keeps_fp32s_range(k):
    """Three layers of the planes run (csrc/chain.hip) whose input, noise strength and biases are 2^k times the usual: the
    network is positively homogeneous in exactly those, so every layer's output -- planes, the folded ToRGB partial sums, the
    fp32 exit -- is bit for bit 2^k times the unscaled run's, and each layer agrees with the fp32-MFMA kernel."""
    B, C, H = 2, 512, 32
    HW = H * H
    f = 2.0 ** k
    scale = 1.0 / math.sqrt(C)
    x0 = cu(weights.det_normal("rp.x", (B, C, H, H), 1.0, 1) * (1.0 + 5.0 * weights.det_unit_uniform("rp.m", (B, C, 1, 1), 2)))

    def run(fac):
        cur_x = x0 * fac
        cur_p = hip.to_planes(cur_x)
        e = cur_p.cips3d_exp.cpu()               # [B, blocks of 128 pixels]: the conversion writes the sample's everywhere
        assert e.shape == (B, HW // 128) and bool((e == e[:, :1]).all())
        m = cur_x.abs().amax(dim=(1, 2, 3)).cpu()
        st0 = m * 2.0 ** (-e[:, 0].double())
        assert bool(((st0 >= 2.0 ** 14) & (st0 < 2.0 ** 15)).all())
        outs = []
        for layer in range(3):
            Cout = 512 if layer < 2 else 256
            W = cu(weights.det_normal(f"rp.W{layer}", (1, Cout, C, 1, 1), 1.0, 3))
            s = cu(1.0 + weights.det_uniform(f"rp.s{layer}", (B, C), 0.4, 4))
            bias = cu(weights.det_uniform(f"rp.b{layer}", (Cout,), 0.3, 5)) * fac
            nz = cu(weights.det_normal(f"rp.n{layer}", (B if layer == 1 else 1, 1, H, H), 1.0, 6))
            nw = torch.full((1,), 0.2 * fac, device=DEV)
            wm_x = hip.modulate_weights(W, s, C, B, Cout, C, 1, scale, True, True)
            wm_s = hip.modulate_weights(W, s, C, B, Cout, C, 1, scale, True, True, split=True)
            if layer < 2:
                Wr = cu(weights.det_normal(f"rp.Wr{layer}", (1, 3, Cout, 1, 1), 1.0, 7))
                wr = hip.modulate_weights(Wr, cu(1.0 + weights.det_uniform("rp.sr", (B, Cout), 0.3, 8)), Cout, B, 3, Cout, 1,
                                          1.0 / math.sqrt(Cout), False, False)
                part = torch.zeros(Cout // 64, B, 3, HW, device=DEV)
                out_p = hip.modconv1x1_planes(cur_p, wm_s, Cout, HW, "planes", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr,
                                              rgb_part=part)
                ref = hip.modconv1x1(cur_x, wm_x, Cout, epilogue=1, noise=nz, noise_w=nw, bias=bias)
                got = hip.from_planes(out_p, H, H)
                assert bool(torch.isfinite(got).all())
                assert maxdiff(got, ref) < 4e-6 * float(ref.abs().max()), layer
                # every block of 128 pixels chose its own exponent (from the maximum of ITS input tile): the stored values stay
                # below 2^15, and the block's largest within 2^10 of it (sqrt(2) sqrt(512) max|in| + c0 against what a
                # unit-norm row really does)
                mo = got.abs().amax(dim=1).reshape(B, HW // 128, 128).amax(dim=2)
                st = (mo.cpu().double() * 2.0 ** (-out_p.cips3d_exp.cpu().double()))
                assert bool(((st < 2.0 ** 15) & (st > 2.0 ** 5)).all()), st
                assert len(torch.unique(out_p.cips3d_exp)) >= 1
                outs += [got, part]
                cur_p, cur_x = out_p, ref
            else:
                o32 = hip.modconv1x1_planes(cur_p, wm_s, Cout, HW, "fp32")
                ref = hip.modconv1x1(cur_x, wm_x, Cout, epilogue=0)
                assert maxdiff(o32.view_as(ref), ref) < 4e-6 * float(ref.abs().max())
                assert torch.equal(hip.amax_value(hip.amax_of(o32, measure=False)), o32.abs().amax(dim=(1, 2)))
                outs.append(o32)
        return outs

    base, scaled = run(1.0), run(f)
    for i, (a, b) in enumerate(zip(base, scaled)):
        assert torch.equal(a * f, b), i


@pytest.mark.parametrize("k", SCALES)
@pytest.mark.parametrize("C,H,B", [(256, 32, 1), (128, 32, 2), (64, 64, 1), (32, 64, 2)])
def test_fused_stage_keeps_fp32s_range(C, H, B, k):
    """cips3d_fused_up_conv[_next] in split mode with y_lo, both noise strengths and both biases 2^k times the usual (the ToRGB
    bias and the skip image as well): out2, rgb and the chained y_next are bit for bit 2^k times the unscaled stage's and
    agree with the fp32-MFMA instantiation."""
    chains = hip.fused_up_conv_chains(C)
    lib = _lib.load()
    f = 2.0 ** k

    def mod(W, s, flags):
        out = torch.empty(B * W.shape[0] * C, device=DEV)
        _lib.check(lib.cips3d_modulate_weights(W.data_ptr(), s.data_ptr(), C, out.data_ptr(), B, W.shape[0], C, 1,
                                               1.0 / math.sqrt(C), flags, torch.cuda.current_stream().cuda_stream), "mod")
        return out

    y = cu(weights.det_normal("rf.y", (B, C, H, H), 1.0, C))
    if B > 1:
        y[1] *= 2.0 ** -7
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    fir = cu(k1.outer(k1) / 16.0)
    n1 = cu(weights.det_normal("rf.n1", (1, 1, 2 * H, 2 * H), 1.0, 2))
    n2 = cu(weights.det_normal("rf.n2", (1, 1, 2 * H, 2 * H), 1.0, 3))
    b1, b2 = cu(weights.det_uniform("rf.b1", (C,), 0.2, 4)), cu(weights.det_uniform("rf.b2", (C,), 0.2, 5))
    W2, Wn, Wr = (cu(weights.det_normal(f"rf.{n}", shp, 1.0, 6)) for n, shp in (("W2", (C, C)), ("Wn", (C // 2, C)), ("Wr", (3, C))))
    s2, sn, sr = (cu(1.0 + weights.det_uniform(f"rf.s{i}", (B, C), 0.3, 7)) for i in range(3))
    wmr = mod(Wr, sr, 0)
    brgb = cu(weights.det_uniform("rf.brgb", (3,), 0.1, 8))
    skip = cu(weights.det_normal("rf.skip", (B, 3, H, H), 1.0, 9))

    def run(fac, split):
        f16 = hip.MOD_SPLIT16 if split else 0
        wm2 = mod(W2, s2, hip.MOD_DEMODULATE | hip.MOD_PACKED | f16)
        wmn = mod(Wn, sn, hip.MOD_DEMODULATE | hip.MOD_PACKED | hip.MOD_CHAINED | f16) if chains else None
        nw1, nw2 = torch.full((1,), 0.3 * fac, device=DEV), torch.full((1,), -0.2 * fac, device=DEV)
        return hip.fused_up_conv(y * fac, fir, n1, nw1, b1 * fac, wm2, n2, nw2, b2 * fac, wmr, brgb * fac, skip * fac, skip_up=True,
                                 wm_next=wmn, split=split)

    base, scaled, exact = run(1.0, True), run(f, True), run(f, False)
    for name, a, b_, c in zip(("out2", "rgb", "y_next"), base, scaled, exact):
        assert bool(torch.isfinite(b_).all()), name
        assert torch.equal(a * f, b_), name
        for smp in range(B):
            d, rng = float((c[smp] - b_[smp]).abs().max()), float(c[smp].abs().max())
            assert d < 2e-6 * rng, (name, smp, d, rng)
    if chains:      # the chained y_next carries its recorded maximum for the next stage
        assert torch.equal(hip.amax_value(hip.amax_of(scaled[2], measure=False)), scaled[2].abs().amax(dim=(1, 2, 3)))


def _scaled_generator(res, k, seed=3):
    """A generator whose decoder activations are ~2^k: every StyledConv's noise strength and bias (and every ToRGB bias) times
    2^k -- with random-init weights the decoder's activations are otherwise O(1) whatever the checkpoint-independent inputs."""
    cfg = configs.ffhq_G_cfg(res, 2)
    G = pkg.build_generator(cfg, DEV, seed=seed)
    f = 2.0 ** k
    with torch.no_grad():
        for m in [G.decoder.conv1] + list(G.decoder.convs):
            m.noise.weight.fill_(0.3 * f)
            m.activate.bias.copy_(cu(weights.det_uniform("rg.gb", tuple(m.activate.bias.shape), 0.5, 1)) * f)
        for m in [G.decoder.to_rgb1] + list(G.decoder.to_rgbs):
            m.bias.mul_(f)
    return cfg, G


@pytest.mark.parametrize("k", [17, -14, 0])
def test_one_call_forward_keeps_fp32s_range(k):
    """The planned forward at 256^2 (planes run + chained fused stages, every split site of the default precision) on a decoder
    whose activations are ~2^k: finite, as close to an fp64 evaluation as the fp32 oracle (= the reference's arithmetic) is,
    and equal to the fp32_exact decoder to the usual few 1e-6 of the range."""
    cfg, G = _scaled_generator(256, k)
    zs, nb, _ = weights.synth_inputs(cfg, seed=4)
    loc = torch.tensor([[0.2, 0.05]])
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=cu(loc))
    ncfg = dict(N_samples=12, perturb=False, static_viewdirs=False)
    kw = dict(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[cu(b) for b in nb], nerf_cfg=ncfg)
    a = G(**kw)["rgb"].clone()
    plan = list(G._plans.values())[0]
    assert plan.ranged and plan.plan.range_ws
    assert bool(torch.isfinite(a).all())
    rng = float(a.abs().max())
    assert rng > 0.05 * 2.0 ** max(k, 0)
    G.set_decoder_precision("fp32_exact")
    b = G(**kw)["rgb"].clone()
    assert not list(G._plans.values())[0].ranged
    d = maxdiff(a, b)
    print(f"2^{k}: split vs fp32_exact decoder {d:.2e} on range {rng:.2e}")
    assert d < 2e-5 * rng
    sd = {kk: v.detach().cpu() for kk, v in G.state_dict().items()}
    cam = [t.cpu() for t in (e, f, n, fa)]
    outs = {}
    for dt in (torch.float32, torch.float64):
        sdd = {kk: (v.to(dt) if v.is_floating_point() else v) for kk, v in sd.items()}
        outs[dt] = O.generator_forward(sdd, cfg, [z.to(dt) for z in zs], cam[0].to(dt), cam[1].to(dt), 64, cam[2].to(dt), cam[3].to(dt),
                                       ncfg, [t.to(dt) for t in nb])["rgb"]
    e_hip = float((a.cpu().double() - outs[torch.float64]).abs().max())
    e_32 = float((outs[torch.float32].double() - outs[torch.float64]).abs().max())
    print(f"2^{k}: |hip - fp64| {e_hip:.2e}   |fp32 oracle - fp64| {e_32:.2e}")
    assert e_hip <= 2.0 * e_32 + 2e-6 * rng


def test_one_call_forward_with_caller_noise_measures_its_bound():
    """Caller-supplied noise maps far outside N(0,1)'s usual reach (|n| up to 1e4) enter the layers' bound constants through
    their measured maximum; the planned forward stays finite and equal to the fp32_exact decoder."""
    cfg, G = _scaled_generator(256, 0, seed=5)
    zs, nb, _ = weights.synth_inputs(cfg, seed=6)
    nb = [cu(b) for b in nb]
    nb[1][0, 0, 5, 7] = 1.0e4
    nb[4][0, 0, 9, 1] = -3.0e4
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=torch.tensor([[0.0, 0.1]], device=DEV))
    kw = dict(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=nb,
              nerf_cfg=dict(N_samples=12, perturb=False, static_viewdirs=False))
    a = G(**kw)["rgb"].clone()
    plan = list(G._plans.values())[0]
    assert plan._noise_bound(nb, False) == 3.0e4
    nb[4][0, 0, 9, 1] = -6.0e4                          # an in-place change is seen (version counter)
    assert plan._noise_bound(nb, False) == 6.0e4
    nb[4][0, 0, 9, 1] = -3.0e4
    G.set_decoder_precision("fp32_exact")
    b = G(**kw)["rgb"].clone()
    assert bool(torch.isfinite(a).all())
    assert maxdiff(a, b) < 2e-5 * float(b.abs().max())


@pytest.mark.parametrize("k", [17, -14])
def test_per_op_decoder_is_power_of_two_equivariant(k):
    """The per-op route (Decoder.forward: stand-alone GEMMs that split in registers, FIR + activation, ToRGB): features, noise
    strengths and biases times 2^k give the image times 2^k bit for bit -- every GEMM took its exponent from the measured
    maximum of its input (attached by the producing kernel or measured on demand)."""
    cfg = configs.ffhq_G_cfg(256, 2)
    G = pkg.build_generator(cfg, DEV, seed=2)
    dec = G.decoder
    B = 2
    feats = cu(weights.det_normal("rg.f", (B, 256, 16, 16), 0.5, 1))
    styles = cu(weights.det_normal("rg.st", (B, dec.n_latent, dec.style_dim), 1.0, 2))
    sizes = [16]
    cur = 16
    for i in range(dec.log_in_size + 1, dec.log_size + 1):
        if 2 ** i in dec.upsample_list:
            cur *= 2
        sizes += [cur, cur]
    noise = [cu(weights.det_normal(f"rg.nz{i}", (1, 1, s_, s_), 1.0, 3)) for i, s_ in enumerate(sizes)]

    def run(fac):
        with torch.no_grad():
            for m in [dec.conv1] + list(dec.convs):
                m.noise.weight.fill_(0.25 * fac)
                m.activate.bias.copy_(cu(weights.det_uniform("rg.pb", tuple(m.activate.bias.shape), 0.5, 4)) * fac)
            for i, m in enumerate([dec.to_rgb1] + list(dec.to_rgbs)):
                m.bias.copy_(cu(weights.det_uniform(f"rg.tb{i}", tuple(m.bias.shape), 0.2, 5)) * fac)
            return dec(feats * fac, styles, noise=noise).clone()

    base, scaled = run(1.0), run(2.0 ** k)
    assert bool(torch.isfinite(scaled).all())
    assert torch.equal(base * 2.0 ** k, scaled)
