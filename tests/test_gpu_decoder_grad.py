"""GPU tests of the one-call decoder backward's building blocks and of the call itself (BASELINE config 5;
the reference gets this backward from one `loss.backward()`, /root/reference/exp/cips3d/models/projector_v10.py:1203-1209).
Every kernel is compared with an fp64 evaluation of the same formula; bars are stated relative to what the fp32 evaluation of
that formula itself misses against fp64."""
import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


# (2144 pixels: chunks of 9 steps of 32 and a last one of 4 -- step counts that are no multiple of the register ring; 32: one step)
@pytest.mark.parametrize("B,M,K,P", [(2, 512, 512, 4096), (1, 64, 32, 65536), (2, 32, 32, 1024), (3, 96, 160, 2048), (1, 64, 64, 2144),
                                     (2, 128, 64, 2144), (1, 32, 32, 32)])
@pytest.mark.parametrize("sa,sb", [(1.0, 1.0), (2.0 ** -30, 2.0 ** 17), (2.0 ** 20, 2.0 ** -24)])
def test_split_wgrad_is_as_accurate_as_fp32_at_any_magnitude(B, M, K, P, sa, sb):
    """dwm[b] = dy[b] x[b]^T over the pixels on split-fp16 products: gradients of any magnitude (the scale comes from the
    measured maxima) within 1.5x of what the fp32-MFMA kernel misses against fp64, no inf / nan; accumulate mode adds."""
    g = torch.Generator(device=DEV).manual_seed(M + K + B)
    dy = torch.randn(B, M, P, device=DEV, generator=g) * sa
    x = torch.randn(B, K, P, device=DEV, generator=g) * sb
    x[:, :, ::7] *= 1e-3                                   # a wide spread inside one tensor
    ref = torch.einsum("bmp,bkp->bmk", dy.double().cpu(), x.double().cpu())
    f32 = hip.gemm_wgrad(dy, x)
    sp = hip.gemm_wgrad_split(dy, x, hip.absmax(dy), hip.absmax(x))
    assert bool(torch.isfinite(sp).all())
    e32, esp = _rel(f32, ref), _rel(sp, ref)
    assert esp <= 1.5 * e32 + 1e-7, (esp, e32)
    acc = sp.clone()
    hip.gemm_wgrad_split(dy, x, hip.absmax(dy), hip.absmax(x), out=acc)
    assert _rel(acc, 2 * ref) <= 1.5 * e32 + 1e-7


@pytest.mark.parametrize("B,Cin,Cout,S,split,rgb,per_sample_noise", [
    (2, 512, 512, 64, True, True, False), (2, 512, 512, 64, False, False, False), (1, 256, 512, 32, True, True, True),
    (2, 128, 256, 32, True, False, True), (2, 64, 128, 64, True, True, False), (3, 32, 64, 16, True, True, True),
    (2, 32, 32, 48, False, True, False)])
@pytest.mark.parametrize("gs", [1.0, 2.0 ** -22])
def test_dgrad_with_activation_backward_epilogue(B, Cin, Cout, S, split, rgb, per_sample_noise, gs):
    """cips3d_modconv1x1_actbwd = data-gradient GEMM + ToRGB's contribution + leaky-ReLU derivative + the row reductions
    (bias, noise weight, ToRGB weight gradients), against an fp64 evaluation; gradients 2^-22 small go through the split
    mode unharmed (measured operand scale).  Here Cin = channels of g (the layer's outputs), Cout = channels of y."""
    g_ = torch.Generator(device=DEV).manual_seed(Cin + Cout + S)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g_)
    g = rn(B, Cin, S, S) * gs
    wm = rn(B, Cin, Cout) / Cout ** 0.5                      # the forward's modulated weights [B, out = Cin here, in = Cout]
    y = rn(B, Cout, S, S)
    noise = rn(B if per_sample_noise else 1, 1, S, S)
    rgb_w = rn(B, 3, Cout) / Cout ** 0.5 if rgb else None
    drgb = rn(B, 3, S, S) * gs if rgb else None
    packed_t = hip.pack_weights(wm, transpose=True, split=split)
    d_bias, d_nw = torch.zeros(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    d_rgb_w = torch.zeros(B, 3, Cout, device=DEV) if rgb else None
    dpre, amax = hip.modconv1x1_actbwd(g, packed_t, y, noise=noise, rgb_w=rgb_w, drgb=drgb, split=split,
                                       g_amax=hip.absmax(g) if split else None, d_bias=d_bias, d_noise_w=d_nw, d_rgb_w=d_rgb_w)
    D = lambda t: t.double().cpu()
    gy = torch.einsum("bio,bihw->bohw", D(wm), D(g))
    if rgb:
        gy = gy + torch.einsum("bco,bchw->bohw", D(rgb_w), D(drgb))
    ref = gy * torch.where(D(y) > 0, 2 ** 0.5, 0.2 * 2 ** 0.5)
    tol = 3e-6 if split else 2e-5                            # (fp32 MFMA: plain fp32 accumulation over Cin)
    assert _rel(dpre, ref) < tol
    assert abs(hip.amax_value(amax).max().item() / float(ref.abs().max()) - 1) < 1e-5
    assert _rel(d_bias, ref.sum((0, 2, 3))) < 2e-5 * max(1.0, float(ref.abs().sum((0, 2, 3)).max() / ref.sum((0, 2, 3)).abs().max()))
    nref = (ref * D(noise)).sum((0, 2, 3))
    assert float((D(d_nw) - nref).abs().max()) < 1e-5 * float((ref.abs() * D(noise).abs()).sum((0, 2, 3)).max())
    if rgb:
        wref = torch.einsum("bchw,bohw->bco", D(drgb), D(y))
        assert _rel(d_rgb_w, wref) < 2e-5


def _decoder_case(cfg_fn, res, S0, B, seed, precision="fp32", per_sample_noise=False):
    from cips_3dplusplus_amd import configs
    cfg = cfg_fn()
    G = pkg.build_generator(cfg, DEV, seed=seed)
    dec = G.decoder
    dec.set_precision(precision)
    g = torch.Generator(device=DEV).manual_seed(seed)
    with torch.no_grad():
        for p in dec.parameters():                      # noise weights / biases start at zero: make every gradient path live
            if p.abs().max() == 0:
                p.copy_(0.1 * torch.randn(p.shape, device=DEV, generator=g))
    feats = torch.randn(B, cfg["decoder_cfg"]["in_channel"], S0, S0, device=DEV, generator=g) * 0.5
    styles = torch.randn(B, dec.n_latent, dec.style_dim, device=DEV, generator=g)
    noise = [torch.randn(B if per_sample_noise else 1, *b.shape[1:], device=DEV, generator=g) for b in G.create_noise_bufs(S0, DEV)]
    return dec, feats, styles, noise


def _grads(dec, feats, styles, noise, one_call, target):
    from cips_3dplusplus_amd import autograd as AG
    from cips_3dplusplus_amd import decoder_grad
    AG.ONE_CALL_DECODER = one_call
    params = decoder_grad.parameters_of(dec)
    for p in dec.parameters():
        p.requires_grad_(True)
        p.grad = None
    f, s = feats.clone().requires_grad_(True), styles.clone().requires_grad_(True)
    rgb = AG.decoder_forward(dec, f, s, noise)
    loss = ((rgb - target) ** 2).mean() * 3.0
    loss.backward()
    AG.ONE_CALL_DECODER = True
    return rgb.detach(), f.grad, s.grad, [p.grad for p in params]


@pytest.mark.parametrize("name,res,S0,B", [("tiny", 32, 8, 2), ("tiny", 32, 16, 3), ("config5", 256, 64, 2), ("r1024", 1024, 64, 1),
                                           ("config1_no_upsampling", 64, 64, 2)])
def test_one_call_decoder_matches_the_per_op_route(name, res, S0, B):
    """Decoder.forward + backward as one node (csrc/decoder_grad.hip) against the chain of per-op nodes (autograd.py, itself
    pinned to the reference's gradients by tests/golden/backward.npz and config5.npz): image, feature / style gradients and
    every parameter gradient.  The two routes share no backward kernel except the style-table one."""
    from cips_3dplusplus_amd import configs
    if name == "tiny":
        cfg_fn = lambda: configs.tiny_G_cfg(32, 2, 1)           # noqa: E731
    else:
        cfg_fn = lambda: configs.ffhq_G_cfg(res, 2)             # noqa: E731
    dec, feats, styles, noise = _decoder_case(cfg_fn, res, S0, B, seed=5)
    from cips_3dplusplus_amd import decoder_grad
    assert decoder_grad.plan_for(dec, B, S0, S0, feats.device) is not None, "the one-call plan must cover this decoder"
    with torch.no_grad():
        from cips_3dplusplus_amd import autograd as AG
        shape = AG.decoder_forward(dec, feats, styles, noise).shape
    target = torch.randn(*shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    ref = _grads(dec, feats, styles, noise, False, target)
    one = _grads(dec, feats, styles, noise, True, target)
    assert _rel(one[0], ref[0].double().cpu()) < 2e-5
    names = ["d_features", "d_styles"]
    for tag, a, b in zip(names, one[1:3], ref[1:3]):
        assert a is not None and _rel(a, b.double().cpu()) < 5e-4, tag
    pnames = [n for n, _ in dec.named_parameters()]
    by_id = {id(p): n for n, p in dec.named_parameters()}
    for p, a, b in zip(decoder_grad.parameters_of(dec), one[3], ref[3]):
        assert a is not None and b is not None, by_id[id(p)]
        assert a.shape == p.shape, by_id[id(p)]
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 1e-3 * scale + 1e-9, (by_id[id(p)], err, scale)
    # nothing the node returned aliases the plan: a second backward must not change the first one's gradients
    keep = [g.clone() for g in one[3]]
    _grads(dec, feats, styles * 0.5, noise, True, target)
    assert all(torch.equal(a, b) for a, b in zip(keep, one[3]))


@pytest.mark.parametrize("precision,per_sample_noise,B", [("fp32_exact", False, 2), ("fp32", True, 2), ("fp32_exact", True, 4)])
def test_one_call_decoder_other_modes(precision, per_sample_noise, B):
    """The one-call node on the fp32 matrix instruction (`fp32_exact`: fp32 GEMMs and the fp32 weight-gradient kernel), with
    per-sample noise maps, at batch 4 -- against the per-op route; the bf16 modes and batches above 4 are handed to that route."""
    from cips_3dplusplus_amd import configs, decoder_grad
    dec, feats, styles, noise = _decoder_case(lambda: configs.tiny_G_cfg(32, 2, 1), 32, 8, B, seed=9, precision=precision,
                                              per_sample_noise=per_sample_noise)
    assert decoder_grad.plan_for(dec, B, 8, 8, feats.device) is not None
    target = torch.randn(B, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    ref = _grads(dec, feats, styles, noise, False, target)
    one = _grads(dec, feats, styles, noise, True, target)
    assert _rel(one[0], ref[0].double().cpu()) < 2e-5
    for a, b in list(zip(one[1:3], ref[1:3])) + list(zip(one[3], ref[3])):
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-9
    # two graphs alive at once: the plan keeps ONE forward's activations -- the older graph's backward fails loudly
    from cips_3dplusplus_amd import autograd as AG
    s1 = styles.clone().requires_grad_(True)
    r1 = AG.decoder_forward(dec, feats, s1, noise)
    r2 = AG.decoder_forward(dec, feats, styles.clone().requires_grad_(True), noise)
    with pytest.raises(RuntimeError, match="ran forward again"):
        r1.sum().backward()
    r2.sum().backward()
    import copy
    twin = copy.deepcopy(dec)                      # (the projector's first step) the copy plans for its own parameters
    two = _grads(twin, feats, styles, noise, True, target)
    assert torch.equal(two[0], one[0]) and decoder_grad.plan_for(twin, B, 8, 8, feats.device) is not decoder_grad.plan_for(dec, B, 8, 8, feats.device)
    dec.set_precision("bf16")
    assert decoder_grad.plan_for(dec, B, 8, 8, feats.device) is None            # key changed: re-planned, refused
    dec.set_precision("fp32")
    assert decoder_grad.plan_for(dec, 5, 8, 8, feats.device) is None


@pytest.mark.parametrize("B,C,H,W,per_sample", [(2, 64, 16, 16, False), (3, 32, 8, 12, True), (1, 256, 64, 64, False)])
def test_up2_fir_act_records_its_maximum(B, C, H, W, per_sample):
    """cips3d_up2_fir_act(out_amax): the per-sample maximum of the up-sampled, activated tensor, recorded by the kernel that
    writes it (grid-stride loop: a thread may cross samples) -- what cips3d_absmax would measure in a pass of its own."""
    g = torch.Generator(device=DEV).manual_seed(B * C + H)
    y_lo = torch.randn(B, C, H, W, device=DEV, generator=g) * torch.tensor([1.0, 30.0, 0.01][:B], device=DEV).view(B, 1, 1, 1)
    fir = torch.tensor([1.0, 3.0, 3.0, 1.0], device=DEV)
    fir = (fir[:, None] * fir[None, :] / 64.0 * 4.0).contiguous()
    noise = torch.randn(B if per_sample else 1, 1, 2 * H, 2 * W, device=DEV, generator=g)
    nw, bias = torch.tensor([0.3], device=DEV), torch.randn(C, device=DEV, generator=g) * 0.1
    out = hip.up2_fir_act(y_lo, fir, noise, nw, bias, track=True)
    ref = hip.up2_fir_act(y_lo, fir, noise, nw, bias)
    assert torch.equal(out, ref)
    got = hip.amax_value(hip.amax_of(out, measure=False))
    assert torch.equal(got, ref.abs().amax(dim=(1, 2, 3)))
