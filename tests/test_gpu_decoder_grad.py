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


@pytest.mark.parametrize("B,M,K,P", [(2, 512, 512, 4096), (1, 64, 32, 65536), (2, 32, 32, 1024), (3, 96, 160, 2048)])
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
