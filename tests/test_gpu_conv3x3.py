"""The LDS-tiled MFMA 3x3 ModulatedConv2d (csrc/conv3x3.hip; reference models/model_v3.py:264-314) against the oracle
(which the reference's own `mc_k3_*` fixtures pin, tests/test_oracle_golden.py): plain and up-sampling branches, ragged
tiles, fused StyledConv epilogue, tap-major weight packing, fallback for shapes the kernel does not tile."""
import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, hip, weights
from conftest import maxdiff
from oracle import path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(t):
    return t.to(DEV).contiguous()


def _dec():
    import cips_3dplusplus_amd.decoder as dec
    return dec


def _conv(cin, cout, up, demod, seed):
    dec = _dec()
    m = dec.ModulatedConv2d(cin, cout, 3, 32, demodulate=demod, upsample=up)
    sd = {k: weights.det_normal(f"c3.{seed}.{k}", v.shape, 1.0, seed) if v.is_floating_point() and "kernel" not in k else v
          for k, v in m.state_dict().items()}
    sd["modulation.bias"] = 1.0 + weights.det_uniform(f"c3.{seed}.mb", (cin,), 0.3, seed)
    m.load_state_dict(sd)
    return m, sd


def test_tap_major_packing_matches_plain_layout():
    """cips3d_modulate_weights(ksq = 9, PACKED [| FLIP]) holds exactly the values of the plain layout, fragment-ordered per tap."""
    B, Cin, Cout = 2, 32, 48
    m, _ = _conv(Cin, Cout, False, True, 1)
    m = m.to(DEV)
    style = cu(weights.det_normal("c3.style", (B, 32), 1.0, 1))
    plain = m.modulated_weight(style, packed=False).view(B, Cout, Cin, 9).cpu()
    for flip in (False, True):
        packed = m.modulated_weight(style, packed=True, flip=flip).view(B, 9, Cout // 16, Cin // 16, 64, 4).cpu()
        # packed[b][t'][ot][kq][lane = (q << 4) | o_lo][j] = plain[b][16 ot + o_lo][16 kq + 4 j + q][t],  t' = 8 - t if flip
        p = packed.view(B, 9, Cout // 16, Cin // 16, 4, 16, 4)                # b, t', ot, kq, q, o_lo, j
        un = p.permute(0, 2, 5, 3, 6, 4, 1).reshape(B, Cout, Cin, 9)          # b, ot, o_lo, kq, j, q, t'
        if flip:
            un = un.flip(-1)
        assert torch.equal(un, plain)


@pytest.mark.parametrize("cin,cout,H,W,up,B,demod", [
    (16, 16, 8, 8, False, 1, True), (32, 48, 12, 20, False, 2, True), (64, 32, 7, 68, False, 1, False),
    (128, 128, 32, 64, False, 1, True), (16, 32, 5, 6, True, 2, True), (64, 64, 16, 16, True, 1, True),
    (128, 64, 9, 40, True, 1, False), (32, 16, 64, 64, True, 1, True)])
def test_modconv3x3_vs_oracle(cin, cout, H, W, up, B, demod):
    m, sd = _conv(cin, cout, up, demod, cin + cout + H)
    x = weights.det_normal("c3.x", (B, cin, H, W), 1.0, H)
    style = weights.det_normal("c3.s", (B, 32), 1.0, W)
    sdp = {"m." + k: v for k, v in sd.items()}
    ref = O.modulated_conv2d(sdp, "m", x, style, demodulate=demod, upsample=up)
    m = m.to(DEV)
    assert m.tiled3x3(H, W)
    y = m(cu(x), cu(style))
    assert y.shape == ref.shape and y.is_contiguous()
    assert maxdiff(y.cpu(), ref) < 3e-5 * max(1.0, float(ref.abs().max())), (cin, cout, H, W, up)


def test_split_tap_pair_packing_matches_plain_layout():
    """cips3d_modulate_weights(ksq = 9, PACKED | SPLIT [| FLIP]): the tap-pair fragments of modconv3x3_split_kernel hold hi + lo of
    2^8 w of the plain layout -- lane quarter q = channels 8 (q & 1) .. + 7 of tap 2 pair + (q >> 1), zeros in the tenth tap."""
    B, Cin, Cout = 2, 32, 48
    m, _ = _conv(Cin, Cout, False, True, 1)
    m = m.to(DEV)
    style = cu(weights.det_normal("c3.style", (B, 32), 1.0, 1))
    plain = m.modulated_weight(style, packed=False).view(B, Cout, Cin, 9).cpu()
    for flip in (False, True):
        buf = m.modulated_weight(style, packed=True, flip=flip, split=True)
        n = B * 10 * Cout * Cin * 2                                    # fp16 elements written (five tap pairs)
        h = buf.view(torch.float16)[:n].view(B, 5, Cout // 16, Cin // 16, 2, 4, 16, 8).float().cpu()   # b, pair, ot, kq, plane, q, o_lo, j
        val = (h[:, :, :, :, 0] + h[:, :, :, :, 1]) / 256.0            # hi + lo of 2^8 w
        # q = 2 h + cg: tap = 2 pair + h, channel = 16 kq + 8 cg + j
        v = val.view(B, 5, Cout // 16, Cin // 16, 2, 2, 16, 8)         # b, pair, ot, kq, h, cg, o_lo, j
        w = v.permute(0, 2, 6, 3, 5, 7, 1, 4).reshape(B, Cout, Cin, 10)    # b, (ot, o_lo), (kq, cg, j), (pair, h) = tap
        assert float(w[..., 9].abs().max()) == 0.0
        got = w[..., :9].flip(-1) if flip else w[..., :9]
        assert maxdiff(got, plain) <= 2.0 ** -21 * float(plain.abs().max())


@pytest.mark.parametrize("cin,cout,H,W,up,B,demod,scale", [
    (16, 16, 8, 8, False, 1, True, 1.0), (32, 48, 12, 20, False, 2, True, 1e-6), (64, 32, 7, 68, False, 1, False, 3e4),
    (128, 128, 32, 64, False, 1, True, 1.0), (16, 32, 5, 6, True, 2, True, 1.0), (64, 64, 16, 16, True, 1, True, 1e5),
    (128, 64, 9, 40, True, 1, False, 1e-5), (32, 16, 64, 64, True, 1, True, 1.0), (512, 512, 16, 16, False, 1, True, 1.0)])
def test_modconv3x3_split_vs_oracle(cin, cout, H, W, up, B, demod, scale):
    """modconv3x3_split_kernel (three fp16 products per fp32 product on v_mfma_f32_16x16x32_f16, two taps per MFMA step) against
    the oracle at the fp32 kernel's bar, on data of any magnitude (the input is split under its measured maximum)."""
    m, sd = _conv(cin, cout, up, demod, cin + cout + H)
    x = weights.det_normal("c3.x", (B, cin, H, W), 1.0, H) * scale
    style = weights.det_normal("c3.s", (B, 32), 1.0, W)
    ref = O.modulated_conv2d({"m." + k: v for k, v in sd.items()}, "m", x, style, demodulate=demod, upsample=up)
    m = m.to(DEV)
    wm = m.modulated_weight(cu(style), packed=True, flip=up, split=True)
    y = hip.modconv3x3(cu(x), wm, cout, up=up, fir=m.blur.kernel if up else None, split=True)
    y32 = m(cu(x), cu(style))
    assert y.shape == ref.shape
    r = max(float(ref.abs().max()), 1e-30)
    assert maxdiff(y.cpu(), ref) < 3e-5 * r, (cin, cout, H, W, up, maxdiff(y.cpu(), ref) / r)
    assert maxdiff(y, y32) < 3e-5 * r


@pytest.mark.parametrize("up,per_sample_noise", [(False, False), (False, True), (True, False), (True, True)])
def test_styled_conv_k3_fused_epilogue_vs_oracle(up, per_sample_noise):
    dec = _dec()
    B, cin, cout, H, W = 2, 32, 64, 10, 12
    sc = dec.StyledConv(cin, cout, 3, 32, upsample=up)
    sd = {k: weights.det_normal(f"sc3.{k}", v.shape, 1.0, 4) if v.is_floating_point() and "kernel" not in k else v
          for k, v in sc.state_dict().items()}
    sd["noise.weight"] = torch.full((1,), 0.3)
    sd["activate.bias"] = weights.det_uniform("sc3.ab", (cout,), 0.3, 4)
    sc.load_state_dict(sd)
    x = weights.det_normal("sc3.x", (B, cin, H, W), 1.0, 5)
    style = weights.det_normal("sc3.s", (B, 32), 1.0, 6)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    nz = weights.det_normal("sc3.n", (B if per_sample_noise else 1, 1, Ho, Wo), 1.0, 7)
    ref = O.styled_conv({"s." + k: v for k, v in sd.items()}, "s", x, style, nz, upsample=up)
    sc = sc.to(DEV)
    assert sc.conv.tiled3x3(H, W)
    for split in (True, False):                 # the default arithmetic (split-fp16 products) and the fp32 matrix instruction
        sc.split = split
        y = sc(cu(x), cu(style), noise=cu(nz))
        assert y.shape == ref.shape and maxdiff(y.cpu(), ref) < 3e-5 * max(1.0, float(ref.abs().max())), split


def test_k3_generator_uses_the_tiled_kernel_and_matches_the_reference(golden):
    """The k = 3 tiny generator of the reference fixture: every StyledConv of its decoder is tiled by the MFMA kernel (so the
    golden comparison in test_gpu_parity.py::test_tiny_generator_golden[h32_d2_k3] exercises it), and shapes the kernel does
    not tile (the reference's Cin = 8 fixtures) still run on the direct kernel."""
    fx = golden("tiny_generator")
    cfg = configs.tiny_G_cfg(32, 2, 3)
    G = pkg.build_generator(cfg, DEV, state_dict=fx.sub("h32_d2_k3.sd."))
    dec = G.decoder
    res = 8
    for conv in [dec.conv1] + list(dec.convs):
        assert conv.conv.kernel_size == 3 and conv.conv.tiled3x3(res, res), conv
        if conv.conv.upsample:
            res *= 2
    assert not _dec().ModulatedConv2d(8, 12, 3, 16).tiled3x3(6, 6)
    assert hip.modconv3x3_supported(16, 16, 4, 4, False) and not hip.modconv3x3_supported(16, 16, 4, 6, False)
    assert hip.modconv3x3_supported(16, 16, 3, 6, True) and not hip.modconv3x3_supported(16, 24, 4, 4, False)
