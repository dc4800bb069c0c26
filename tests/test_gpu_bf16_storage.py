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
    """256^2 generator (both up-sampling stages fused and chained): the storage mode against the oracle with the same
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
    assert d < 2e-2 * scale and _psnr(r16s.cpu(), ref) > 55.0
    assert p_s > 35.0
    G.set_decoder_precision("fp32")
    assert torch.equal(G(**kw)["rgb"], r32)
