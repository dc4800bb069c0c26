"""Timing of the LDS-tiled MFMA 3x3 modulated conv (csrc/conv3x3.hip) against the fp32 MFMA peak, and of the direct
kernel it replaces (cips3d_modconv_kxk + upfirdn2d + noise_bias_act) on the same shapes.

    python tools/bench_conv3x3.py
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd.decoder as dec
from cips_3dplusplus_amd import hip, op

PEAK = 157.3
dev = "cuda"
rows = []
for cin, cout, H, up in ((512, 512, 64, False), (256, 256, 128, False), (128, 128, 256, False), (64, 64, 512, False),
                         (32, 32, 1024, False), (512, 256, 64, True), (128, 64, 256, True), (64, 32, 512, True)):
    sc = dec.StyledConv(cin, cout, 3, 512, upsample=up).to(dev).requires_grad_(False)
    sc.noise.weight.data.fill_(0.1)
    x = torch.randn(1, cin, H, H, device=dev)
    style = torch.randn(1, 512, device=dev)
    Ho = 2 * H if up else H
    nz = torch.randn(1, 1, Ho, Ho, device=dev)
    wm = sc.conv.modulated_weight(style, packed=True, flip=up)
    wm_s = sc.conv.modulated_weight(style, packed=True, flip=up, split=True)
    xa = hip.absmax(x)
    wm_plain = sc.conv.modulated_weight(style, packed=False)

    def tiled():
        return hip.modconv3x3(x, wm, cout, up=up, fir=sc.conv.blur.kernel if up else None, epilogue=1, noise=nz,
                              noise_w=sc.noise.weight, bias=sc.activate.bias)

    def tiled_split():
        return hip.modconv3x3(x, wm_s, cout, up=up, fir=sc.conv.blur.kernel if up else None, epilogue=1, noise=nz,
                              noise_w=sc.noise.weight, bias=sc.activate.bias, split=True, x_amax=xa)

    def direct():
        y = hip.modconv_kxk(x, wm_plain, cout, 3, transpose2=up)
        if up:
            y = sc.conv.blur(y)
        return hip.noise_bias_act(y, nz, sc.noise.weight, sc.activate.bias)

    res = {}
    for name, fn, reps in (("tiled", tiled, 20), ("split", tiled_split, 20), ("direct", direct, 3)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps
    flop = 2.0 * 9 * cin * cout * Ho * Ho            # MFMA work of the tiled form (up: the 3x3 correlation runs at the output size)
    rows.append({"shape": f"{cin}->{cout} @{H}^2{' up' if up else ''}", "tiled_ms": res["tiled"], "split_ms": res["split"],
                 "split_frac_of_split_peak": 3 * flop / res["split"] / 1e9 / 2500.0, "direct_ms": res["direct"],
                 "tiled_TFLOPs": flop / res["tiled"] / 1e9, "frac_of_fp32_mfma_peak": flop / res["tiled"] / 1e9 / PEAK,
                 "speedup_vs_direct": res["direct"] / res["tiled"]})
    print(json.dumps(rows[-1]))
