"""Bit-exact repeat check of the folded ToRGB partial sums (planes16 and split planes kernels at batch 4, two workgroups per CU for
planes16): usage python tools/fold_repeat.py [runs].  Prints the number of runs whose partial sums differ from the first one."""
import math, sys
import torch
sys.path.insert(0, ".")
from cips_3dplusplus_amd import hip, weights
DEV = "cuda"
cu = lambda t: t.to(DEV).contiguous()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, C, Cout, H = 4, 512, 512, 64
HW = H * H
x = cu(weights.det_normal("p16.x", (B, C, H, H), 2.0, 1))
scale = 1.0 / math.sqrt(C)
Wt = cu(weights.det_normal("p16.W", (1, Cout, C, 1, 1), 1.0, 3))
s = cu(1.0 + weights.det_uniform("p16.s", (B, C), 0.4, 4))
bias = cu(weights.det_uniform("p16.b", (Cout,), 0.3, 5))
nw = torch.full((1,), 0.2, device=DEV)
nz = cu(weights.det_normal("p16.n", (1, 1, H, H), 1.0, 6))
Wr = cu(weights.det_normal("p16.Wr", (1, 3, Cout, 1, 1), 1.0, 7))
wr = hip.modulate_weights(Wr, cu(1.0 + weights.det_uniform("p16.sr", (B, Cout), 0.3, 8)), Cout, B, 3, Cout, 1, 1.0 / math.sqrt(Cout), False, False)
for name in ("planes16", "planes"):
    if name == "planes16":
        p = hip.to_planes16(x)
        wm = hip.modulate_weights(Wt, s, C, B, Cout, C, 1, scale, True, True, bf16=True)
        f = lambda part: hip.modconv1x1_planes16(p, wm, Cout, HW, "planes16", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr, rgb_part=part)
    else:
        p = hip.to_planes(x)
        wm = hip.modulate_weights(Wt, s, C, B, Cout, C, 1, scale, True, True, split=True)
        f = lambda part: hip.modconv1x1_planes(p, wm, Cout, HW, "planes", epilogue=1, noise=nz, noise_w=nw, bias=bias, rgb_w=wr, rgb_part=part)
    first, bad, worst = None, 0, 0
    for r in range(runs):
        part = torch.full((Cout // 64, B, 3, HW), float("nan"), device=DEV)
        f(part)
        torch.cuda.synchronize()
        if first is None:
            first = part.clone()
        else:
            n = int((part != first).sum())
            bad += n > 0
            worst = max(worst, n)
    print(f"{name}: {bad} of {runs - 1} repeats differ from the first run (worst: {worst} of {first.numel()} partial sums)")
