"""Timing of the stand-alone `upfirdn2d` op (csrc/upfirdn2d.hip) on the shapes the path uses it with when a caller goes
through the op-level API instead of the fused stages: the 2x FIR up-sampling of the RGB skip (Upsample, models/model_v3.py
blur kernel [1, 3, 3, 1]), the blur after a transposed 3x3 conv (pad (1, 1)), a 2x down-sampling (the discriminator-side
use of the op), priced against HBM: the op reads every input element and writes every output element once.  Calls are timed back to back through the
Python wrapper (about 20 us of host time per call): only the cases of >= 250 MB are GPU-bound here, the rocprofv3 summary of
this script (profiles/) has the kernel times of the small ones.

    python tools/bench_upfirdn2d.py
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import op

HBM_PEAK_GBS = 8000.0
dev = "cuda"
k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
k = (k1[None, :] * k1[:, None]) / k1.sum() ** 2
k = k.to(dev)
for name, shape, kern, up, down, pad in (
        ("skip up 3ch 512->1024", (1, 3, 512, 512), k * 4, 2, 1, (2, 1)),
        ("skip up 3ch 64->128", (1, 3, 64, 64), k * 4, 2, 1, (2, 1)),
        ("up 64ch 256->512", (1, 64, 256, 256), k * 4, 2, 1, (2, 1)),
        ("blur 32ch 1025->1024", (1, 32, 1025, 1025), k * 4, 1, 1, (1, 1)),
        ("blur 256ch 129->128", (1, 256, 129, 129), k * 4, 1, 1, (1, 1)),
        ("down 64ch 512->256", (1, 64, 512, 512), k, 1, 2, (1, 1)),
        ("batch 4 up 32ch 512->1024", (4, 32, 512, 512), k * 4, 2, 1, (2, 1)),
        ("batch 8 up 64ch 256->512", (8, 64, 256, 256), k * 4, 2, 1, (2, 1)),
        ("batch 4 blur 32ch 1025->1024", (4, 32, 1025, 1025), k * 4, 1, 1, (1, 1)),
        ("batch 8 down 64ch 512->256", (8, 64, 512, 512), k, 1, 2, (1, 1))):
    x = torch.randn(*shape, device=dev)
    y = op.upfirdn2d(x, kern, up=up, down=down, pad=pad)
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = op.upfirdn2d(x, kern, up=up, down=down, pad=pad)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 4.0 * (x.numel() + y.numel())
    print(json.dumps({"case": name, "in": list(x.shape), "out": list(y.shape), "ms_per_call_incl_launch": round(ms, 4),
                      "algorithmic_MB": round(nbytes / 1e6, 2), "GBs": round(nbytes / ms / 1e6, 1),
                      "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 3)}))
