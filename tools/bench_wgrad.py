"""Timing of the weight-gradient GEMMs (fp32 MFMA vs split-fp16) at the inversion step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import hip
dev = "cuda"
for (B, M, K, P) in ((2, 512, 512, 4096), (2, 512, 256, 4096), (2, 256, 512, 4096), (2, 256, 256, 16384), (2, 128, 256, 16384), (2, 128, 128, 65536), (2, 64, 128, 65536), (2, 32, 32, 65536)):
    dy = torch.randn(B, M, P, device=dev) * 1e-4
    x = torch.randn(B, K, P, device=dev)
    a, b = hip.absmax(dy), hip.absmax(x)
    out = torch.zeros(B, M, K, device=dev)
    res = {}
    for name, fn in (("fp32", lambda: hip.gemm_wgrad(dy, x)), ("split", lambda: hip.gemm_wgrad_split(dy, x, a, b)),
                     ("split_acc", lambda: hip.gemm_wgrad_split(dy, x, a, b, out=out))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    print((B, M, K, P), {k: round(v, 1) for k, v in res.items()}, "us;  GF", 2 * B * M * K * P / 1e9)
