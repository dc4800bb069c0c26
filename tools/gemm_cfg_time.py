"""Time the 512->512 @64^2 stand-alone GEMM (batch 1) in its three arithmetic modes and, through CIPS3D_GEMM_CFG, other tilings.
    [CIPS3D_GEMM_CFG=n] python tools/gemm_cfg_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import hip
dev = "cuda"
C, H = 512, 64
x = torch.randn(1, C, H, H, device=dev)
W = torch.randn(1, C, C, 1, 1, device=dev)
s = torch.rand(1, C, device=dev) + 0.5
nz = torch.randn(1, 1, H, H, device=dev)
nw = torch.full((1,), 0.1, device=dev)
bias = torch.zeros(C, device=dev)
out = torch.empty(1, C, H, H, device=dev)
for name, kw in (("fp32 MFMA", {}), ("bf16", {"bf16": True}), ("split-fp16", {"split": True})):
    wm = hip.modulate_weights(W, s, C, 1, C, C, 1, C ** -0.5, True, True, split=kw.get("split", False))
    g = torch.cuda.CUDAGraph()
    fn = lambda: [hip.modconv1x1(x, wm, C, epilogue=1, noise=nz, noise_w=nw, bias=bias, out=out, **kw) for _ in range(20)]
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"cfg {os.environ.get('CIPS3D_GEMM_CFG', '0')}: {name:10s} {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us per launch (20 back-to-back launches per graph)")
