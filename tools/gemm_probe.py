"""Fixed cost vs slope of modconv1x1: sweep K (and the epilogue) at M=512, N=64*64.  Run under rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import hip
dev = "cuda"
M, HW = int(os.environ.get("M", 512)), int(os.environ.get("HW", 4096))
side = int(HW ** 0.5)
for K in (64, 128, 256, 512, 1024):
    x = torch.randn(1, K, side, side, device=dev)
    wm = torch.randn(M * K, device=dev)
    bias = torch.randn(M, device=dev)
    nz = torch.randn(1, 1, side, side, device=dev)
    nw = torch.full((1,), 0.1, device=dev)
    for ep in (0, 1):
        for _ in range(12):
            hip.modconv1x1(x, wm, M, epilogue=ep, noise=nz if ep else None, noise_w=nw if ep else None, bias=bias if ep else None)
    torch.cuda.synchronize()
