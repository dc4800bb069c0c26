"""Long modconv1x1 (M=K=512, 512^2 pixels = 137 GFLOP) to read the sustained MFMA rate and clock."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import hip
M = K = 512; side = 512
x = torch.randn(1, K, side, side, device="cuda")
wm = torch.randn(M * K, device="cuda")
for _ in range(6):
    hip.modconv1x1(x, wm, M, epilogue=0)
torch.cuda.synchronize()
