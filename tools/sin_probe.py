"""Diagnostic (-DCIPS3D_CLOCK build): accuracy of the SIREN sine forms against fp64 over the argument range of the path."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import _lib as L
L.load()
raw = ctypes.CDLL(L.LIB_PATH)
for lo, hi in ((-4, 4), (-40, 40), (-100, 100), (-1000, 1000)):
    x = (torch.rand(1 << 22, device="cuda", dtype=torch.float64) * (hi - lo) + lo).float()
    ya, yh = torch.empty_like(x), torch.empty_like(x)
    raw.cips3d_debug_sin(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(ya.data_ptr()), ctypes.c_void_p(yh.data_ptr()),
                         ctypes.c_int(x.numel()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    ref = torch.sin(x.double())
    print(f"x in [{lo},{hi}]: max |sin_accurate - fp64| = {float((ya.double() - ref).abs().max()):.3e}   "
          f"max |v_sin(reduced) - fp64| = {float((yh.double() - ref).abs().max()):.3e}   "
          f"mean {float((yh.double() - ref).abs().mean()):.3e} vs {float((ya.double() - ref).abs().mean()):.3e}")
