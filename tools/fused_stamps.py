"""Per-phase cycle sums of the fused up-sampling stage kernels (diagnostic build: CIPS3D_HIPCC_FLAGS=-DCIPS3D_FUSED_STAMPS, set on
the GPU box too): mean cycles per workgroup (wave 0) of each phase, for C = 32 / 64 / 128 / 256 of the default forward."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import _lib as L, configs
from cips_3dplusplus_amd.camera import Camera
G = pkg.build_generator(configs.ffhq_G_cfg(1024, 2), "cuda", seed=0)
cam, focal, near, far, _ = Camera.generate_camera_params(64, "cuda", locations=torch.zeros(1, 2), fov_ang=6, dist_radius=0.12)
zs = [torch.randn(1, 256, device="cuda"), torch.randn(1, 256, device="cuda")]
run = lambda: G(zs=zs, cam_poses=cam, focals=focal, img_size=64, near=near, far=far, nerf_cfg={"N_samples": 24, "perturb": True, "static_viewdirs": False})
names = ["operand requests, noise staged, barrier", "stage-0 FIR + act + split -> LDS, barrier", "K loop (conv2 MFMAs, next patches)",
         "conv2 epilogue (+ToRGB partials)", "chained GEMM MFMAs", "exchange through LDS, y_next store", "rgb: skip FIR, store"]
with torch.no_grad():
    for _ in range(5): run()
    raw = ctypes.CDLL(L.LIB_PATH)
    buf = (ctypes.c_ulonglong * 32)()
    raw.cips3d_debug_read_fused_stamps(buf)
    for _ in range(10): run()
    raw.cips3d_debug_read_fused_stamps(buf)
for sl, C in enumerate((32, 64, 128, 256)):
    n = buf[sl * 8 + 7]
    if not n: continue
    tot = sum(buf[sl * 8 + i] for i in range(7))
    print(f"C = {C}: {n // 10} workgroups sampled per launch (every 32nd), {tot / n:.0f} cycles per workgroup")
    for i, nm in enumerate(names):
        v = buf[sl * 8 + i]
        print(f"    {nm:44s} {v / n:8.0f}  {100.0 * v / tot:5.1f} %")
