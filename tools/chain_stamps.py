"""Per-phase cycle sums of chain_gemm_kernel (diagnostic build: CIPS3D_HIPCC_FLAGS=-DCIPS3D_CHAIN_STAMPS, set on the GPU box too).
Runs the default forward a few times and prints the mean cycles per workgroup of: launch -> first stage landed, main loop,
epilogue arithmetic + store issue, store acknowledgement."""
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
with torch.no_grad():
    for _ in range(5): run()
    raw = ctypes.CDLL(L.LIB_PATH)
    buf = (ctypes.c_ulonglong * 8)()
    raw.cips3d_debug_read_chain_stamps(buf)
    for _ in range(10): run()
    raw.cips3d_debug_read_chain_stamps(buf)
n = buf[7]
for nm, v in zip(["launch -> first stage landed", "main loop (8 K stages)", "epilogue + store issue", "store acknowledgement"], buf[:4]):
    print(f"  {nm:32s} {v / n:9.0f} cycles per workgroup")
for nm, v in zip(["  in loop: vmcnt wait + barrier", "  in loop: DMA issue", "  in loop: fragment reads + MFMA issue"], buf[4:7]):
    print(f"  {nm:42s} {v / n / 8:9.0f} cycles per stage (wave 0)")
print(f"  ({n} workgroups sampled)")
