"""Host-side cost of one frame of the multi-view loop (enqueue only) against the GPU time per frame: is config 4 GPU-bound?
    python tools/multiview_host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.multiview import sample_multi_view

dev = "cuda"
G = pkg.build_generator(configs.ffhq_G_cfg(1024, 2), dev, seed=0)
zs = [torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)]
cam_cfg = {"img_size": 64, "fov_ang": configs.FFHQ_CAM_CFG["fov_ang"], "dist_radius": configs.FFHQ_CAM_CFG["dist_radius"]}
nb = G.create_noise_bufs(64, dev)
for n_samples in (128, 8):
    ncfg = {"N_samples": n_samples, "perturb": False, "static_viewdirs": False}
    run = lambda: sample_multi_view(G, cam_cfg, ncfg, zs, N_frames=8, truncation_ratio=0.5, N_samples=n_samples, noise_bufs=nb)  # noqa: E731
    run(); run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"N = {n_samples}: host returns after {t_host / 40 * 1e6:.0f} us per frame; GPU done after {t_all / 40 * 1e6:.0f} us per frame")
