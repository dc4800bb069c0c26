"""Which torch ops launch device work inside the multi-view frame loop (config 4): torch.profiler over a few sequences.
    python tools/multiview_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.multiview import sample_multi_view

dev = "cuda"
G = pkg.build_generator(configs.ffhq_G_cfg(1024, 2), dev, seed=0)
zs = [torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)]
cam_cfg = {"img_size": 64, "fov_ang": configs.FFHQ_CAM_CFG["fov_ang"], "dist_radius": configs.FFHQ_CAM_CFG["dist_radius"]}
ncfg = {"N_samples": 128, "perturb": False, "static_viewdirs": False}
nb = G.create_noise_bufs(64, dev)
run = lambda: sample_multi_view(G, cam_cfg, ncfg, zs, N_frames=8, truncation_ratio=0.5, N_samples=128, noise_bufs=nb)  # noqa: E731
run(); run()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as p:
    for _ in range(N):
        run()
    torch.cuda.synchronize()
rows = []
for e in p.key_averages(group_by_stack_n=14):
    dt = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if dt > 0:
        where = [f for f in e.stack if "cips_3dplusplus_amd" in f or "tools/" in f][:3]
        rows.append((e.count / N, dt / N, e.key, " <- ".join(w.strip()[-60:] for w in where)))
for e in p.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::to", "aten::_to_copy"):
        print("  ", e.key, e.count / N, e.input_shapes)
for c, t, k, w in sorted(rows, key=lambda r: -r[0])[:30]:
    print(f"{k[:40]:40s} {c:5.1f}/sequence {t:8.1f} us/sequence  {w}")
