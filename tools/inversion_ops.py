"""Which torch ops launch kernels inside one flip-inversion step (the launches that are not the library's): torch.profiler
over 20 pose-phase steps, ops with device time, per step.   python tools/inversion_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.projector import FlipProjector, surrogate_loss

dev = "cuda"
G = pkg.build_generator(configs.ffhq_G_cfg(256, 6), dev, seed=0)
cam_cfg = {"img_size": 64, "fov_ang": configs.COMPCARS_CAM_CFG["fov_ang"], "dist_radius": configs.COMPCARS_CAM_CFG["dist_radius"]}
ncfg = {"N_samples": 24, "perturb": False, "static_viewdirs": True}
g = torch.Generator(device=dev).manual_seed(1)
t_rgb = torch.randn(2, 3, 256, 256, device=dev, generator=g).clamp(-1, 1)
t_thumb = torch.randn(2, 3, 64, 64, device=dev, generator=g).clamp(-1, 1)
proj = FlipProjector(G, dev)
STEPS, SKIP = 24, 4
state = {}

def on_step(step, loss, azim, elev):
    if step == SKIP - 1:
        torch.cuda.synchronize()
        state["p"] = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
        state["p"].__enter__()

proj.project_wplus(cam_cfg, ncfg, surrogate_loss(t_rgb, t_thumb), N_steps_pose=STEPS, N_steps_app=0, w_avg_samples=500,
                   on_step=on_step, azim_init=(-1.0, 3.0))
torch.cuda.synchronize()
state["p"].__exit__(None, None, None)
n = STEPS - SKIP
rows = []
for e in state["p"].key_averages(group_by_stack_n=12):
    dt = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if dt > 0 and e.key.startswith("aten::"):
        where = [f for f in e.stack if "cips_3dplusplus_amd" in f or "tools/" in f][:2]
        rows.append((e.count / n, dt / n, e.key, " <- ".join(w.strip()[-70:] for w in where)))
for c, t, k, w in sorted(rows, key=lambda r: -r[0]):
    print(f"{k:16s} {c:5.1f}/step {t:7.1f} us/step  {w}")
