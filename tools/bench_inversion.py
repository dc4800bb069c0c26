"""BASELINE config 5: CompCars 256^2 generator + pose phase of the flip-inversion loop (steps/s).

    python tools/bench_inversion.py [--depth 6] [--steps 200] [--res 256]

One step = forward (batch 2: image + mirrored view) + backward + three Adam steps over {azim, elev}, the NeRF W+ style and
(with lr 0 in this phase, as projector_v10.py:1074-1075 sets it) the decoder W+ / parameters.  Surrogate loss
(SURVEY 8d): MSE(rgb) + 50 MSE(thumb) against fixed random targets.  Random-init weights."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.projector import FlipProjector, surrogate_loss

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=6)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--n-samples", type=int, default=24)
ap.add_argument("--app-steps", type=int, default=0)
a = ap.parse_args()
dev = "cuda"
cfg = configs.ffhq_G_cfg(a.res, a.depth)
G = pkg.build_generator(cfg, dev, seed=0)
cam_cfg = {"img_size": 64, "fov_ang": configs.COMPCARS_CAM_CFG["fov_ang"], "dist_radius": configs.COMPCARS_CAM_CFG["dist_radius"]}
ncfg = {"N_samples": a.n_samples, "perturb": False, "static_viewdirs": True}
g = torch.Generator(device=dev).manual_seed(1)
t_rgb = torch.randn(2, 3, a.res, a.res, device=dev, generator=g).clamp(-1, 1)
t_thumb = torch.randn(2, 3, 64, 64, device=dev, generator=g).clamp(-1, 1)
proj = FlipProjector(G, dev)
marks = {}

def on_step(step, loss, azim, elev):
    if step == 4:                       # first steps: allocator warm-up, plan builds
        torch.cuda.synchronize(); marks["t0"] = time.perf_counter(); marks["s0"] = step
    marks["last"] = float(loss.detach()) if step % 50 == 0 else marks.get("last")

out = proj.project_wplus(cam_cfg, ncfg, surrogate_loss(t_rgb, t_thumb), N_steps_pose=a.steps, N_steps_app=a.app_steps,
                         w_avg_samples=2000, on_step=on_step, azim_init=(-1.0, 3.0))
torch.cuda.synchronize()
dt = time.perf_counter() - marks["t0"]
n = a.steps + a.app_steps - 1 - marks["s0"]
print(json.dumps({"metric": "flip-inversion steps/s (fwd + bwd + Adam, batch 2)", "value": n / dt, "unit": "steps/s",
                  "ms_per_step": dt / n * 1e3, "config": {"workload": f"compcars_r{a.res}_D{a.depth}_N{a.n_samples}_B2_pose_phase",
                  "steps": a.steps, "app_steps": a.app_steps}, "dtype": "f32", "data": "synthetic",
                  "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))
