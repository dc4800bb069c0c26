"""Host-side cost of one Generator.forward call (enqueue only) against the GPU time of the step.

    python tools/host_time.py [--steps 300]
With the queue kept shallow (a synchronize every `--depth` steps) the enqueue time is what the host needs per call."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.camera import Camera

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
a = ap.parse_args()
dev = "cuda"
G = pkg.build_generator(configs.ffhq_G_cfg(1024, 2), dev, seed=0)
e, f, n, fa, _ = Camera.generate_camera_params(64, dev, locations=torch.zeros(1, 2, device=dev))
zs = [torch.randn(1, 256, device=dev), torch.randn(1, 256, device=dev)]
ncfg = dict(N_samples=24, perturb=True, static_viewdirs=False)
fn = lambda: G(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, nerf_cfg=ncfg)
for _ in range(20):
    fn()
torch.cuda.synchronize()
# (1) GPU-bound loop
t0 = time.perf_counter()
for _ in range(a.steps):
    fn()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"free-running: enqueue loop returned after {t_enq / a.steps * 1e6:.0f} us/step, GPU done after {t_all / a.steps * 1e6:.0f} us/step")
# (2) host cost with an empty queue: one call at a time
ts = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"single call on an idle queue: median {ts[len(ts) // 2] * 1e6:.0f} us host time to enqueue one forward")
