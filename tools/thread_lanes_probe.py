"""Probe: one host thread per lane (the C call releases the GIL) against the one-thread two-lane pipeline."""
import sys, time, threading, json, torch
sys.path.insert(0, ".")
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.camera import Camera
from cips_3dplusplus_amd.pipeline import ViewPipeline, lane_streams
dev = torch.device("cuda:0")
G = pkg.build_generator(configs.ffhq_G_cfg(1024, 2), dev, seed=0)
zs = [torch.randn(1, 256, device=dev), torch.randn(1, 512, device=dev)]
e, f, n, fa, _ = Camera.generate_camera_params(64, dev, locations=torch.tensor([[0.1, -0.05]], device=dev), fov_ang=configs.FFHQ_CAM_CFG["fov_ang"], dist_radius=configs.FFHQ_CAM_CFG["dist_radius"])
nerf_cfg = {"N_samples": 24, "perturb": True, "static_viewdirs": False}
def fwd():
    with torch.no_grad():
        return G(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, truncation=1, nerf_cfg=nerf_cfg)["rgb"]
N = 400
out = {}
for _ in range(5): fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): fwd()
torch.cuda.synchronize(); out["one_stream_ms"] = (time.perf_counter() - t0) / N * 1e3
t0 = time.perf_counter()
for _ in range(N): fwd()
out["one_stream_host_enqueue_ms"] = (time.perf_counter() - t0) / N * 1e3
torch.cuda.synchronize()
for L in (2, 3):
    pipe = ViewPipeline(G, L, device=dev)
    for _ in range(3 * L): pipe.run(fwd, wait_inputs=False)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): pipe.run(fwd, wait_inputs=False)
    pipe.drain(); torch.cuda.synchronize(); out[f"lanes{L}_one_thread_ms"] = (time.perf_counter() - t0) / N * 1e3
    S = lane_streams(dev, L)
    def worker(s, k):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(s):
            for _ in range(k): fwd()
    for rep in range(2):
        th = [threading.Thread(target=worker, args=(s, N // L)) for s in S]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    out[f"lanes{L}_thread_per_lane_ms"] = (t2 - t0) / (N // L * L) * 1e3
    out[f"lanes{L}_thread_per_lane_host_ms"] = (t1 - t0) / (N // L * L) * 1e3
print(json.dumps(out))
