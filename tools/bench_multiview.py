"""BASELINE config 4 on one GPU: multiview.sample_multi_view (yaw, N = 128, truncation 0.5, fixed noise) with the sequence's style
tables hoisted (default) against per-frame recomputation, at several frames-per-call, interleaved in ONE process (rule 24).
    python tools/bench_multiview.py [--frames 8] [--rounds 5]"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.multiview import sample_multi_view

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--n-samples", type=int, default=128)
ap.add_argument("--chunks", default="1,2,4,8")
ap.add_argument("--lanes", default="2", help="comma list: sample_multi_view(lanes=...) values to compare (2 = its default)")
ap.add_argument("--only", default="", help="substring filter on the variant names (for rocprofv3 runs of ONE variant)")
a = ap.parse_args()
dev = "cuda"
G = pkg.build_generator(configs.ffhq_G_cfg(1024, 2), dev, seed=0)
g = torch.Generator(device=dev).manual_seed(4)
zs = [torch.randn(1, 256, device=dev, generator=g), torch.randn(1, 256, device=dev, generator=g)]
cam_cfg = {"img_size": 64, "fov_ang": configs.FFHQ_CAM_CFG["fov_ang"], "dist_radius": configs.FFHQ_CAM_CFG["dist_radius"]}
ncfg = {"N_samples": a.n_samples, "perturb": False, "static_viewdirs": False}
nb = G.create_noise_bufs(64, dev)
lanes = [int(x) for x in a.lanes.split(",")]
variants = [(f"chunk{c}_{'hoist' if h else 'per_frame'}{'' if u else '_u8_kernel_off'}{'' if len(lanes) == 1 else f'_lanes{l}'}", c, h, u, l)
            for c in map(int, a.chunks.split(",")) for h, u in ((False, True), (True, True), (True, False)) for l in lanes]
if a.only:
    variants = [v for v in variants if a.only in v[0] and (a.only.endswith("off") or not v[0].endswith("off"))]
run = lambda c, h, u=True, l=2: sample_multi_view(G, cam_cfg, ncfg, zs, view_mode="yaw", N_frames=a.frames, truncation_ratio=0.5,   # noqa: E731
                                             N_samples=a.n_samples, noise_bufs=nb, chunk=c, hoist=h, uint8_in_kernel=u, lanes=l)
for _, c, h, u, l in variants:
    run(c, h, u, l); run(c, h, u, l)
torch.cuda.synchronize()
times = {v[0]: [] for v in variants}
for _ in range(a.rounds):
    for n, c, h, u, l in variants:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run(c, h, u, l)
        torch.cuda.synchronize()
        times[n].append((time.perf_counter() - t0) / 3 / a.frames)
for n, v in times.items():
    print(f"{n:42s} median {statistics.median(v) * 1e6:7.1f} us/frame  min {min(v) * 1e6:7.1f}  -> {1 / statistics.median(v):7.1f} views/s")
