"""Probe (timing only): how fast do views go when the host is out of the way?  K generators (one forward plan each), each forward
captured into a HIP graph on its own stream; the graphs are replayed round-robin.  Compares with the eager two-lane pipeline."""
import json, sys, time
import torch
sys.path.insert(0, ".")
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
from cips_3dplusplus_amd.camera import Camera
from cips_3dplusplus_amd.pipeline import ViewPipeline

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = 400
cfg = configs.ffhq_G_cfg(1024, 2)
nerf_cfg = {"N_samples": 24, "perturb": True, "static_viewdirs": False}
Gs = [pkg.build_generator(cfg, dev, seed=0) for _ in range(K)]
zs = [torch.randn(1, 256, device=dev), torch.randn(1, 512, device=dev)]
cam = Camera.generate_camera_params(64, dev, locations=torch.tensor([[0.1, -0.05]], device=dev),
                                    **{k: v for k, v in configs.FFHQ_CAM_CFG.items() if k in ("fov_ang", "dist_radius")})
e, f, n, fa, _ = cam


nb = Gs[0].create_noise_bufs(64, dev)
u = torch.rand(1, 64, 64, 1, device=dev)


def fwd(G):   # (fixed jitter / noise buffers: the generator's own draws read the torch generator, which a capture cannot)
    with torch.no_grad():
        return G(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, truncation=1, nerf_cfg=nerf_cfg, noise_bufs=nb,
                 perturb_u=u)["rgb"]


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for G in Gs:
    for _ in range(3):
        fwd(G)
torch.cuda.synchronize()
out = {}
out["eager_1_stream_ms"] = timed(lambda n: [fwd(Gs[0]) for _ in range(n)], steps)
pipe = ViewPipeline(Gs[0], 2, device=dev)
for _ in range(6):
    pipe.run(lambda: fwd(Gs[0]))
pipe.drain()


def eager2(n):
    for _ in range(n):
        pipe.run(lambda: fwd(Gs[0]), wait_inputs=False)
    pipe.drain()
out["eager_2_lanes_ms"] = timed(eager2, steps)

streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
graphs, outs = [], []
for G, s in zip(Gs, streams):
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd(G)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        o = fwd(G)
    graphs.append(g); outs.append(o)
torch.cuda.synchronize()
for k in range(1, K + 1):
    def rep(n, k=k):
        for i in range(n):
            j = i % k
            with torch.cuda.stream(streams[j]):
                graphs[j].replay()
    rep(20); torch.cuda.synchronize()
    out[f"graph_{k}_streams_ms"] = timed(rep, steps)
    # host time alone: enqueue without waiting
    torch.cuda.synchronize()
    t0 = time.perf_counter(); rep(50); t1 = time.perf_counter()
    torch.cuda.synchronize()
    out[f"graph_{k}_streams_host_enqueue_ms"] = (t1 - t0) / 50 * 1e3
print(json.dumps(out))
