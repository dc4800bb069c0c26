"""BASELINE config 1 ("FFHQ 64x64 NeRF-only output, 1 view, CPU", SURVEY 8d / BASELINE.md section 3): the CPU restatement of the
path (oracle/, test infrastructure) timed on this host with all threads and with one, whole Generator.forward and
renderer-only, D = 8 and D = 2, N = 24, B = 1, test__rendering_time semantics -- and the HIP path on the same workload when
a GPU is present.  Prints one JSON object.

    python tools/cpu_baseline_config1.py [--views 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, weights
from oracle import path as O

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=5)
a = ap.parse_args()
all_threads = torch.get_num_threads()
nerf_cfg = {"N_samples": 24, "perturb": True, "static_viewdirs": False}
res = {"host_threads": all_threads, "cpu_model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
       "rows": []}


def median(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2]


for D in (8, 2):
    cfg = configs.ffhq_G_cfg(64, D)
    torch.manual_seed(0)
    G = pkg.Generator(**cfg)
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0)
    g = torch.Generator().manual_seed(12345)
    zs = [torch.randn(1, 256, generator=g), torch.randn(1, 256, generator=g)]
    cam = O.camera_params(torch.zeros(1, 2), 64, 6, 0.12)
    for nt in (all_threads, 1):
        torch.set_num_threads(nt)
        tf, tr = [], []
        with torch.no_grad():
            for it in range(a.views + 1):
                nb = O.create_noise_bufs(cfg, 64, generator=g)
                u = torch.rand(1, 64, 64, 1, generator=g)
                t0 = time.perf_counter()
                O.generator_forward(sd, cfg, zs, cam[0], cam[1], 64, cam[2], cam[3], nerf_cfg, nb, perturb_u=u)
                t1 = time.perf_counter()
                # renderer only: rays -> samples -> MLP -> compositing
                styles = O.mapping_renderer(sd, cfg, zs[0])
                ro, rd, vd = O.rays_in_world(cam[1], 64, cam[0], False)
                z = O.z_vals(cam[2], cam[3], 1, 64, 64, 24, u)
                pts = O.ray_points(ro, rd, z)
                t2 = time.perf_counter()
                O.renderer_forward(sd, "renderer", pts.reshape(1, 4096, 24, 3), rd.reshape(1, 4096, 3), vd.reshape(1, 4096, 3),
                                   z.reshape(1, 4096, 24), cam[2], cam[3], styles, D)
                t3 = time.perf_counter()
                if it:
                    tf.append(t1 - t0); tr.append(t3 - t2)
        res["rows"].append({"D": D, "threads": nt, "forward_ms": round(median(tf) * 1e3, 1), "renderer_only_ms": round(median(tr) * 1e3, 1),
                            "views_per_s": round(1.0 / median(tf), 3)})
    torch.set_num_threads(all_threads)
    if torch.cuda.is_available():
        Gd = pkg.build_generator(cfg, "cuda", seed=0)
        from cips_3dplusplus_amd.camera import Camera
        e, f, n, fa, _ = Camera.generate_camera_params(64, "cuda", locations=torch.zeros(1, 2, device="cuda"))
        zd = [z_.cuda() for z_ in zs]
        fn = lambda: Gd(zs=zd, cam_poses=e, focals=f, img_size=64, near=n, far=fa, nerf_cfg=nerf_cfg)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300
        res["rows"].append({"D": D, "device": torch.cuda.get_device_name(0), "forward_ms": round(dt * 1e3, 3), "views_per_s": round(1 / dt, 1)})
print(json.dumps(res, indent=1))
