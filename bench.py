"""Benchmark of the generator-forward hot path: rendered views / second.

    python bench.py [--gpus N --steps K --warmup W] [--res 1024 --depth 2 --n-samples 24 --batch 1]

One step = one `Generator.forward` over one batch of synthetic views with inputs resident in HBM
(mapping -> fused NeRF render -> decoder -> rgb in device memory), i.e. the loop body of the
reference's `test__rendering_time` (/root/reference/exp/tests/test_cips3dpp.py:709-748): FFHQ 1024^2
release recipe (N_layers_renderer=2, 1x1 modulated convs), batch 1, 64x64 rays x 24 samples,
perturb=True, fresh decoder noise every call, random-init (synthetic) weights, fp32.
For N > 1 GPUs (launched with torch.distributed.run) every rank renders its own batch per step and the
images are gathered to rank 0 over RCCL inside the timed region (weak scaling).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PUBLISHED_VIEWS_PER_S = 46.93085418313323   # test__rendering_time docstring, unknown CUDA GPU
MFMA_F32_PEAK_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_* dense peak


def nerf_flops_per_point(H, D):
    # first layer + (D-1) hidden + view layer (H+3 inputs) + sigma head + rgb head
    return 2 * 3 * H + (D - 1) * 2 * H * H + 2 * (H + 3) * H + 2 * H * 1 + 2 * H * 3


def cpu_baseline(cfg, nerf_cfg, batch, seconds_budget=25.0):
    """The oracle (CPU port of the reference path) timed on this host on the same workload."""
    from cips_3dplusplus_amd import weights
    from oracle import path as O
    import cips_3dplusplus_amd as pkg
    torch.manual_seed(0)
    G = pkg.Generator(**cfg)
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0)
    del G
    # the per-op torch CPU path stops scaling early: on the GPU box's 64-core / 128-thread host one 1024^2 view takes 4.0 s
    # with 128 threads, 2.5 s with 64, 1.9 s with 32, 1.7 s with 16, 1.9 s with 8 -- time it where it is fastest
    all_threads = torch.get_num_threads()
    cores = min(all_threads, 16)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(12345)
    zs = [torch.randn(batch, 256, generator=g), torch.randn(batch, 256, generator=g)]
    cam = O.camera_params(torch.zeros(batch, 2), 64, 6, 0.12)
    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        for it in range(6):
            nb = O.create_noise_bufs(cfg, 64, generator=g)
            u = torch.rand(batch, 64, 64, 1, generator=g)
            t0 = time.perf_counter()
            O.generator_forward(sd, cfg, zs, cam[0], cam[1], 64, cam[2], cam[3], nerf_cfg, nb, perturb_u=u)
            times.append(time.perf_counter() - t0)
            if it >= 1 and time.perf_counter() - t_all > seconds_budget:
                break
    torch.set_num_threads(all_threads)
    timed = times[1:] if len(times) > 1 else times   # first call is the warm-up
    per_view = sum(timed) / len(timed) / batch
    return {"value": 1.0 / per_view, "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{len(timed)} forward(s) of the same workload after 1 warm-up ({per_view * 1e3:.0f} ms/view; "
                      f"{cores} of {all_threads} host threads: more are slower)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--n-samples", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="A/B: no HIP events around the dominant kernel (roofline = null)")
    ap.add_argument("--deterministic", action="store_true", help="perturb off + fixed noise buffers (demo semantics)")
    ap.add_argument("--decoder-precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = BASELINE config 3 (decoder GEMMs on bf16 MFMA, fp32 accumulate; NeRF stays fp32)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # CIPS3D_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (ranks then share
    # devices round-robin); the driver's runs use the default: nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("CIPS3D_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    if a.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs, hip
    from cips_3dplusplus_amd.camera import Camera
    from cips_3dplusplus_amd.multiview import gather_views_async

    cfg = configs.ffhq_G_cfg(a.res, a.depth)
    nerf_cfg = {"N_samples": a.n_samples, "perturb": not a.deterministic, "static_viewdirs": False}
    G = pkg.build_generator(cfg, dev, seed=0)
    G.set_decoder_precision(a.decoder_precision)
    B = a.batch
    gen = torch.Generator(device=dev).manual_seed(12345 + rank)
    zs = [torch.randn(B, 256, device=dev, generator=gen), torch.randn(B, 256, device=dev, generator=gen)]
    locs = torch.zeros(B, 2, device=dev) if B == 1 else torch.randn(B, 2, device=dev, generator=gen) * torch.tensor([0.3, 0.15], device=dev)
    e, f, n, fa, _ = Camera.generate_camera_params(64, dev, locations=locs, **{k: v for k, v in configs.FFHQ_CAM_CFG.items()
                                                                               if k in ("fov_ang", "dist_radius")})
    noise_bufs = G.create_noise_bufs(64, dev) if a.deterministic else None

    pending = [None]

    def step():
        r = G(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, truncation=1, noise_bufs=noise_bufs,
              nerf_cfg=nerf_cfg)
        rgb = r["rgb"]
        if world > 1:
            # the one exchange step of the path: finished images -> rank 0.  uint8 on the device first (what the
            # demo loop turns every frame into anyway) = 4x fewer bytes over xGMI; asynchronous, so the gather of
            # step i overlaps the rendering of step i+1.
            if pending[0] is not None:
                pending[0].wait(assemble=False)       # the gathered blocks stay on rank 0; no per-step concatenation copy
            pending[0] = gather_views_async(hip.rgb_to_uint8(rgb), B * world)
        return rgb

    def barrier():
        if world > 1:
            if pending[0] is not None:
                pending[0].wait(assemble=False)
                pending[0] = None
            torch.cuda.synchronize()
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    # HIP events around the dominant kernel, recorded inside cips3d_generator_forward on the stream it launches on.  An
    # event record between kernels drains the queue (~6 us each, measured), so every `stride`-th step is instrumented, with
    # the event handles created here, before the timed region.
    stride = 8 if a.steps >= 64 else 1
    hip.KERNEL_EVENTS_STRIDE = stride
    hip.prepare_event_pairs(a.steps // stride + 2)
    barrier()
    if not a.no_kernel_events:
        hip.KERNEL_EVENTS["nerf_render"] = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    events = hip.KERNEL_EVENTS.pop("nerf_render", [])
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        views = a.steps * B * world
        value = views / elapsed
        kern_ms = sum(s.elapsed_time(t) for s, t in events) / max(1, len(events)) if events else float("nan")
        H = cfg["renderer_cfg"]["hidden_dim"]
        flops = B * 64 * 64 * a.n_samples * nerf_flops_per_point(H, a.depth)
        achieved = flops / (kern_ms * 1e-3) / 1e12
        published_cfg = (a.res == 1024 and a.depth == 2 and a.n_samples == 24 and B == 1 and not a.deterministic and
                         a.decoder_precision == "fp32")
        # HBM bytes per launch of the dominant kernel come from a separate rocprofv3 --pmc pass (FETCH_SIZE and
        # WRITE_SIZE cannot share a pass with timing); the committed summary is quoted for the matching config.
        traffic = None
        tp = os.path.join(ROOT, "profiles", "r01_pmc_nerf_traffic.json")
        if a.depth == 2 and a.n_samples == 24 and B == 1 and os.path.exists(tp):
            traffic = json.load(open(tp)).get("traffic_bytes_per_launch")
        line = {
            "metric": "rendered views/sec at FFHQ 1024^2 (generator forward: 64x64-ray NeRF + StyleGAN2 decoder)",
            "value": value, "unit": "views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (value / PUBLISHED_VIEWS_PER_S) if published_cfg else None,
            "dtype": "f32" if a.decoder_precision == "fp32" else "bf16 decoder GEMMs (f32 accumulate), f32 NeRF",
            "data": "synthetic",
            "config": {"workload": f"ffhq_r{a.res}_nerf64x64x{a.n_samples}_D{a.depth}_B{B}_{a.decoder_precision} "
                                   f"(test__rendering_time loop body: perturb={not a.deterministic}, "
                                   f"{'fixed' if a.deterministic else 'fresh'} decoder noise, random-init weights)",
                       "views_per_step_per_gpu": B, "img_size": 64, "n_samples": a.n_samples,
                       "N_layers_renderer": a.depth, "resolution": a.res, "parallelism": f"views x{world}"},
            "roofline": {"kernel": "nerf_render_kernel (FiLM-SIREN point MLP + compositing)", "bound": "mfma",
                         "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                         "avg_launch_ms": kern_ms, "flop_per_launch": flops, "launches_timed": len(events),
                         "timed_every_nth_step": stride},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, {**nerf_cfg, "perturb": True}, B)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
