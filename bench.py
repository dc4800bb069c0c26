"""Benchmark of the generator-forward hot path: rendered views / second.

    python bench.py [--gpus N --steps K --warmup W] [--res 1024 --depth 2 --n-samples 24 --batch 1]

One step = one `Generator.forward` over one batch of synthetic views with inputs resident in HBM
(mapping -> fused NeRF render -> decoder -> rgb in device memory), i.e. the loop body of the
reference's `test__rendering_time` (/root/reference/exp/tests/test_cips3dpp.py:709-748): FFHQ 1024^2
release recipe (N_layers_renderer=2, 1x1 modulated convs), batch 1, 64x64 rays x 24 samples,
perturb=True, fresh decoder noise every call, random-init (synthetic) weights, fp32.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (a fresh
`python -m torch.distributed.run` child, before this process touches the GPU) -- the launch form of the
reference's own multi-process scripts (exp/tests/test_cips3dpp.py:814-820, scripts/gen_images.py:44-84).
Every rank renders its own batch per step and the images are gathered to rank 0 over RCCL inside the
timed region (weak scaling).

The timed region is K steps between (barrier + device synchronise) brackets, max over ranks; it is
repeated `--repeats` times and the MEDIAN repeat is reported (all repeats are listed).  At N = 1 the other
BASELINE.json configurations are timed after the headline and reported under "also" in the same line.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import math
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DTYPE_NAMES = {"fp32": "f32 (NeRF point MLP and decoder GEMMs: fp32-equivalent split-fp16 MFMA products -- three exact fp16 products per "
                       "fp32 product, fp32 accumulate, fp32 storage; operands range-tracked: per-tensor power-of-two scales)",
               "fp32_exact": "f32 (IEEE fp32 products everywhere: the NeRF point MLP AND the decoder GEMMs on the fp32 matrix instruction "
                             "v_mfma_f32_16x16x4_f32, fp32 accumulate, fp32 storage)",
               "bf16": "bf16 decoder GEMM operands (f32 accumulate, f32 storage), f32 NeRF",
               "bf16_storage": "bf16 decoder GEMM operands + bf16 storage of the up-sampling stages' activations "
                               "(f32 accumulate), f32 NeRF"}
PUBLISHED_VIEWS_PER_S = 46.93085418313323   # test__rendering_time docstring, unknown CUDA GPU
MFMA_F32_PEAK_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (the fp32 matrix instruction)
MFMA_F16_PEAK_TFLOPS = 2500.0              # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak
SPLIT_PRODUCTS = 3                         # fp16 products the render kernel issues per fp32 product (w_hi x_hi + w_hi x_lo + w_lo x_hi)
EVENT_STRIDE = 8                           # HIP events around the dominant kernel on every 8th step (a record drains the queue)
# HBM traffic per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass with timing or
# with each other): tools/profile_round.sh writes one summary per workload, stamped with the source hash of the library that ran.
# An entry is replayed only while that hash equals the loaded library's (a kernel edit silently invalidated round 3's file).
TRAFFIC_FILES = {"headline": "pmc_all_kernels.json", "n64": "pmc_n64_traffic.json", "b4_bf16": "pmc_b4_traffic.json"}   # profiles/rNN_<name>, newest round first
HBM_PEAK_TBS = 8.0                         # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured with a float4 copy)
PREROLL_MAX = 10                           # untimed regions until two consecutive ones agree within 1 % (clocks / caches settled)


def dedupe_traffic(rows):
    """A replayed counter belongs to the kernel's dominant shape only (same name, same grid, different K: the 256 -> 512 chain layer
    reads half the planes of the 512 -> 512 ones): of the GEMM rows that share a kernel name only the one with the most launches
    per step keeps `traffic`.  Fused-stage rows are keyed by their width already."""
    for r in rows:
        if r.get("traffic") is None or r.get("kind") == "fused_stage":
            continue
        same = [q for q in rows if q.get("kernel") == r.get("kernel") and q.get("kind") != "fused_stage"]
        if r is not max(same, key=lambda q: q.get("launches_per_step", 0)):
            r["traffic"], r["traffic_source"] = None, "no counter pass keyed on this shape"
    return rows


def nerf_flops_per_point(H, D):
    # first layer + (D-1) hidden + view layer (H+3 inputs) + sigma head + rgb head
    return 2 * 3 * H + (D - 1) * 2 * H * H + 2 * (H + 3) * H + 2 * H * 1 + 2 * H * 3


def make_inputs(rank, B, dev):
    """Per-rank synthetic inputs of a step (device RNG seeded by rank): latent codes and camera angles."""
    gen = torch.Generator(device=dev).manual_seed(12345 + rank)
    zs = [torch.randn(B, 256, device=dev, generator=gen), torch.randn(B, 256, device=dev, generator=gen)]
    if B == 1:
        locs = torch.zeros(B, 2, device=dev)      # test__rendering_time: frontal camera
    else:
        locs = torch.randn(B, 2, device=dev, generator=gen) * torch.tensor([0.3, 0.15], device=dev)
    return zs, locs


def cpu_baseline(cfg, nerf_cfg, batch, seconds_budget=16.0):
    """The oracle (CPU port of the reference path) timed on this host on the same workload."""
    from cips_3dplusplus_amd import weights
    from oracle import path as O
    import cips_3dplusplus_amd as pkg
    G = pkg.Generator(**cfg)
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0)
    del G
    # the per-op torch CPU path stops scaling early: on the GPU box's 64-core / 128-thread host one 1024^2 view takes 4.0 s
    # with 128 threads, 2.5 s with 64, 1.9 s with 32, 1.7 s with 16, 1.9 s with 8 -- time it where it is fastest, and once
    # with a single thread
    all_threads = torch.get_num_threads()
    cores = min(all_threads, 16)
    zs, _, _ = weights.synth_inputs(cfg, batch=batch)
    cam = O.camera_params(torch.zeros(batch, 2), 64, 6, 0.12)
    g = torch.Generator().manual_seed(12345)

    def one_view():
        nb = O.create_noise_bufs(cfg, 64, generator=g)
        u = torch.rand(batch, 64, 64, 1, generator=g)
        t0 = time.perf_counter()
        O.generator_forward(sd, cfg, zs, cam[0], cam[1], 64, cam[2], cam[3], nerf_cfg, nb, perturb_u=u)
        return time.perf_counter() - t0

    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        torch.set_num_threads(cores)
        for it in range(3 + 5):                  # 3 warm-ups, then up to 5 timed views inside the budget
            times.append(one_view())
            if it >= 3 and time.perf_counter() - t_all > seconds_budget:
                break
        warm = min(3, len(times) - 1)
        timed = times[warm:]
        torch.set_num_threads(1)
        t1 = one_view()
        # SURVEY 8d asks for "all physical cores" beside it: one view with every host thread torch was given (slower than `cores`
        # threads on the GPU box's 128-thread host -- which is why `value` is not timed there)
        t_allthr = None
        if all_threads != cores:
            torch.set_num_threads(all_threads)
            one_view()
            t_allthr = one_view()
    torch.set_num_threads(all_threads)
    per_view = statistics.median(timed) / batch
    return {"value": 1.0 / per_view, "unit": "views/s", "cores": cores, "host_threads": all_threads, "kind": "port",
            "value_1_thread": batch / t1, "value_all_threads": (batch / t_allthr) if t_allthr else 1.0 / per_view,
            "sample": f"median of {len(timed)} forward(s) of the same workload after {warm} warm-up(s) "
                      f"({per_view * 1e3:.0f} ms/view with {cores} of {all_threads} host threads: more are slower; "
                      f"{t1 / batch * 1e3:.0f} ms/view with 1 thread, one view)"}


class ForwardWorkload:
    """One BASELINE configuration of the generator forward: builds the generator + inputs, runs steps."""

    def __init__(self, dev, rank, world, res, depth, n_samples, batch, precision, deterministic, lanes=1):
        import cips_3dplusplus_amd as pkg
        from cips_3dplusplus_amd import configs
        from cips_3dplusplus_amd.camera import Camera
        self.dev, self.rank, self.world = dev, rank, world
        self.res, self.depth, self.n_samples, self.B = res, depth, n_samples, batch
        self.precision, self.deterministic = precision, deterministic
        self.cfg = configs.ffhq_G_cfg(res, depth)
        self.nerf_cfg = {"N_samples": n_samples, "perturb": not deterministic, "static_viewdirs": False}
        self.G = pkg.build_generator(self.cfg, dev, seed=0)
        self.G.set_precision(precision)              # "fp32_exact" switches the renderer's point MLP as well as the decoder
        self.zs, locs = make_inputs(rank, batch, dev)
        self.cam = Camera.generate_camera_params(64, dev, locations=locs, **{k: v for k, v in configs.FFHQ_CAM_CFG.items()
                                                                             if k in ("fov_ang", "dist_radius")})
        self.noise_bufs = self.G.create_noise_bufs(64, dev) if deterministic else None
        if deterministic:                          # every rank must use the same buffers
            g = torch.Generator(device=dev).manual_seed(777)
            self.noise_bufs = [torch.randn(b.shape, device=dev, generator=g) for b in self.noise_bufs]
        self.pending = None
        self.last_gathered = None
        # views in flight: consecutive (independent) steps alternate between `lanes` streams (pipeline.ViewPipeline); the timed
        # regions end with every lane drained and the device synchronised.  1 = every step on the one current stream.
        from cips_3dplusplus_amd.pipeline import ViewPipeline
        self.lanes = lanes
        self.pipe = ViewPipeline(self.G, lanes, device=dev) if lanes > 1 else None

    def name(self):
        return f"ffhq_r{self.res}_nerf64x64x{self.n_samples}_D{self.depth}_B{self.B}_{self.precision}"

    def note(self):
        return (f"test__rendering_time loop body: perturb={not self.deterministic}, "
                f"{'fixed' if self.deterministic else 'fresh'} decoder noise, random-init weights")

    def render(self, post=None):
        """One view (batch) on the next lane.  post: applied to the image on the same lane (the uint8 conversion of the gather)."""
        e, f, n, fa, _ = self.cam

        def one():
            with torch.no_grad():                   # `with torch.set_grad_enabled(False)`, test_cips3dpp.py:724
                rgb = self.G(zs=self.zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, truncation=1,
                             noise_bufs=self.noise_bufs, nerf_cfg=self.nerf_cfg)["rgb"]
            return post(rgb) if post is not None else rgb
        if self.pipe is None:
            return one()
        # (the inputs were made and synchronised before the first timed region: no ordering behind the caller's stream, whose
        # per-step gather waits would otherwise chain the views)
        return self.pipe.run(one, wait_inputs=False)

    def fp32_equivalence(self):
        """One view rendered in the default arithmetic (split-fp16 products in the point MLP and the decoder GEMMs) and with
        IEEE-fp32 products everywhere ("fp32_exact": the fp32 matrix instruction in both), same latents / camera / jitter /
        noise: how far apart the two images are.  (Against fp64 both err like plain fp32: tests/test_gpu_split_fp16.py.)"""
        e, f, n, fa, _ = self.cam
        nb = self.G.create_noise_bufs(64, self.dev)
        u = torch.rand(self.B, 64, 64, 1, device=self.dev)
        kw = dict(zs=self.zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, truncation=1, noise_bufs=nb, perturb_u=u,
                  nerf_cfg=self.nerf_cfg)
        with torch.no_grad():
            a = self.G(**kw)["rgb"]
            self.G.set_precision("fp32_exact")
            b = self.G(**kw)["rgb"]
            self.G.set_precision(self.precision)
        return {"max_abs_rgb_difference_split_vs_fp32_mfma": float((a - b).abs().max()),
                "rgb_max_abs": float(b.abs().max()),
                "note": "split-fp16 = three exact fp16 products per fp32 product, fp32 accumulate; against fp64 it errs no more "
                        "than the fp32 instruction (tests/test_gpu_split_fp16.py); parity bar of the path: 1e-3 max-abs"}

    def step(self):
        from cips_3dplusplus_amd import hip
        from cips_3dplusplus_amd.multiview import gather_views_async
        if self.world == 1:
            return self.render()
        # the one exchange step of the path: finished images -> rank 0.  uint8 on the device first (what the
        # demo loop turns every frame into anyway) = 4x fewer bytes over xGMI; asynchronous, so the gather of
        # step i overlaps the rendering of step i+1.
        u8 = self.render(post=hip.rgb_to_uint8)
        if self.pipe is not None:
            self.pipe.wait_lane(self.pipe.last_lane)    # the caller's stream (where the collective is enqueued) behind this view's lane
        if self.pending is not None:
            self.pending.wait(assemble=False)           # the gathered blocks stay on rank 0; no per-step concatenation copy
        self.pending = gather_views_async(u8, self.B * self.world)
        return u8

    def barrier(self):
        if self.pipe is not None:
            self.pipe.drain()
        if self.world > 1:
            if self.pending is not None:
                self.last_gathered = self.pending.wait(assemble=True) if self.keep_last else self.pending.wait(assemble=False)
                self.pending = None
            torch.cuda.synchronize()
            torch.distributed.barrier()
        torch.cuda.synchronize()

    keep_last = False

    def measure(self, steps, warmup, repeats, kernel_events=True):
        """`repeats` timed regions of `steps` steps each (barrier + synchronise on both sides, max over ranks)."""
        from cips_3dplusplus_amd import hip
        for _ in range(warmup):
            self.step()
        # pre-roll: --warmup steps do not reach steady clocks (round 2's driver-form repeats fell monotonically by 8 %): run
        # whole untimed regions until two consecutive ones agree within 1 %, then time
        self.preroll = []
        for _ in range(PREROLL_MAX):
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.barrier()
            dt = time.perf_counter() - t0
            if self.world > 1:
                t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                dt = float(t)
            self.preroll.append(dt)
            if len(self.preroll) >= 2 and abs(self.preroll[-1] - self.preroll[-2]) <= 0.01 * self.preroll[-1]:
                break
        # Short regions (the driver's --steps 20) time ONE launch per region, the one in the middle: every record drains the
        # queue (~6-10 us), and three of them in a 7 ms window are 0.4 % of it; long regions keep one launch in eight.
        stride = EVENT_STRIDE if steps >= 64 else max(1, steps)
        hip.KERNEL_EVENTS_STRIDE = stride
        hip.KERNEL_EVENTS_PHASE = stride // 2
        hip._event_calls.pop("nerf_render", None)           # (the phase counts calls from here)
        hip.prepare_event_pairs(repeats * (steps // stride + 2))
        self.event_stride = stride
        events, elapsed = [], []
        for _ in range(repeats):
            self.barrier()
            if kernel_events:
                hip.KERNEL_EVENTS["nerf_render"] = []
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.barrier()
            dt = time.perf_counter() - t0
            events += hip.KERNEL_EVENTS.pop("nerf_render", [])
            if self.world > 1:
                t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                dt = float(t)
            elapsed.append(dt)
        med = statistics.median(elapsed)
        kern_ms = sum(s.elapsed_time(t) for s, t in events) / len(events) if events else None
        return med, elapsed, kern_ms, len(events)

    def traffic_table(self):
        """{kernel-name prefix: (bytes per launch, source note)} of the committed PMC summary that matches this workload AND the
        loaded library; ({}, reason) when there is none."""
        from cips_3dplusplus_amd import build
        key = None
        if (self.res, self.depth, self.B) == (1024, 2, 1) and self.precision == "fp32":
            key = {24: "headline", 64: "n64"}.get(self.n_samples)
        elif (self.res, self.depth, self.B, self.n_samples) == (1024, 2, 4, 24) and self.precision == "bf16":
            key = "b4_bf16"
        if key is None:
            return {}, "no PMC pass committed for this workload"
        import glob
        try:
            with open(build.STAMP) as fh:
                lib_hash = fh.read().strip()
        except OSError:
            lib_hash = None
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + TRAFFIC_FILES[key])), reverse=True)
        if not cands:
            return {}, f"no profiles/rNN_{TRAFFIC_FILES[key]}"
        d, rel = None, None
        for tp in cands:
            try:
                dd = json.load(open(tp))
            except (OSError, ValueError):
                continue
            if dd.get("lib_srchash") and dd.get("lib_srchash") == lib_hash:
                d, rel = dd, os.path.relpath(tp, ROOT)
                break
        if d is None:
            return {}, (f"{os.path.relpath(cands[0], ROOT)} was collected on another library build than the loaded "
                        f"{str(lib_hash)[:12]}: not replayed")
        out = {}
        # summaries since round 5 carry one row per (kernel, grid); the row with the most launches is the kernel's dominant shape
        for e in sorted(d.get("kernels", []), key=lambda e: e.get("launches", 0)):
            if "hbm_fetch_MB_x2" in e and "hbm_write_MB" in e:
                out[e["kernel"]] = (e["hbm_fetch_MB_x2"] * 1e6 + e["hbm_write_MB"] * 1e6,
                                    f"replayed from {rel} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                    f"this workload on this library build; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)")
        return out, None

    def decoder_kernels(self, n_calls=24):
        """Per-kernel table of the decoder for `roofline.kernels`: launch-to-launch intervals of the one-call forward's own
        timeline marks (cips3d_forward_io.ev_marks) over `n_calls` extra forwards OUTSIDE the timed region (a record between
        dependent launches costs 1-2 us of queue drain, so the intervals are upper bounds of the kernel times; the rocprofv3
        summary under profiles/ is the reference for them), medians per (kind, widths, resolution), priced against the bound
        each kernel has."""
        from cips_3dplusplus_amd import hip
        per = {}
        for _ in range(3):
            self.step()
        for _ in range(n_calls):
            m = hip.DecoderMarks()
            hip.DECODER_MARKS = m
            self.step()
            torch.cuda.synchronize()
            for kind, ci, co, hh, us in m.intervals():
                per.setdefault((kind, ci, co, hh), []).append(us)
        # what a record between two launches costs by itself: the two START marks are recorded back to back
        rec_us = statistics.median(per.pop(("start", 0, 0, 0), [0.0]))
        traffic, why = self.traffic_table()
        B, split = self.B, self.precision in ("fp32",)
        mm_peak = (MFMA_F16_PEAK_TFLOPS / SPLIT_PRODUCTS) if split else (MFMA_F32_PEAK_TFLOPS if self.precision == "fp32_exact" else MFMA_F16_PEAK_TFLOPS)
        rows = []
        for (kind, ci, co, hh), v in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
            raw_us = statistics.median(v)
            # a record between two dependent launches costs less than two records back to back (measured: the corrected
            # interval falls ~1.5 us below rocprofv3's kernel time, the raw one ~3.5 us above): the midpoint is priced, both kept
            lo_us = max(raw_us - rec_us, 0.25 * raw_us)
            us = 0.5 * (raw_us + lo_us)
            per_view = len(v) // n_calls
            row = {"kind": kind, "c_in": ci, "c_out": co, "out_res": hh, "launches_per_step": per_view,
                   "avg_launch_ms": us * 1e-3, "launch_ms_bounds": [lo_us * 1e-3, raw_us * 1e-3], "record_cost_ms": rec_us * 1e-3,
                   "interval": "midpoint of [median mark-to-mark interval minus the cost of two back-to-back records, the raw "
                               "interval]; the rocprofv3 --kernel-trace --stats summary under profiles/ is the reference"}
            hw = hh * hh
            if kind in ("planes_gemm", "gemm", "lowres_gemm"):
                flops = 2.0 * B * ci * co * hw
                # (the low-resolution exit of a planes run is the chain kernel on half a grid + the riding ToRGB fold: no counter pass is
                # keyed on that shape, and modconv1x1_kernel's bytes -- the fp32_exact check's launches -- are not its)
                kname = "chain_gemm_kernel" if kind == "planes_gemm" else (
                    "chain_gemm_kernel (low-res exit)" if (kind == "lowres_gemm" and split) else "modconv1x1_kernel")
                row.update(kernel=kname, bound="mfma",
                           flop_per_launch=flops, achieved=flops / (us * 1e-6) / 1e12, peak=mm_peak, unit="TFLOP/s")
            elif kind == "fused_stage":
                C = ci
                # bytes: low-res input + noise maps (two) + next stage's low-res output (or nothing) + rgb out + skip in; fp32
                act_b = 2 if self.precision == "bf16_storage" else 4
                byts = B * (C * (hw // 4) * act_b + co * hw * act_b + 3 * hw * 4 + 3 * (hw // 4) * 4) + 2 * hw * 4
                flops = 2.0 * B * C * C * hw + 2.0 * B * C * co * hw
                row.update(kernel=f"fused_up_conv_kernel<{C}, ...>", bound="hbm", algorithmic_bytes=byts, flop_per_launch=flops,
                           achieved=byts / (us * 1e-6) / 1e12, peak=HBM_PEAK_TBS, unit="TB/s")
            else:
                row.update(kernel="torgb_reduce_kernel / torgb" if kind == "torgb" else kind, bound="latency")
            if "achieved" in row:
                row["frac"] = row["achieved"] / row["peak"]
            for name, (tb, note) in traffic.items():
                if row.get("kernel", "").split("<")[0] and name.startswith(row["kernel"].split("<")[0]) and \
                        (kind != "fused_stage" or name.startswith(f"fused_up_conv_kernel<{ci},")):
                    row["traffic"], row["traffic_source"] = tb, note
                    break
            else:
                row["traffic"] = None
            rows.append(row)
        dedupe_traffic(rows)
        return rows, why

    def roofline(self, kern_ms, n_events, kernels=False):
        if not kern_ms:                              # no kernel events were taken (--no-kernel-events): nothing to price
            return None
        H = self.cfg["renderer_cfg"]["hidden_dim"]
        flops = self.B * 64 * 64 * self.n_samples * nerf_flops_per_point(H, self.depth)
        achieved = flops / (kern_ms * 1e-3) / 1e12
        table, why = self.traffic_table()
        traffic, src = None, why
        for name, (tb, note) in table.items():
            if name.startswith("nerf_render"):
                traffic, src = tb, note
        # The point-MLP GEMMs run as fp32-equivalent SPLIT-fp16 products on v_mfma_f32_16x16x32_f16 (csrc/nerf.hip): every
        # algorithmic fp32 multiply-add costs three fp16 ones, so the matrix-core ceiling for ALGORITHMIC flops is the fp16
        # dense peak / 3.  `achieved` counts algorithmic flops, as before; the fp32 matrix instruction's own peak
        # (what the round-1 kernel was bounded by) is kept beside it.
        exact = self.precision == "fp32_exact"
        peak = MFMA_F32_PEAK_TFLOPS if exact else MFMA_F16_PEAK_TFLOPS / SPLIT_PRODUCTS
        head = {"kernel": ("nerf_render_kernel<16,4,F32> (v_mfma_f32_16x16x4_f32)" if exact else
                           "nerf_render_kernel<16,4>") + " (FiLM-SIREN point MLP + compositing)", "bound": "mfma",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_definition": (f"fp32 matrix instruction peak {MFMA_F32_PEAK_TFLOPS:.1f} TFLOP/s" if exact else
                                    f"fp16 dense MFMA peak {MFMA_F16_PEAK_TFLOPS:.0f} TFLOP/s / {SPLIT_PRODUCTS} fp16 products per "
                                    f"fp32 product (split-fp16 arithmetic, fp32 accumulate)"),
                "executed_f16_mfma_tflops": None if exact else achieved * SPLIT_PRODUCTS,
                "vs_fp32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS, "fp32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": src,
                "avg_launch_ms": kern_ms, "flop_per_launch": flops, "launches_timed": n_events,
                "timed_every_nth_step": getattr(self, "event_stride", EVENT_STRIDE)}
        if kernels:       # every big kernel of the step, the dominant one first (its row repeats the fields above)
            try:
                head["kernels"] = [{"kind": "render", "launches_per_step": 1, **{k: head[k] for k in (
                    "kernel", "avg_launch_ms", "bound", "flop_per_launch", "achieved", "peak", "unit", "frac", "traffic", "traffic_source")}}] \
                    + self.decoder_kernels()[0]
            except Exception as exc:           # noqa: BLE001 (the per-kernel table is secondary to the line)
                head["kernels_error"] = f"{type(exc).__name__}: {exc}"[:400]
        return head


def inversion_workload(dev, steps, warmup, repeats, depth=6, res=256, n_samples=24, phase="pose"):
    """BASELINE config 5: CompCars 256^2, D = 6, batch 2 (image + mirrored view): steps/s of the pose phase of the
    flip-inversion loop (forward + backward + Adam; /root/reference/exp/cips3d/models/projector_v10.py:915-1216)."""
    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs, hip
    from cips_3dplusplus_amd.projector import FlipProjector, surrogate_loss
    G = pkg.build_generator(configs.ffhq_G_cfg(res, depth), dev, seed=0)
    cam_cfg = {"img_size": 64, "fov_ang": configs.COMPCARS_CAM_CFG["fov_ang"],
               "dist_radius": configs.COMPCARS_CAM_CFG["dist_radius"]}
    ncfg = {"N_samples": n_samples, "perturb": False, "static_viewdirs": True}
    g = torch.Generator(device=dev).manual_seed(1)
    t_rgb = torch.randn(2, 3, res, res, device=dev, generator=g).clamp(-1, 1)
    t_thumb = torch.randn(2, 3, 64, 64, device=dev, generator=g).clamp(-1, 1)
    proj = FlipProjector(G, dev)
    warmup = max(warmup, 4)                         # first steps: allocator warm-up, plan builds
    elapsed = []
    for _ in range(repeats):
        marks = {}

        def on_step(step, loss, azim, elev):
            if step == warmup - 1:
                torch.cuda.synchronize()
                marks["t0"] = time.perf_counter()

        if phase == "pose":
            proj.project_wplus(cam_cfg, ncfg, surrogate_loss(t_rgb, t_thumb), N_steps_pose=warmup + steps, N_steps_app=0,
                               w_avg_samples=2000, on_step=on_step, azim_init=(-1.0, 3.0))
        else:
            # the appearance phase of the released recipe (train_cips3d_compcars_v10.yaml:585-596: decoder W+ and decoder parameters
            # trainable at their learning rates, truncated NeRF style, decoder-style flips every 10th step, zero noise buffers that
            # are not optimised): every step of the call is an appearance step
            proj.project_wplus(cam_cfg, ncfg, surrogate_loss(t_rgb, t_thumb), N_steps_pose=0, N_steps_app=warmup + steps,
                               w_avg_samples=2000, on_step=on_step, azim_init=(-1.0, 3.0), truncation_psi=0.7)
        torch.cuda.synchronize()
        elapsed.append(time.perf_counter() - marks["t0"])
    med = statistics.median(elapsed)
    H = 256
    line = {"metric": "flip-inversion steps/s (forward + backward + Adam, batch 2 = image + mirrored view)",
            "value": steps / med, "unit": "steps/s", "ms_per_step": med / steps * 1e3, "steps": steps, "repeats": repeats,
            "ms_per_step_repeats": [e / steps * 1e3 for e in elapsed], "dtype": "f32",
            "config": {"workload": f"BASELINE config 5: compcars_r{res}_nerf64x64x{n_samples}_D{depth}_B2 {phase} phase "
                                   f"(surrogate loss, random-init weights)"},
            "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
    if phase != "pose":
        return line
    from cips_3dplusplus_amd import autograd as AG
    line["route"] = ("decoder = one autograd node (cips3d_decoder_grad_forward / _backward), fused NeRF backward, HIP Adam (optim.HipAdam)"
                     if AG.ONE_CALL_DECODER else "decoder = one autograd node per op (CIPS3D_ONE_CALL_DECODER=0)")
    line["roofline"] = hip.inversion_roofline(G.renderer, B=2, n_samples=n_samples)
    # the decoder node's two C calls (forward with kept activations; the whole backward), between HIP events on a few extra steps
    # outside the timed regions: algorithmic flop of the node's GEMMs against the split-fp16 ceiling
    if AG.ONE_CALL_DECODER:
        from cips_3dplusplus_amd import decoder_grad as DG
        hip.KERNEL_EVENTS["decoder_grad_forward"], hip.KERNEL_EVENTS["decoder_grad_backward"] = [], []
        stride, hip.KERNEL_EVENTS_STRIDE = hip.KERNEL_EVENTS_STRIDE, 1
        kept = proj.project_wplus(cam_cfg, ncfg, surrogate_loss(t_rgb, t_thumb), N_steps_pose=14, N_steps_app=0, w_avg_samples=2000,
                                 azim_init=(-1.0, 3.0))          # (kept: the node's plans are keyed weakly by the loop's copy of the decoder)
        torch.cuda.synchronize()
        hip.KERNEL_EVENTS_STRIDE = stride
        evf, evb = hip.KERNEL_EVENTS.pop("decoder_grad_forward"), hip.KERNEL_EVENTS.pop("decoder_grad_backward")
        # (the loop works on a deep copy of G: take the layer list from whichever decoder the node planned for)
        infos = [pl.info for plans in DG._PLANS.values() for pl in plans.values()]
        if evf and evb and infos:
            info = infos[-1]
            gemm = sum(2.0 * 2 * i["Cin"] * i["Cout"] * i["H"] * i["W"] for i in info if i["kind"] < 2)
            rgbf = sum(2.0 * 2 * i["Cin"] * 3 * i["Ho"] * i["Wo"] for i in info if i["kind"] >= 2)
            med = lambda ev: statistics.median([a.elapsed_time(b) for a, b in ev][2:])      # noqa: E731 (ms; the first two steps build plans)
            peak = MFMA_F16_PEAK_TFLOPS / SPLIT_PRODUCTS
            rows = []
            for name, ev, mult, what in (("decoder_node_forward", evf, 1.0, "cips3d_decoder_grad_forward: modulation heads + table, 26 GEMMs, FIR / activation / ToRGB launches, outputs kept"),
                                         ("decoder_node_backward", evb, 2.0, "cips3d_decoder_grad_backward: per layer the weight-gradient and the data-gradient GEMM (+ fused activation / ToRGB backward), modulation and style backward")):
                ms = med(ev)
                fl = mult * (gemm + rgbf)
                rows.append({"kind": name, "what": what, "launch_group_ms": ms, "flop": fl, "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12,
                             "peak": peak, "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / peak,
                             "share_of_step": ms / line["ms_per_step"]})
            line["roofline"]["kernels"] = rows
        del kept
    return line


def multiview_workload(dev, repeats, n_frames=8, n_samples=128, res=1024):
    """BASELINE config 4 with the demo's own semantics on ONE GPU: `multiview.sample_multi_view(view_mode="yaw", N_frames=8,
    N_samples=128, truncation_ratio=0.5)` -- one z pair, ONE set of noise buffers, perturb=False, truncated styles with cached
    means, return_xyz=True, uint8 frames (/root/reference/exp/cips3d/models/render_video_web_v10.py:1806-1826).  The multi-GPU
    leg shards these frames over ranks (`--gpus N`); this entry is the per-GPU work of that loop."""
    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs
    from cips_3dplusplus_amd.multiview import sample_multi_view
    G = pkg.build_generator(configs.ffhq_G_cfg(res, 2), dev, seed=0)
    g = torch.Generator(device=dev).manual_seed(4)
    zs = [torch.randn(1, 256, device=dev, generator=g), torch.randn(1, 256, device=dev, generator=g)]
    cam_cfg = {"img_size": 64, "fov_ang": configs.FFHQ_CAM_CFG["fov_ang"], "dist_radius": configs.FFHQ_CAM_CFG["dist_radius"]}
    ncfg = {"N_samples": n_samples, "perturb": False, "static_viewdirs": False}
    nb = G.create_noise_bufs(64, dev)
    run = lambda chunk=1, hoist=True, lanes=2: sample_multi_view(G, cam_cfg, ncfg, zs, view_mode="yaw", N_frames=n_frames,   # noqa: E731
                                                                 truncation_ratio=0.5, N_samples=n_samples, noise_bufs=nb, chunk=chunk,
                                                                 hoist=hoist, lanes=lanes)
    variants = {"chunk1_per_frame_tables": (1, False, 2), "chunk1_per_frame_tables_one_stream": (1, False, 1),
                "chunk1_hoisted": (1, True, 2), "chunk1_hoisted_one_stream": (1, True, 1), "chunk8_hoisted": (n_frames, True, 1)}
    for c, h, l in variants.values():                # mean latents (10 000 samples, cached on G), plans, allocator
        run(c, h, l)
        run(c, h, l)
    torch.cuda.synchronize()
    elapsed = {k: [] for k in variants}
    for _ in range(repeats):                         # interleaved in one process: same-box, same-clock comparison
        for k, (c, h, l) in variants.items():
            t0 = time.perf_counter()
            for _ in range(3):
                out = run(c, h, l)
            torch.cuda.synchronize()
            elapsed[k].append((time.perf_counter() - t0) / 3)
    # `value` is the quantity every round has reported: each frame recomputes its style tables, as the reference's frame loop does
    # (render_video_web_v10.py:1806-1826).  The sequence plan (tables resident after the first frame) is a variant beside it.
    med = statistics.median(elapsed["chunk1_per_frame_tables"])
    assert out["rgb"].dtype == torch.uint8 and out["rgb"].shape[0] == n_frames
    return {"tag": "config4_multiview_8f_n128", "what": "BASELINE config 4, the demo loop's semantics on one GPU: sample_multi_view(yaw, 8 frames, N = 128, truncation 0.5, "
                    "fixed noise buffers, perturb off, xyz returned, uint8 frames), one frame per call, every frame recomputing its style "
                    "tables, frames alternating between two streams (value: sample_multi_view's lanes = 2; *_one_stream: lanes = 1); variants: the sequence's tables (mapping networks, FiLM table, 26 modulated decoder matrices) computed "
                    "by the first frame and resident for the other seven (chunk1_hoisted: sample_multi_view's default, bit-identical), "
                    "and all eight frames in one batch (chunk8_hoisted)",
            "metric": "rendered views/s", "value": n_frames / med, "unit": "views/s", "ms_per_step": med / n_frames * 1e3,
            "steps": 3 * n_frames, "repeats": repeats, "ms_per_step_repeats": [e / n_frames * 1e3 for e in elapsed["chunk1_per_frame_tables"]], "dtype": DTYPE_NAMES["fp32"],
            "variants_views_per_s": {k: n_frames / statistics.median(v) for k, v in elapsed.items()},
            "config": {"workload": f"ffhq_r{res}_nerf64x64x{n_samples}_D2_B1 x {n_frames} frames (multiview.sample_multi_view)"}}


def measure_with_single_stream(wl, steps, warmup, repeats, batch, kernel_events=True, single_too=True):
    """The workload's timed regions (views in flight as configured), then -- when it runs more than one lane -- a few regions
    of the same steps on ONE stream: the quantity every earlier round reported, and where the dominant kernel's duration is taken
    (alone on the chip).  -> (median region s, region times, kernel ms for the roofline, events, kernel ms in flight | None,
    single-stream dict | None)."""
    med, elapsed, kern_ms, n_ev = wl.measure(steps, warmup, repeats, kernel_events=kernel_events)
    if wl.pipe is None or not single_too:
        return med, elapsed, kern_ms, n_ev, None, None
    pipe, wl.pipe = wl.pipe, None
    try:
        m1, e1, k1, n1 = wl.measure(steps, max(3, warmup // 4), min(3, repeats), kernel_events=kernel_events)
    finally:
        wl.pipe = pipe
    single = {"value": steps * batch / m1, "unit": "views/s", "ms_per_step": m1 / steps * 1e3,
              "ms_per_step_repeats": [e / steps * 1e3 for e in e1],
              "what": "the same steps issued on one stream (views strictly one after the other): the `value` of rounds 1-5"}
    return med, elapsed, (k1 if k1 else kern_ms), (n1 if k1 else n_ev), kern_ms, single


LINE_LIMIT = 6144                          # the driver's record keeps a bounded tail of stdout: round 4's 36 KB line was not parsed
DTYPE_SHORT = {"fp32": "f32 (split-fp16 MFMA products x3, f32 accumulate and storage)",
               "fp32_exact": "f32 (IEEE fp32 MFMA products everywhere)",
               "bf16": "bf16 decoder GEMM operands (f32 accumulate/storage), f32 NeRF",
               "bf16_storage": "bf16 decoder GEMM operands + bf16 stage activations, f32 NeRF"}


def _r(x, sig=4):
    """Numbers of the compact line carry `sig` significant digits (the detail file keeps them whole)."""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if math.isfinite(x) else None
    return x


def _roof_head(rf):
    if not rf:
        return None
    keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "in_flight_avg_launch_ms")
    out = {k: _r(rf.get(k)) for k in keys if k in rf}
    out["kernel"] = str(out["kernel"] or "")[:60]
    if "measured_in" in rf:
        out["measured_in"] = "single-stream regions of this run"
    return out


def compact(line, detail_name):
    """The driver's line: the contract's fields, the roofline head with one short row per big kernel, cpu_baseline, one short
    row per `also` entry.  Everything else (repeats, pre-roll, marks, bounds, provenance prose) stays in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "data", "repeats", "rccl_ranks", "ranks", "dist_backend", "physical_gpus")
    out = {k: _r(line[k], 6) for k in keep if k in line}
    out["dtype"] = line.get("dtype_short", line.get("dtype", ""))[:80]
    out["ms_per_step_repeats"] = [_r(v) for v in line.get("ms_per_step_repeats", [])][:8]
    out["config"] = {k: (v[:100] if isinstance(v, str) else v) for k, v in line.get("config", {}).items()}
    rf = line.get("roofline")
    out["roofline"] = _roof_head(rf)
    if rf and rf.get("kernels"):
        rows = []
        for k in rf["kernels"][:10]:
            rows.append({"kind": k.get("kind"), "c_in": k.get("c_in"), "res": k.get("out_res"), "n": k.get("launches_per_step"),
                         "us": _r(k["avg_launch_ms"] * 1e3) if k.get("avg_launch_ms") else None, "bound": k.get("bound"),
                         "frac": _r(k.get("frac")), "traffic": _r(k.get("traffic"))})
        out["roofline"]["kernels"] = rows
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: (_r(v) if not isinstance(v, str) else v[:120]) for k, v in cb.items()
                               if k in ("value", "unit", "cores", "kind", "sample", "error", "value_1_thread", "value_all_threads", "host_threads")}
    if "also" in line:
        rows = []
        for e in line["also"][:14]:
            row = {"workload": str(e.get("tag") or e.get("config", {}).get("workload") or e.get("what", ""))[:48]}
            if "error" in e:
                row["error"] = e["error"][:80]
            else:
                row.update(value=_r(e.get("value")), unit=e.get("unit"), ms_per_step=_r(e.get("ms_per_step")),
                           frac=_r((e.get("roofline") or {}).get("frac")))
                if e.get("single_stream_views_per_s"):
                    row["one_stream"] = _r(e["single_stream_views_per_s"])
            rows.append(row)
        out["also"] = rows
    if "single_stream" in line:
        out["single_stream"] = {k: _r(v, 6) for k, v in line["single_stream"].items() if k in ("value", "ms_per_step")}
    if "fp32_equivalence" in line and "max_abs_rgb_difference_split_vs_fp32_mfma" in line["fp32_equivalence"]:
        out["split_vs_fp32_exact_max_abs"] = _r(line["fp32_equivalence"]["max_abs_rgb_difference_split_vs_fp32_mfma"])
    out["detail"] = detail_name
    text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    # never exceed the limit: drop the secondary tables from the end, one at a time (the detail file has them)
    for victim in ("also", "ms_per_step_repeats", "cpu_baseline.sample", "roofline.kernels"):
        if len(text) <= LINE_LIMIT:
            break
        a, _, b = victim.partition(".")
        if b and isinstance(out.get(a), dict):
            out[a].pop(b, None)
        else:
            out.pop(a, None)
        out["truncated"] = out.get("truncated", []) + [victim]
        text = json.dumps(out, allow_nan=False, separators=(",", ":"))
    return text


def finite(o):
    if isinstance(o, float) and not math.isfinite(o):
        return None
    if isinstance(o, dict):
        return {k: finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [finite(v) for v in o]
    return o


def emit(line, detail_path):
    """Full record -> `detail_path` (JSON, one object); compact line -> the LAST line of stdout."""
    line = finite(line)
    name = os.path.relpath(detail_path, ROOT) if detail_path else None
    if detail_path:
        try:
            with open(detail_path, "w") as fh:
                json.dump(line, fh, allow_nan=False, indent=1)
        except OSError as exc:                 # a read-only tree must not cost the line
            name = f"not written: {exc}"[:80]
    text = compact(line, name)
    print(text, flush=True)
    return text


def spawn_ranks(a):
    """`--gpus N` without a launcher: start N fresh ranks (this process has not touched the GPU and never will)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n_dev = torch.cuda.device_count()               # counting devices does not initialise HIP on this image
    if n_dev < a.gpus and "CIPS3D_DIST_BACKEND" not in env:
        # fewer devices than ranks (a 1-GPU box): the ranks share devices round-robin and the exchange runs over gloo, so
        # that the N > 1 leg is still exercised end to end; RCCL needs one device per rank.  The line says so.
        env["CIPS3D_DIST_BACKEND"] = "gloo"
        print(f"bench.py: {n_dev} device(s) for {a.gpus} ranks -> ranks share devices, gather over gloo", file=sys.stderr)
    port = env.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--n-samples", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--lanes", type=int, default=2,
                    help="views in flight: consecutive steps alternate between this many streams (pipeline.ViewPipeline); 1 = one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the other BASELINE configurations (N=64, config 3, config 5)")
    ap.add_argument("--no-kernel-events", action="store_true", help="A/B: no HIP events around the dominant kernel (roofline = null)")
    ap.add_argument("--deterministic", action="store_true", help="perturb off + fixed noise buffers (demo semantics)")
    ap.add_argument("--decoder-precision", default="fp32", choices=["fp32", "fp32_exact", "bf16", "bf16_storage"],
                    help="bf16 = BASELINE config 3 (decoder GEMMs on bf16 MFMA, fp32 accumulate; NeRF stays fp32); "
                         "bf16_storage = additionally the up-sampling stages' pre-FIR activations live in HBM as bf16")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where rank 0 writes the full record (repeats, pre-roll, per-kernel bounds, provenance); '' = nowhere")
    ap.add_argument("--dump-gathered", default=None, help="rank 0 saves the last step's gathered uint8 frames (torch.save)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
                         f"(or without a launcher: bench.py starts its own ranks)")
    # Nothing below touches a GPU before the rank has bound ITS device and joined the process group: counting devices does
    # not initialise HIP on this image, and a rank that initialised device 0 first would hold a context there for nothing.
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # CIPS3D_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (ranks then share
    # devices round-robin); the driver's runs use the default: nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("CIPS3D_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, n_dev)
    if world > 1 and torch.cuda.is_initialized():
        raise SystemExit("bench.py: the GPU was initialised before the rank bound its device (a bug in this file)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    wl = ForwardWorkload(dev, rank, world, a.res, a.depth, a.n_samples, a.batch, a.decoder_precision, a.deterministic, lanes=a.lanes)
    wl.keep_last = a.dump_gathered is not None
    med, elapsed, kern_ms, n_ev, kern_in_flight, single = measure_with_single_stream(
        wl, a.steps, a.warmup, a.repeats, a.batch, kernel_events=not a.no_kernel_events, single_too=(world == 1))
    if a.dump_gathered and rank == 0:
        frames = wl.last_gathered if world > 1 else None
        if frames is None:
            from cips_3dplusplus_amd import hip
            frames = hip.rgb_to_uint8(wl.render())
        torch.save(frames.cpu(), a.dump_gathered)

    if rank == 0:
        B = a.batch
        value = a.steps * B * world / med
        published_cfg = (a.res == 1024 and a.depth == 2 and a.n_samples == 24 and B == 1 and not a.deterministic and
                         a.decoder_precision == "fp32")
        line = {
            "metric": "rendered views/sec at FFHQ 1024^2 (generator forward: 64x64-ray NeRF + StyleGAN2 decoder)",
            "value": value, "unit": "views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": med / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (value / PUBLISHED_VIEWS_PER_S) if published_cfg else None,
            "dtype": DTYPE_NAMES[a.decoder_precision], "dtype_short": DTYPE_SHORT[a.decoder_precision],
            "data": "synthetic",
            "repeats": a.repeats, "ms_per_step_repeats": [e / a.steps * 1e3 for e in elapsed],
            "preroll_ms_per_step": [e / a.steps * 1e3 for e in wl.preroll],
            "config": {"workload": wl.name(), "loop": wl.note(), "views_per_step_per_gpu": B, "img_size": 64, "n_samples": a.n_samples,
                       "N_layers_renderer": a.depth, "resolution": a.res, "parallelism": f"views x{world}",
                       "views_in_flight": wl.lanes,
                       "streams": (f"consecutive steps alternate between {wl.lanes} HIP streams (independent views overlap: one view's launch-bound "
                                   "style phase runs under another's large kernels); regions end drained + device-synchronised") if wl.lanes > 1
                                  else "one stream"},
            "roofline": wl.roofline(kern_ms, n_ev, kernels=(world == 1)),
        }
        if line["roofline"] and kern_in_flight is not None:
            # The kernel's duration is taken where it has the chip to itself: the single-stream regions of this run.  In the
            # headline's regions two views are in flight and the launch shares the chip with the other view's small kernels
            # (its wall time there: in_flight_avg_launch_ms) -- that is time of the step, not of this kernel's work.
            line["roofline"]["in_flight_avg_launch_ms"] = kern_in_flight
            line["roofline"]["measured_in"] = "the single-stream timed regions of this run (single_stream); in_flight_avg_launch_ms: the headline's regions"
        if single is not None:
            line["single_stream"] = single
        if world == 1 and a.decoder_precision == "fp32":
            # the default arithmetic (split-fp16 products) against the fp32 matrix instruction on this run's own inputs
            try:
                line["fp32_equivalence"] = wl.fp32_equivalence()
            except Exception as exc:           # noqa: BLE001
                line["fp32_equivalence"] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        if world > 1:
            # ranks that actually exchanged over RCCL: 0 when the gather ran over gloo (ranks sharing a device on a small box)
            line["rccl_ranks"] = torch.distributed.get_world_size() if backend == "nccl" else 0
            line["ranks"] = torch.distributed.get_world_size()
            line["dist_backend"] = backend + (" (RCCL)" if backend == "nccl" else " (ranks share devices: not an RCCL measurement)")
            line["physical_gpus"] = n_dev
        if world == 1 and not a.no_also and published_cfg:
            # the other BASELINE.json configurations, same command, same box (each its own timed regions)
            also = []
            del wl
            torch.cuda.empty_cache()
            for short, tag, kw in (
                            ("n64", "metric's '64^3' reading: 64x64 rays x 64 samples", dict(n_samples=64, batch=1, precision="fp32")),
                            ("fp32_exact", "headline workload with IEEE-fp32 products everywhere (fp32_exact: the point MLP and the decoder GEMMs on the "
                             "fp32 matrix instruction) instead of the default split-fp16 products", dict(n_samples=24, batch=1, precision="fp32_exact")),
                            ("config2_r256_D2", "BASELINE config 2: FFHQ 256^2, D = 2, single view", dict(n_samples=24, batch=1, precision="fp32", res=256)),
                            ("config2_r256_D8", "BASELINE config 2 with the deep renderer: FFHQ 256^2, D = 8", dict(n_samples=24, batch=1, precision="fp32",
                                                                                               res=256, depth=8)),
                            ("config1_r64_D8_hip", "BASELINE config 1 on the HIP path: FFHQ 64x64 output (no up-sampling), D = 8, N = 24, single view (its CPU "
                             "figure: profiles/r01_config1_cpu_vs_gpu.json)", dict(n_samples=24, batch=1, precision="fp32", res=64, depth=8)),
                            ("config4_shape_n128", "BASELINE config 4's per-view shape on one GPU: 1024^2 with N = 128 samples per ray (the reference demo's "
                             "value; the 8-GPU leg of the metric shards whole views, `--gpus N`)", dict(n_samples=128, batch=1, precision="fp32")),
                            ("config3_B4_bf16", "BASELINE config 3: 1024^2, batch 4, bf16 decoder GEMM operands (fp32 storage: the faster of the two bf16 "
                             "modes on this build)", dict(n_samples=24, batch=4, precision="bf16")),
                            ("config3_B4_bf16_storage", "BASELINE config 3, storage mode: bf16 operands + bf16 storage of the up-sampling stages' activations "
                             "(HBM bytes of those stages halved; slower: the stages are VALU-bound)",
                             dict(n_samples=24, batch=4, precision="bf16_storage"))):
                # (a secondary entry must never cost the headline line: a failure is reported in its place)
                try:
                    w2 = ForwardWorkload(dev, 0, 1, kw.get("res", 1024), kw.get("depth", 2), kw["n_samples"], kw["batch"], kw["precision"], False,
                                         lanes=a.lanes)
                    steps2 = max(10, a.steps // 2)
                    m2, e2, k2, n2, k2f, s2 = measure_with_single_stream(w2, steps2, max(3, a.warmup // 2), a.repeats, kw["batch"])
                    also.append({"tag": short, "what": tag, "metric": "rendered views/s", "value": steps2 * kw["batch"] / m2, "unit": "views/s",
                                 "ms_per_step": m2 / steps2 * 1e3, "steps": steps2, "repeats": a.repeats,
                                 "single_stream_views_per_s": s2["value"] if s2 else None,
                                 "ms_per_step_repeats": [e / steps2 * 1e3 for e in e2],
                                 "preroll_ms_per_step": [e / steps2 * 1e3 for e in w2.preroll],
                                 "dtype": DTYPE_NAMES[kw["precision"]],
                                 "config": {"workload": w2.name(), "views_in_flight": w2.lanes}, "roofline": w2.roofline(k2, n2, kernels=kw["batch"] == 4)})
                    del w2
                except Exception as exc:       # noqa: BLE001
                    also.append({"tag": short, "what": tag, "error": f"{type(exc).__name__}: {exc}"[:400]})
                torch.cuda.empty_cache()
            try:
                also.append(multiview_workload(dev, min(a.repeats, 3)))
            except Exception as exc:           # noqa: BLE001
                also.append({"tag": "config4_multiview_8f_n128", "what": "BASELINE config 4 (sample_multi_view)", "error": f"{type(exc).__name__}: {exc}"[:400]})
            try:
                inv = inversion_workload(dev, max(10, min(a.steps, 60)), max(4, a.warmup // 2), min(a.repeats, 3))
                inv["what"], inv["tag"] = "BASELINE config 5: one flip-inversion step", "config5_inversion_pose"
                also.append(inv)
            except Exception as exc:           # noqa: BLE001
                also.append({"tag": "config5_inversion_pose", "what": "BASELINE config 5: one flip-inversion step", "error": f"{type(exc).__name__}: {exc}"[:400]})
            try:
                app = inversion_workload(dev, max(10, min(a.steps, 60)), max(4, a.warmup // 2), min(a.repeats, 3), phase="appearance")
                app["what"], app["tag"] = "BASELINE config 5: one step of the appearance phase (decoder trainable)", "config5_inversion_appearance"
                also.append(app)
            except Exception as exc:           # noqa: BLE001
                also.append({"tag": "config5_inversion_appearance", "what": "BASELINE config 5, appearance phase", "error": f"{type(exc).__name__}: {exc}"[:400]})
            line["also"] = also
        if world == 1 and not a.no_cpu_baseline:
            from cips_3dplusplus_amd import configs
            try:
                line["cpu_baseline"] = cpu_baseline(configs.ffhq_G_cfg(a.res, a.depth),
                                                    {"N_samples": a.n_samples, "perturb": True, "static_viewdirs": False}, B)
            except Exception as exc:           # noqa: BLE001 (the reported baseline must not cost the measured line)
                line["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        emit(line, a.detail)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
