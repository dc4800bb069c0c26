"""Run one piece of the path in isolation (for rocprofv3 counter passes).
    python tools/run_kernel.py nerf|gemm64|forward [--iters N] [--depth D] [--n-samples N]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, hip
from cips_3dplusplus_amd.camera import Camera

ap = argparse.ArgumentParser()
ap.add_argument("what")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--n-samples", type=int, default=24)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32_exact"])
ap.add_argument("--clock-json", default=None, help="-DCIPS3D_CLOCK build: append the in-kernel clock record to this file")
a = ap.parse_args()
dev = "cuda"
cfg = configs.ffhq_G_cfg(a.res, a.depth)
G = pkg.build_generator(cfg, dev, seed=0)
G.set_precision(a.precision)
B = a.batch
e, f, n, fa, _ = Camera.generate_camera_params(64, dev, locations=torch.zeros(B, 2, device=dev))
torch.manual_seed(0)
if a.what == "nerf":
    styles = torch.randn(B, a.depth + 1, 256, device=dev)
    fn = lambda: G.renderer.render(e, f, n, fa, styles, 64, a.n_samples)
elif a.what == "gemm64":
    import cips_3dplusplus_amd.decoder as dec
    sc = G.decoder.convs[0]
    x = torch.randn(B, 512, 64, 64, device=dev)
    st = torch.randn(B, 512, device=dev)
    nz = torch.randn(1, 1, 64, 64, device=dev)
    wm = sc.conv.modulated_weight(st, packed=True)
    fn = lambda: sc(x, st, noise=nz, wm=wm)
else:
    zs = [torch.randn(B, 256, device=dev), torch.randn(B, 256, device=dev)]
    ncfg = dict(N_samples=a.n_samples, perturb=True, static_viewdirs=False)
    fn = lambda: G(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, nerf_cfg=ncfg)
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    fn()
torch.cuda.synchronize()
print(f"{a.what}: {(time.perf_counter() - t0) / a.iters * 1e3:.3f} ms/iter (host wall)")
if a.what == "nerf":
    hip.KERNEL_EVENTS["nerf_render"] = []
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    ev = hip.KERNEL_EVENTS.pop("nerf_render")
    ts = sorted(s.elapsed_time(t) for s, t in ev)
    print(f"  nerf_render kernel: median {ts[len(ts)//2]*1e3:.1f} us  min {ts[0]*1e3:.1f} us")
if a.what == "nerf":
    import ctypes
    lib = ctypes.CDLL(pkg._lib.LIB_PATH) if hasattr(pkg, "_lib") else None
    from cips_3dplusplus_amd import _lib as L
    raw = ctypes.CDLL(L.LIB_PATH)
    if hasattr(raw, "cips3d_debug_read_stamps"):
        buf = (ctypes.c_ulonglong * 16)()
        raw.cips3d_debug_read_stamps(buf)
        fn(); 
        raw.cips3d_debug_read_stamps(buf)
        nwg = 256 * a.batch
        names = ["prologue", "setup+layer0", "hidden", "sigma+weight", "view", "tail", "finish"]
        tot = sum(buf[:7])
        for nme, v in zip(names, buf[:7]):
            print(f"  {nme:14s} {v / nwg:10.0f} cycles/wg  {100.0 * v / tot:5.1f}%")
        if sum(buf[8:13]):
            print("  inside the MFMA layers (per slab step, same wave; the coarse 'hidden'/'view' rows above then only hold what follows the last step):")
            steps = 24 * nwg          # 3 samples x 2 layers x 4 steps at the default shape
            for nme, v in zip(["dma issue", "matrix block", "barrier (late waves)", "epilogue", "barrier (early waves)"], buf[8:13]):
                print(f"    {nme:22s} {v / steps:8.0f} cycles/step")

if a.what in ("nerf", "forward"):
    import ctypes
    from cips_3dplusplus_amd import _lib as L
    raw = ctypes.CDLL(L.LIB_PATH)
    if hasattr(raw, "cips3d_debug_read_clock"):          # -DCIPS3D_CLOCK build: shader clock inside the render kernel
        t_end = time.perf_counter() + 2.5
        while time.perf_counter() < t_end:               # >= 2 s of back-to-back launches first
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 2)()
        raw.cips3d_debug_read_clock(buf)                 # zero the sums
        for _ in range(300):
            fn()
        raw.cips3d_debug_read_clock(buf)
        print(f"  in-kernel shader clock of nerf_render_kernel ({a.what} loop, {a.precision}): {buf[0] / buf[1] * 100:.0f} MHz "
              f"(sum over workgroups: {buf[0]} shader cycles / {buf[1]} ticks of the 100 MHz reference)")
        rec = {"loop": a.what, "precision": a.precision, "n_samples": a.n_samples, "depth": a.depth, "batch": a.batch,
               "launches_summed": 300, "MHz_sum_over_workgroups": buf[0] / buf[1] * 100,
               "cycles_per_workgroup_mean": buf[0] / 300 / (256 * a.batch)}
        if hasattr(raw, "cips3d_debug_read_clock_wg"):
            import statistics
            nwg = 256 * a.batch
            wg = (ctypes.c_ulonglong * (2 * nwg))()
            raw.cips3d_debug_read_clock_wg(wg, nwg)
            mhz = sorted(wg[2 * i] / wg[2 * i + 1] * 100 for i in range(nwg) if wg[2 * i + 1])
            rec.update(MHz_median_over_workgroups=statistics.median(mhz), MHz_p05=mhz[len(mhz) // 20], MHz_p95=mhz[-1 - len(mhz) // 20],
                       workgroups=len(mhz), stamped_ticks_median=statistics.median(wg[2 * i + 1] for i in range(nwg)),
                       stamped_cycles_median=statistics.median(wg[2 * i] for i in range(nwg)))
            print(f"  median over {len(mhz)} workgroups of the last launch: {rec['MHz_median_over_workgroups']:.0f} MHz "
                  f"(p05 {rec['MHz_p05']:.0f}, p95 {rec['MHz_p95']:.0f}); stamped span {rec['stamped_ticks_median'] * 10:.0f} ns, "
                  f"{rec['stamped_cycles_median']:.0f} shader cycles")
        if a.clock_json:
            import json
            with open(a.clock_json, "a") as fh:
                fh.write(json.dumps(rec) + "\n")
