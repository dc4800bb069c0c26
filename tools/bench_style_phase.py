"""Style phase alone: one launch (cips3d_style_phase mode 1) against the launches (mode 0), HIP-event timed.
usage: python tools/bench_style_phase.py [res] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
G = pkg.build_generator(configs.ffhq_G_cfg(res, 2), "cuda", seed=0)
plan = G._forward_plan(B, 64, 24, False)
z_r, z_d = torch.randn(B, G.z_dim, device="cuda"), torch.randn(B, G.z_dim, device="cuda")
n = B * plan.noise_total
normal, uniform = torch.empty(n, device="cuda"), torch.empty(B * 4096, device="cuda")
junk = torch.empty(32 << 20, device="cuda")
for rng in (None, (1, 0, normal, uniform)):
    for mode in (0, 1, 0, 1):
        for _ in range(20):
            plan.style_phase(z_r, z_d, mode=mode, rng=rng)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            junk.add_(1.0)        # the weights do not stay in L2 between forwards either
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                plan.style_phase(z_r, z_d, mode=mode, rng=rng)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"rng={'yes' if rng else 'no '} mode={mode}: {min(ts):7.2f} us per phase (median {sorted(ts)[2]:.2f})", flush=True)
print("timeouts:", int(plan.style_sync[1]), "generation:", int(plan.style_sync[0]))
st = plan.style_sync.cpu().tolist()
if st[3]:       # a -DCIPS3D_SP_STAMPS build (CIPS3D_HIPCC_FLAGS): workgroup 0's timeline
    print("shader clock over workgroup 0's run: %.0f MHz" % (st[24] / max(st[3], 1) * 100))
    print("mod heads: begin, partial sums, reductions", [st[21 + k] / 100 for k in range(3)])
    print("workgroup 0, us from its start: mod heads done", st[20] / 100, "end", st[3] / 100,
          " per stage (inputs staged, rows published):", [(st[4 + 2 * l] / 100, st[5 + 2 * l] / 100) for l in range(7)])
