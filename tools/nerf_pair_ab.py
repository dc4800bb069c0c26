"""Render kernel A/B on one box: the 32-points-per-wave kernel (csrc/nerf_pair.hip) against the 16-point kernel
(csrc/nerf.hip, CIPS3D_NERF_PAIR=0) -- max-abs differences of every output map and the median kernel time of each.
    python tools/nerf_pair_ab.py [--depth D] [--n-samples N] [--batch B] [--iters K]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, hip
from cips_3dplusplus_amd.camera import Camera

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--n-samples", type=int, default=24)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--no-perturb", action="store_true")
ap.add_argument("--ws", action="store_true", help="compare the weight-stationary kernel (csrc/nerf_ws.hip, CIPS3D_NERF_WS=1) instead")
ap.add_argument("--abl", default="", help="comma list of CIPS3D_PAIR_ABL values to time as well (a -DCIPS3D_PAIR_ABLATIONS build)")
a = ap.parse_args()
dev = "cuda"
G = pkg.build_generator(configs.ffhq_G_cfg(256, a.depth), dev, seed=0)
B = a.batch
loc = torch.tensor([[0.3, -0.1]] * B, device=dev) * torch.linspace(1, 2, B, device=dev)[:, None]
e, f, n, fa, _ = Camera.generate_camera_params(64, dev, locations=loc)
torch.manual_seed(0)
styles = torch.randn(B, a.depth + 1, 256, device=dev)
pu = None if a.no_perturb else torch.rand(B, 64 * 64, device=dev)


def run(pair):
    os.environ["CIPS3D_NERF_WS" if a.ws else "CIPS3D_NERF_PAIR"] = "1" if pair else "0"
    out = G.renderer.render(e, f, n, fa, styles, 64, a.n_samples, perturb_u=pu, return_sdf=True)
    out = dict(zip(("thumb", "features", "sdf", "mask", "xyz"), out))
    for _ in range(3):
        G.renderer.render(e, f, n, fa, styles, 64, a.n_samples)
    torch.cuda.synchronize()
    hip.KERNEL_EVENTS["nerf_render"] = []
    for _ in range(a.iters):
        G.renderer.render(e, f, n, fa, styles, 64, a.n_samples)
    torch.cuda.synchronize()
    ev = hip.KERNEL_EVENTS.pop("nerf_render")
    ts = sorted(s.elapsed_time(t) for s, t in ev)
    return out, ts[len(ts) // 2] * 1e3, ts[0] * 1e3


ref, t_ref, m_ref = run(False)
new, t_new, m_new = run(True)
print(f"D={a.depth} N={a.n_samples} B={B}: 16-point kernel {t_ref:.1f} us (min {m_ref:.1f})   {'weight-stationary' if a.ws else 'pair'} kernel {t_new:.1f} us (min {m_new:.1f})")
for ab in [x for x in a.abl.split(",") if x]:
    os.environ["CIPS3D_PAIR_ABL"] = ab
    _, t_ab, m_ab = run(True)
    print(f"  ablation {ab:>2s} (1 no DMA, 2 no epilogue, 4 no layer 0, 8 no MFMA): {t_ab:.1f} us (min {m_ab:.1f})")
os.environ.pop("CIPS3D_PAIR_ABL", None)
names = ref.keys() if isinstance(ref, dict) else range(len(ref))
bad = False
for k in names:
    x, y = ref[k], new[k]
    if not torch.is_tensor(x):
        continue
    d = (x.float() - y.float()).abs().max().item()
    r = x.float().abs().max().item()
    print(f"  {str(k):12s} shape {tuple(x.shape)}  max|16pt| {r:.4f}  max|diff| {d:.3e}  finite {bool(torch.isfinite(y.float()).all())}")
    bad |= not (d <= 2e-4 * max(r, 1.0))
print("MISMATCH" if bad else "match")
sys.exit(1 if bad else 0)
