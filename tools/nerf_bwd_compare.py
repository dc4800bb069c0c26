"""Fused NeRF backward (csrc/nerf_bwd_fused.hip) against the materialised sequence (csrc/nerf_bwd.hip) on one shape:
max |difference| of d film and d cam_poses relative to their max-abs, and the time of each.
usage: python tools/nerf_bwd_compare.py [--hidden 256 --depth 6 --batch 2 --n-samples 24 --img 64 --iters 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cips_3dplusplus_amd as pkg  # noqa: E402
from cips_3dplusplus_amd import autograd as AG, configs, hip, weights  # noqa: E402
from cips_3dplusplus_amd.camera import Camera  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--n-samples", type=int, default=24)
    ap.add_argument("--img", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda"
    cfg = configs.tiny_G_cfg(a.hidden, a.depth, 1) if a.hidden < 256 else configs.ffhq_G_cfg(256, a.depth)
    G = pkg.build_generator(cfg, dev, seed=3)
    r = G.renderer
    B, S, N, H, D = a.batch, a.img, a.n_samples, r.hidden_dim, r.N_layers_renderer
    locs = torch.tensor([[0.25, 0.1], [-0.4, -0.05], [0.1, 0.2], [0.0, 0.0]])[:B].to(dev)
    cam, focal, near, far = Camera.generate_camera_params(locations=locs, img_size=S, device=dev, fov_ang=6, dist_radius=0.12)[:4]
    styles = (0.5 * weights.det_normal("cmp.styles", (B, D + 1, r.style_dim), 1.0, 1)).to(dev)
    film = AG.film_table(r, styles).detach()
    u = weights.det_unit_uniform("cmp.u", (B, S, S, 1), 2).to(dev)
    dF = (1e-5 * weights.det_normal("cmp.dF", (B, H, S, S), 1.0, 3)).to(dev)
    dT = (1e-4 * weights.det_normal("cmp.dT", (B, 3, S, S), 1.0, 4)).to(dev)
    packed, layer_bias = r._derived_buffers()
    args = (r.network, r.sigmoid_beta.detach(), cam, focal, near, far, u, film, layer_bias)
    res = {}
    for name in ("materialised", "fused"):
        def run():
            if name == "fused":
                return hip.nerf_backward_fused(*args, packed, r._packed_transposed(), S, N, False, dF, dT)
            return hip.nerf_backward(*args, S, N, False, dF, dT)
        out = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = run()
        torch.cuda.synchronize()
        res[name] = (out, (time.perf_counter() - t0) / a.iters * 1e3)
    (f0, c0), t0 = res["materialised"]
    (f1, c1), t1 = res["fused"]
    print(f"H={H} D={D} B={B} N={N} S={S}: materialised {t0:.3f} ms, fused {t1:.3f} ms")
    for nm, x, y in (("dfilm", f0, f1), ("dcam", c0, c1)):
        print(f"  {nm}: max|diff| {float((x - y).abs().max()):.3e}  max|ref| {float(x.abs().max()):.3e}  "
              f"rel {float((x - y).abs().max() / x.abs().max()):.2e}  finite {bool(torch.isfinite(y).all())}")
    for l in range(D + 1):
        d = (f0[:, l] - f1[:, l]).abs().max() / f0[:, l].abs().max()
        print(f"    layer {l}: rel {float(d):.2e}")


if __name__ == "__main__":
    main()
