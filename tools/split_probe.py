"""CPU experiment behind the split-fp16 MFMA arithmetic of csrc/nerf.hip: the oracle generator with every FiLM-SIREN GEMM
evaluated as w_hi x_hi + w_hi x_lo + w_lo x_hi (fp16 halves of power-of-two-scaled weights / activations, fp32 sums) against
plain fp32 and against an fp64 run: the split form sits exactly where fp32 sits relative to fp64.

    python tools/split_probe.py        (CPU, ~1 min)
"""
import sys, math, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from oracle import path as O
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, weights

def split16(t):
    hi = t.half().float()
    lo = (t - hi).half().float()
    return hi, lo

MODE = {"on": False, "terms": 3}
orig_affine = O._affine
def film_siren_split(sd, prefix, x, style):
    B = style.shape[0]
    W, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if MODE["on"] and x.shape[-1] >= 32:
        K = W.shape[1]
        Km = K if K % 32 == 0 else K - 3          # view layer: last 3 inputs (view dirs) stay fp32
        s = 2.0 ** math.floor(math.log2(1024.0 / float(W[:, :Km].abs().max())))
        Wh, Wl = split16(W[:, :Km] * s)
        xh, xl = split16(x[..., :Km])
        pre = (F.linear(xh, Wh) + F.linear(xl, Wh) + F.linear(xh, Wl))
        if MODE["terms"] == 4:
            pre = pre + F.linear(xl, Wl)
        pre = pre / s
        if Km < K:
            pre = pre + F.linear(x[..., Km:], W[:, Km:])
        pre = pre + b
    else:
        pre = F.linear(x, W, b)
    bshape = [B] + [1] * (pre.dim() - 2) + [-1]
    gamma = (15.0 * orig_affine(sd, prefix + ".gamma", style) + 30.0).view(*bshape)
    beta = (0.25 * orig_affine(sd, prefix + ".beta", style)).view(*bshape)
    return torch.sin(gamma * pre + beta)
O.film_siren = film_siren_split

torch.set_num_threads(8)
for D in (2, 8):
    cfg = configs.ffhq_G_cfg(256, D)
    G = pkg.Generator(**cfg)
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=1)
    zs, nb, means = weights.synth_inputs(cfg, seed=12345)
    cam = O.camera_params(torch.tensor([[0.31, -0.08]]), 64, 6, 0.12)
    ncfg = dict(N_samples=24, perturb=False, static_viewdirs=False)
    outs = {}
    for tag, on, dt in (("f32", False, torch.float32), ("split", True, torch.float32), ("f64", False, torch.float64)):
        MODE["on"] = on
        sdd = {k: v.to(dt) if v.is_floating_point() else v for k, v in sd.items()}
        with torch.no_grad():
            r = O.generator_forward(sdd, cfg, [z.to(dt) for z in zs], *[c.to(dt) for c in cam[:2]], 64, cam[2].to(dt), cam[3].to(dt), ncfg, [b.to(dt) for b in nb], return_xyz=True)
        outs[tag] = r
    for k in ("rgb", "thumb_rgb", "_features", "xyz"):
        a, b, c = outs["f32"][k].double(), outs["split"][k].double(), outs["f64"][k].double()
        print(f"D={D} {k:10s} |f32-f64| {float((a-c).abs().max()):.3e}  |split-f64| {float((b-c).abs().max()):.3e}  |split-f32| {float((a-b).abs().max()):.3e}  range {float(c.abs().max()):.2f}")
