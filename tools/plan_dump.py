"""Which route every decoder layer of a plan takes: python tools/plan_dump.py [--res 1024] [--depth 2] [--batch 1] [--precision fp32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
G = pkg.build_generator(configs.ffhq_G_cfg(a.res, a.depth), "cuda", seed=0)
G.set_precision(a.precision)
pl = G._forward_plan(a.batch, 64, 24, False)
kinds = {0: "conv", 1: "conv(up)", 2: "torgb", 3: "torgb(up)"}
for i, li in enumerate(pl._layer_info):
    tags = [k for k in ("planes_in", "planes_out", "p16", "split", "split16", "chained", "flat_head") if li.get(k)]
    print(f"{i:2d} {kinds[li['kind']]:10s} {li['Cin']:4d} -> {li['Cout']:4d} @ {li['H']:4d}  {' '.join(tags)}")
print("u8_capable", pl.u8_capable, " out_res", pl.out_res)
