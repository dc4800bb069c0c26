"""Stand-alone timing of the fused NeRF backward at BASELINE config 5's shape (hip.inversion_roofline)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, hip
G = pkg.build_generator(configs.ffhq_G_cfg(256, 6), "cuda", seed=0)
r = hip.inversion_roofline(G.renderer, B=2, n_samples=24, iters=20)
print(os.environ.get("CIPS3D_HIPCC_FLAGS", ""), json.dumps({k: r[k] for k in ("avg_launch_ms", "frac")}))
