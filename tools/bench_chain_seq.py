"""What does ONE hop of a persistent planes chain cost?  

N consecutive 512 -> 512 StyledConv GEMMs at 64^2 on split-fp16 planes (the shape of the release decoder's run, csrc/chain.hip)
  (a) as N launches of cips3d_modconv1x1_planes, the shipped form;
  (b) as ONE launch of cips3d_modconv1x1_planes_seq (chain_seq_kernel: a per-pixel-block counter between layers), with and without
      the next layer's first weight stage requested in front of the poll;
results compared bit for bit (planes, exponents, patch maxima), both forms timed with HIP events over the same preallocated buffers,
interleaved in one process.  hop = (t_seq(N) - t_seq(1)) / (N - 1) - (matrix time per layer: t_seq(1) is one layer in the seq
kernel); boundary = (t_launches(N) - t_launches(1)) / (N - 1) - t_launches(1).

    python tools/bench_chain_seq.py [--layers 2 4 9] [--batch 1] [--iters 200] [--json FILE]"""
import argparse, ctypes as C, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import _lib, hip, weights

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, nargs="+", default=[1, 2, 4, 9])
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--json", default=None)
a = ap.parse_args()
dev, B, Cc, S = "cuda", a.batch, 512, 64
HW = S * S
lib = _lib.load()
NMAX = max(a.layers)
cu = lambda t: t.to(dev).contiguous()          # noqa: E731


def make_layer(l):
    W = cu(weights.det_normal(f"seq.W{l}", (Cc, Cc), 1.0, l))
    s = cu(1.0 + weights.det_uniform(f"seq.s{l}", (B, Cc), 0.3, 100 + l))
    wm = hip.modulate_weights(W, s, Cc, B, Cc, Cc, 1, 1.0 / Cc ** 0.5, True, True, split=True)
    noise = cu(weights.det_normal(f"seq.n{l}", (1, 1, S, S), 1.0, 200 + l))
    nw = torch.full((1,), 0.1, device=dev)
    bias = cu(weights.det_uniform(f"seq.b{l}", (Cc,), 0.2, 300 + l))
    lconst = hip.range_consts(B, bias, nw, Cc ** 0.5, noise=noise)
    return dict(wm=wm, noise=noise, nw=nw, bias=bias, lconst=lconst)


L = [make_layer(l) for l in range(NMAX)]
x0 = hip.to_planes(cu(weights.det_normal("seq.x", (B, Cc, S, S), 1.0, 7)))
nblk = (HW + 127) // 128


def buffers():
    """outputs of every layer (planes, exponents, patch maxima)"""
    return [dict(out=torch.empty(B, Cc // 8, 2, HW, 8, device=dev, dtype=torch.float16),
                 exp=torch.zeros(B, nblk, device=dev, dtype=torch.int32),
                 pmax=torch.zeros(B, (HW + 63) // 64, Cc // 16, device=dev)) for _ in range(NMAX)]


def layer_struct(l, bufs):
    x, x_exp, x_pmax = (x0, x0.cips3d_exp, x0.cips3d_pmax) if l == 0 else (bufs[l - 1]["out"], bufs[l - 1]["exp"], bufs[l - 1]["pmax"])
    P = _lib.PlanesLayer()
    P.x_planes, P.wm, P.out = x.data_ptr(), L[l]["wm"].data_ptr(), bufs[l]["out"].data_ptr()
    P.out_format, P.Cin, P.Cout, P.epilogue = 1, Cc, Cc, 1
    P.noise, P.noise_bstride, P.noise_w, P.bias = L[l]["noise"].data_ptr(), 0, L[l]["nw"].data_ptr(), L[l]["bias"].data_ptr()
    P.rg.x_exp, P.rg.x_pmax, P.rg.lconst = x_exp.data_ptr(), x_pmax.data_ptr(), L[l]["lconst"].data_ptr()
    P.rg.out_exp, P.rg.out_pmax = bufs[l]["exp"].data_ptr(), bufs[l]["pmax"].data_ptr()
    return P


bufA, bufB = buffers(), buffers()
structsA = [layer_struct(l, bufA) for l in range(NMAX)]
structsB = (_lib.PlanesLayer * NMAX)(*[layer_struct(l, bufB) for l in range(NMAX)])
sync = torch.zeros(B, nblk, 2, device=dev, dtype=torch.int32)
fault = torch.zeros(1, device=dev, dtype=torch.int32)
st = lambda: _lib.stream_ptr()          # noqa: E731


def run_launches(n):
    for l in range(n):
        P = structsA[l]
        _lib.check(lib.cips3d_modconv1x1_planes(P.x_planes, P.wm, P.out, 1, B, Cc, Cc, HW, 1, P.noise, 0, P.noise_w, P.bias, None, None,
                                                None, C.byref(P.rg), st()), "cips3d_modconv1x1_planes")


def run_seq(n, flags):
    _lib.check(lib.cips3d_modconv1x1_planes_seq(structsB, n, B, HW, sync.data_ptr(), fault.data_ptr(), flags, st()),
               "cips3d_modconv1x1_planes_seq")


# ---- parity: every layer's outputs, bit for bit
ok = True
for flags in (0, 1):
    for b_ in bufB:
        b_["out"].zero_(); b_["exp"].zero_(); b_["pmax"].zero_()
    run_launches(NMAX)
    run_seq(NMAX, flags)
    torch.cuda.synchronize()
    assert int(fault.item()) == 0, "a hop's poll gave up"
    assert int(sync.abs().sum().item()) == 0, "the kernel left its counters non-zero"
    for l in range(NMAX):
        same = all(torch.equal(bufA[l][k], bufB[l][k]) for k in ("out", "exp", "pmax"))
        ok &= same
        if not same:
            print(f"flags {flags} layer {l}: DIFFERENT  planes {int((bufA[l]['out'] != bufB[l]['out']).sum())} values, "
                  f"exp {int((bufA[l]['exp'] != bufB[l]['exp']).sum())}, pmax {int((bufA[l]['pmax'] != bufB[l]['pmax']).sum())}")
print("parity (all layers, planes + exponents + patch maxima, both flag settings):", "bit-identical" if ok else "MISMATCH")


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3          # us per call


res = {}
variants = [("launches", lambda n: run_launches(n)), ("seq", lambda n: run_seq(n, 0)), ("seq_prefetch", lambda n: run_seq(n, 1))]
for _ in range(a.rounds):
    for n in a.layers:
        for name, fn in variants:
            res.setdefault((name, n), []).append(timed(lambda: fn(n)))
med = {k: statistics.median(v) for k, v in res.items()}
print(f"512 -> 512 at 64^2, batch {B}; us per call, median of {a.rounds} rounds x {a.iters} calls")
print("layers   " + "".join(f"{name:>16s}" for name, _ in variants))
for n in a.layers:
    print(f"{n:6d}   " + "".join(f"{med[(name, n)]:16.2f}" for name, _ in variants))
out = {"shape": f"512->512 @64^2 B{B}", "us_per_call": {f"{k[0]}_{k[1]}": v for k, v in med.items()}, "parity_bit_identical": bool(ok)}
if 1 in a.layers and len(a.layers) > 1:
    n = max(a.layers)
    per = {name: (med[(name, n)] - med[(name, 1)]) / (n - 1) for name, _ in variants}
    print(f"per additional layer (from {n} vs 1 layers): " + ", ".join(f"{k} {v:.2f} us" for k, v in per.items()))
    print(f"=> a hop costs {per['seq'] - per['launches']:+.2f} us more than a launch boundary ({per['seq_prefetch'] - per['launches']:+.2f} with the "
          f"weight prefetch); one layer alone: launches {med[('launches', 1)]:.2f}, seq kernel {med[('seq', 1)]:.2f}")
    out["per_additional_layer_us"] = per
    out["hop_minus_boundary_us"] = {"seq": per["seq"] - per["launches"], "seq_prefetch": per["seq_prefetch"] - per["launches"]}
if a.json:
    with open(a.json, "a") as fh:
        fh.write(json.dumps(out) + "\n")
assert int(fault.item()) == 0
