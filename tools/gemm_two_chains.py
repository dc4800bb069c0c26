"""Experiment: the nine dependent 512->512 1x1 convs at 64^2 (batch 1) as ONE chain of full-width GEMMs versus TWO
independent half-width chains (a 1x1 conv couples a pixel only to itself, so the left and right halves of the image
never meet) on two streams, so that one chain's launch boundary / DMA ramp / store tail runs under the other's MFMAs.

    [CIPS3D_GEMM_CFG=7] python tools/gemm_two_chains.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips_3dplusplus_amd import hip

dev = "cuda"
L, C, HW = 9, 512, 4096
torch.manual_seed(0)
wm = [hip.modulate_weights(torch.randn(1, C, C, 1, 1, device=dev), torch.rand(1, C, device=dev) + 0.5, C, 1, C, C, 1,
                           1.0 / C ** 0.5, True, True) for _ in range(L)]
bias = torch.zeros(C, device=dev)
nw = torch.full((1,), 0.1, device=dev)


def chain(x, bufs, noise):
    for l in range(L):
        x = hip.modconv1x1(x, wm[l], C, epilogue=1, noise=noise, noise_w=nw, bias=bias, out=bufs[l & 1])
    return x


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


x_full = torch.randn(1, C, 64, 64, device=dev)
n_full = torch.randn(1, 1, 64, 64, device=dev)
b_full = [torch.empty(1, C, 64, 64, device=dev) for _ in range(2)]
print(f"one chain, full width (4096 px): {timeit(lambda: chain(x_full, b_full, n_full)):8.1f} us per 9 layers")

halves = []
for h in range(2):
    halves.append((torch.randn(1, C, 32, 64, device=dev), [torch.empty(1, C, 32, 64, device=dev) for _ in range(2)],
                   torch.randn(1, 1, 32, 64, device=dev)))
print(f"one chain, half width (2048 px): {timeit(lambda: chain(*halves[0])):8.1f} us per 9 layers")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two(offset_layers):
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        chain(*halves[0])
        e1 = torch.cuda.Event(); e1.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        for _ in range(offset_layers):          # a delay: dummy half-layer work ahead of the second chain
            hip.modconv1x1(halves[1][0], wm[0], C, epilogue=0, out=halves[1][1][1])
        chain(*halves[1])
        e2 = torch.cuda.Event(); e2.record()
    torch.cuda.current_stream().wait_event(e1)
    torch.cuda.current_stream().wait_event(e2)


for off in (0, 1):
    print(f"two half-width chains on two streams (second delayed by {off} extra launch): {timeit(lambda: two(off)):8.1f} us")

# the same as HIP graphs (no host launch cost in the way)
def graphed(fn):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g.replay


g_full = graphed(lambda: chain(x_full, b_full, n_full))
print(f"graph: one chain, full width: {timeit(g_full):8.1f} us")
g_half = graphed(lambda: chain(*halves[0]))
print(f"graph: one chain, half width: {timeit(g_half):8.1f} us")


def two_in_capture(offset_layers):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        chain(*halves[0])
    with torch.cuda.stream(s2):
        for _ in range(offset_layers):
            hip.modconv1x1(halves[1][0], wm[0], C, epilogue=0, out=halves[1][1][1])
        chain(*halves[1])
    cur.wait_stream(s1); cur.wait_stream(s2)


for off in (0, 1):
    g2 = graphed(lambda: two_in_capture(off))
    print(f"graph: two half-width chains (second delayed by {off} extra launch): {timeit(g2):8.1f} us")
