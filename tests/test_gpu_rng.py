"""csrc/rng.hip (the fresh NoiseInjection maps and the per-ray jitter of one forward in one launch) against oracle/rng.py:
integer stream bit for bit, Box-Muller to the rounding of the hardware log2 / sine, ragged and unaligned outputs, the
distributions, and the coupling to torch's generator state (manual_seed reproduces a forward)."""
import numpy as np
import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import _lib, configs, hip, weights
from cips_3dplusplus_amd.camera import Camera
from oracle import rng as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("seed,base", [(0, 0), (0x123456789ABCDEF, (1 << 32) - 3), (2 ** 64 - 1, 2 ** 40 + 4)])
def test_integer_stream_is_the_oracles(seed, base):
    got = hip.rng_words(seed, base, 1000, DEV).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, O.raw_words(seed, base, 1000))


@pytest.mark.parametrize("n_normal,n_uniform", [(4096, 4096), (4099, 7), (3, 0), (0, 5), (1 << 20, 0), (17, 1 << 16)])
def test_fill_vs_oracle(n_normal, n_uniform):
    seed, base = 987654321012345, 4 * 12345
    n, u = hip.rng_fill(n_normal, n_uniform, DEV, seed=seed, base=base)
    torch.cuda.synchronize()
    rn, ru = O.fill(seed, base, n_normal, n_uniform)
    if n_normal:
        # fp32 Box-Muller on v_log_f32 / v_sqrt_f32 / v_sin_f32 against the oracle's float64: absolute, |r| < 5.9
        assert np.abs(n.cpu().numpy().astype(np.float64) - rn).max() < 2e-5
    else:
        assert n is None
    if n_uniform:
        assert np.array_equal(u.cpu().numpy(), ru)
    else:
        assert u is None
    assert int(_lib.load().cips3d_rng_fill_threads(n_normal, n_uniform)) == (n_normal + 3) // 4 + (n_uniform + 3) // 4


def test_unaligned_outputs_and_untouched_neighbours():
    lib = _lib.load()
    buf = torch.full((64,), -7.0, device=DEV)
    ubuf = torch.full((32,), -7.0, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.cips3d_rng_fill(5, 8, buf.data_ptr() + 4, 50, ubuf.data_ptr() + 12, 21, st), "rng")     # 4- and 12-byte offsets
    rn, ru = O.fill(5, 8, 50, 21)
    b, ub = buf.cpu().numpy(), ubuf.cpu().numpy()
    assert np.abs(b[1:51].astype(np.float64) - rn).max() < 2e-5 and b[0] == -7.0 and (b[51:] == -7.0).all()
    assert np.array_equal(ub[3:24], ru) and (ub[:3] == -7.0).all() and (ub[24:] == -7.0).all()
    assert lib.cips3d_rng_fill(5, 8, None, 4, None, 0, st) != 0 and lib.cips3d_rng_fill(5, 8, None, 0, None, 0, st) == 0
    assert lib.cips3d_rng_fill(5, 8, buf.data_ptr(), -1, None, 0, st) != 0


def test_distributions_on_device():
    n, u = hip.rng_fill(1 << 22, 1 << 20, DEV, seed=42, base=0)
    nd = n.double()
    assert abs(float(nd.mean())) < 2e-3 and abs(float(nd.std()) - 1.0) < 2e-3
    assert abs(float((nd ** 3).mean())) < 6e-3 and abs(float((nd ** 4).mean()) - 3.0) < 2e-2
    assert bool(torch.isfinite(n).all()) and float(n.abs().max()) < 6.0
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and abs(float(u.double().mean()) - 0.5) < 2e-3
    from scipy import stats
    assert stats.kstest(n[:400000].cpu().numpy(), "norm").statistic < 3e-3
    assert stats.kstest(u[:400000].cpu().numpy(), "uniform").statistic < 3e-3
    # the noise maps of a forward are slices of one draw: different slices are uncorrelated
    a, b = nd[: 1 << 20], nd[1 << 20: 1 << 21]
    assert abs(float(((a - a.mean()) * (b - b.mean())).mean())) < 4e-3


def test_torch_generator_state_governs_the_stream():
    torch.manual_seed(1234)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    off0 = gen.get_offset()
    a, ua = hip.rng_fill(1000, 10, DEV)
    assert gen.get_offset() == off0 + 4 * ((250 + 3 + 3) // 4)             # consumed counters, rounded to torch's multiple of 4
    b, _ = hip.rng_fill(1000, 10, DEV)
    assert not torch.equal(a, b)                                           # the offset advanced: a new draw
    t = torch.randn(8, device=DEV)                                         # torch's own draws go on from the advanced state
    torch.manual_seed(1234)
    a2, ua2 = hip.rng_fill(1000, 10, DEV)
    b2, _ = hip.rng_fill(1000, 10, DEV)
    assert torch.equal(a, a2) and torch.equal(ua, ua2) and torch.equal(b, b2) and torch.equal(t, torch.randn(8, device=DEV))
    state = torch.cuda.get_rng_state()
    c = hip.rng_fill(64, 0, DEV)[0]
    torch.cuda.set_rng_state(state)
    assert torch.equal(c, hip.rng_fill(64, 0, DEV)[0])
    torch.manual_seed(1235)
    assert not torch.equal(a, hip.rng_fill(1000, 10, DEV)[0])


def test_forward_with_fresh_noise_and_jitter_is_reproducible_and_matches_torchs_statistics():
    """The bench's loop body (test__rendering_time: perturb=True, fresh noise): one cips3d_rng_fill per forward; the same
    manual_seed gives the same image; the image's statistics agree with the torch.randn / torch.rand form of the same path."""
    cfg = configs.ffhq_G_cfg(256, 2)
    G = pkg.build_generator(cfg, DEV, seed=3)
    g = torch.Generator(device=DEV).manual_seed(11)
    zs = [torch.randn(2, 256, device=DEV, generator=g), torch.randn(2, 256, device=DEV, generator=g)]
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=torch.tensor([[0.2, -0.1], [-0.3, 0.05]], device=DEV))
    kw = dict(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa,
              nerf_cfg=dict(N_samples=12, perturb=True, static_viewdirs=False))
    old = hip.FAST_RNG
    try:
        hip.FAST_RNG = True
        torch.manual_seed(77)
        a = G(**kw)["rgb"].clone()
        b = G(**kw)["rgb"].clone()
        torch.manual_seed(77)
        a2 = G(**kw)["rgb"].clone()
        assert torch.equal(a, a2) and not torch.equal(a, b)
        hip.FAST_RNG = False
        torch.manual_seed(77)
        t = G(**kw)["rgb"].clone()
        assert not torch.equal(t, a)
        # same network, same inputs, independent noise draws: means and spreads of the images agree
        assert abs(float(a.mean()) - float(t.mean())) < 0.02 * float(t.abs().max())
        assert abs(float(a.std()) - float(t.std())) < 0.05 * float(t.std())
        # explicit noise + explicit jitter: the generator is not touched
        nb = [torch.randn(1, 1, s, s, device=DEV) for s in G.decoder.noise_sizes(64)] if hasattr(G.decoder, "noise_sizes") else \
            G.create_noise_bufs(64, DEV)
        u = torch.rand(2, 64, 64, 1, device=DEV)
        hip.FAST_RNG = True
        gen = torch.cuda.default_generators[torch.cuda.current_device()]
        off = gen.get_offset()
        x = G(**kw, noise_bufs=nb, perturb_u=u)["rgb"].clone()
        assert gen.get_offset() == off
        hip.FAST_RNG = False
        assert torch.equal(x, G(**kw, noise_bufs=nb, perturb_u=u)["rgb"])
    finally:
        hip.FAST_RNG = old


@pytest.mark.parametrize("given_styles", [False, True])
def test_the_forwards_own_draw_is_cips3d_rng_fill(given_styles):
    """cips3d_forward_io.rng_*: the one-call forward makes the draw itself, its threads spread over the mapping networks'
    launches (or as a launch of its own when the caller passes W+ styles and there are none) -- the values are those of one
    cips3d_rng_fill(seed, base) whatever the slicing: the image equals the forward on explicit noise / jitter cut from it."""
    cfg = configs.ffhq_G_cfg(256, 2)
    G = pkg.build_generator(cfg, DEV, seed=5)
    B, S = 3, 64
    g = torch.Generator(device=DEV).manual_seed(21)
    zs = [torch.randn(B, 256, device=DEV, generator=g), torch.randn(B, 256, device=DEV, generator=g)]
    e, f, n, fa, _ = Camera.generate_camera_params(S, DEV, locations=0.2 * torch.randn(B, 2, device=DEV, generator=g))
    kw = dict(zs=zs, cam_poses=e, focals=f, img_size=S, near=n, far=fa, nerf_cfg=dict(N_samples=8, perturb=True, static_viewdirs=False))
    if given_styles:
        s_r, s_d = G.mapping_networks(zs=zs, truncation=1, inject_index=None)
        kw.update(zs=[None, None], style_render=s_r, style_decoder=s_d)
    old = hip.FAST_RNG
    try:
        hip.FAST_RNG = True
        torch.manual_seed(909)
        gen = torch.cuda.default_generators[torch.cuda.current_device()]
        torch.randn(5, device=DEV)                        # a non-zero offset to start from
        seed, base = gen.initial_seed(), gen.get_offset()
        a = G(**kw)
        plan = list(G._plans.values())[0]
        total = B * plan.noise_total
        assert gen.get_offset() == base + 4 * (((total + 3) // 4 + (B * S * S + 3) // 4 + 3) // 4)
        normal, uniform = hip.rng_fill(total, B * S * S, DEV, seed=seed, base=base)
        nb, off = [], 0
        for s in plan.noise_sizes:
            nb.append(normal[off:off + B * s * s].view(B, 1, s, s).contiguous())
            off += B * s * s
        # (the plan would MEASURE the bound of caller-supplied noise, 5.x here, where the call that draws for itself uses the
        # generator's a-priori 6: a different bound may move a planes exponent, and images then agree to rounding, not bit for
        # bit.  The claim under test is about the drawn VALUES: give both calls the same bound.)
        plan._noise_bound = lambda *args: hip.NOISE_BOUND_RNG
        try:
            b = G(**kw, noise_bufs=nb, perturb_u=uniform.view(B, S, S, 1))
        finally:
            del plan._noise_bound
        assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["thumb_rgb"], b["thumb_rgb"])
        c = G(**kw, noise_bufs=nb, perturb_u=uniform.view(B, S, S, 1))         # with the measured bound: equal to rounding
        assert float((c["rgb"] - a["rgb"]).abs().max()) <= 2e-6 * float(a["rgb"].abs().max())
    finally:
        hip.FAST_RNG = old
