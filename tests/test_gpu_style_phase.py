"""The style phase as ONE launch (cips3d_style_phase mode 1: resident workgroups that hand the mapping networks' layer outputs
to each other as tagged granules) against the same phase as a chain of launches (mode 0): bit-identical, launch after launch,
with other work on the queue, at every batch size, inside a captured graph.
Reference: models/model_v3.py:1299-1418 (mapping networks), :254,268 (modulation heads); cips3d/volume_renderer.py:66-67."""
import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _plan(res, B, D=2, seed=5):
    cfg = configs.ffhq_G_cfg(res, D)
    G = pkg.build_generator(cfg, DEV, seed=seed)
    plan = G._forward_plan(B, 64, 12, False)
    assert plan is not None
    return G, plan


def _outputs(plan):
    outs = [plan.styles_r, plan.styles_d, plan.film, plan.s_buf]
    return [o.clone() for o in outs]


def _poison(plan):
    for o in (plan.styles_r, plan.styles_d, plan.film, plan.s_buf):
        o.fill_(float("nan"))


@pytest.mark.parametrize("B", [1, 2, 4, 7])
@pytest.mark.parametrize("trunc", [1.0, 0.6])
def test_one_launch_equals_the_launches(B, trunc):
    G, plan = _plan(256, B)
    g = torch.Generator(device="cpu").manual_seed(B)
    z_r = torch.randn(B, G.z_dim, generator=g).to(DEV)
    z_d = torch.randn(B, G.z_dim, generator=g).to(DEV)
    mr = torch.randn(plan.plan.style_dim_r, generator=g).to(DEV) if trunc < 1 else None
    md = torch.randn(plan.plan.style_dim_d, generator=g).to(DEV) if trunc < 1 else None
    _poison(plan)
    plan.style_phase(z_r, z_d, mode=0, trunc_psi=trunc, mean_r=mr, mean_d=md)
    want = _outputs(plan)
    assert all(bool(torch.isfinite(w).all()) for w in want)
    _poison(plan)
    plan.style_phase(z_r, z_d, mode=1, trunc_psi=trunc, mean_r=mr, mean_d=md)
    got = _outputs(plan)
    for w, o in zip(want, got):
        assert torch.equal(w, o)
    assert int(plan.style_sync[1]) == 0


def test_draw_and_zeroing_ride_on_the_one_launch():
    G, plan = _plan(256, 1)
    assert plan.ranged
    z_r, z_d = torch.randn(1, G.z_dim, device=DEV), torch.randn(1, G.z_dim, device=DEV)
    n, u = 300_001, 4099
    res = []
    for mode in (0, 1):
        normal, uniform = torch.full((n,), 7.0, device=DEV), torch.full((u,), 7.0, device=DEV)
        plan.range_ws.fill_(3.0)
        plan.style_phase(z_r, z_d, mode=mode, rng=(1234, 77, normal, uniform))
        assert float(plan.range_ws.abs().max()) == 0.0
        res.append((normal, uniform))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[1][0].abs().max()) < 6.0 and 0.0 <= float(res[1][1].min()) and float(res[1][1].max()) < 1.0
    ref_n, ref_u = hip.rng_fill(n, u, DEV, seed=1234, base=77)
    assert torch.equal(ref_n, res[1][0]) and torch.equal(ref_u, res[1][1])


def test_two_thousand_launches_back_to_back_with_other_work_on_the_queue():
    """Every launch uses a new tag: a granule left by launch n must never satisfy launch n + 1.  New inputs every launch, a
    bandwidth-heavy kernel between launches (uneven load, L2 contents replaced), results checked on the device."""
    G, plan = _plan(256, 2)
    g = torch.Generator(device="cpu").manual_seed(0)
    zs = torch.randn(40, 2, 2, G.z_dim, generator=g).to(DEV)
    want = []
    for i in range(40):
        plan.style_phase(zs[i, 0], zs[i, 1], mode=0)
        want.append(_outputs(plan))
    junk = torch.empty(64 << 20, device=DEV)
    bad = torch.zeros((), device=DEV)
    for it in range(2000):
        i = (it * 7) % 40
        plan.style_phase(zs[i, 0], zs[i, 1], mode=1)
        for w, o in zip(want[i], (plan.styles_r, plan.styles_d, plan.film, plan.s_buf)):
            bad += (w != o).any()
        if it % 3 == 0:
            junk.add_(1.0)
    assert int(bad) == 0
    assert int(plan.style_sync[1]) == 0
    assert int(plan.style_sync[0]) >= 2000


def test_one_launch_in_a_captured_graph_replays():
    G, plan = _plan(256, 1)
    z_r, z_d = torch.randn(1, G.z_dim, device=DEV), torch.randn(1, G.z_dim, device=DEV)
    plan.style_phase(z_r, z_d, mode=0)
    want = _outputs(plan)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plan.style_phase(z_r, z_d, mode=1)      # warm
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            plan.style_phase(z_r, z_d, mode=1)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(5):
        _poison(plan)
        graph.replay()
        torch.cuda.synchronize()
        for w, o in zip(want, _outputs(plan)):
            assert torch.equal(w, o)
    assert int(plan.style_sync[1]) == 0


def test_deeper_renderer_and_full_size_decoder():
    """D = 8 (nine FiLM layers) with the 1024^2 decoder's 26 modulation heads: more head rows per wave than register slots."""
    G, plan = _plan(1024, 1, D=8)
    z_r, z_d = torch.randn(1, G.z_dim, device=DEV), torch.randn(1, G.z_dim, device=DEV)
    plan.style_phase(z_r, z_d, mode=0)
    want = _outputs(plan)
    _poison(plan)
    plan.style_phase(z_r, z_d, mode=1)
    for w, o in zip(want, _outputs(plan)):
        assert torch.equal(w, o)
    assert int(plan.style_sync[1]) == 0


def test_forward_takes_the_one_launch_on_request_and_matches_the_launches(monkeypatch):
    G, plan = _plan(256, 1)
    from cips_3dplusplus_amd import weights
    from cips_3dplusplus_amd.camera import Camera
    cfg = configs.ffhq_G_cfg(256, 2)
    zs, nb, _ = weights.synth_inputs(cfg, seed=4)
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=torch.tensor([[0.2, 0.05]], device=DEV))
    kw = dict(zs=[z.to(DEV) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[b.to(DEV) for b in nb],
              nerf_cfg=dict(N_samples=12, perturb=False, static_viewdirs=False))
    monkeypatch.delenv("CIPS3D_STYLE_PHASE", raising=False)
    gen0 = int(plan.style_sync[0])
    a = G(**kw)["rgb"].clone()
    assert int(plan.style_sync[0]) == gen0              # the default is the chain of launches (the faster one)
    monkeypatch.setenv("CIPS3D_STYLE_PHASE", "1")
    b = G(**kw)["rgb"].clone()
    assert int(plan.style_sync[0]) == gen0 + 1          # ... the one launch on request
    assert torch.equal(a, b)
