"""GPU parity against fixtures the REFERENCE wrote (tests/golden/make_golden.py) that the first round only checked on the
CPU oracle: ray / sample geometry (rays.npz), FiLM-SIREN layers and the compositing edge rays (siren.npz: alpha -> 1 at
the first sample, background-only ray on the 1e10 interval, sign flip mid-ray), and the BASELINE configurations at their
stated sizes: config 3 (1024^2, batch 4, bf16 decoder) and config 5 (CompCars 256^2, D = 6, batch 2, forward + backward)."""
import math

import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import configs, hip, weights
from cips_3dplusplus_amd.camera import Camera
from conftest import maxdiff
from oracle import path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(t):
    return t.to(DEV).contiguous()


def test_rays_fixture_on_hip(golden):
    """nerf_utils.py:18-121,136-170 fixtures through Render.* on the HIP path."""
    from cips_3dplusplus_amd.nerf_utils import Render
    fx = golden("rays")
    foc, ext, near, far = cu(fx["focal"]), cu(fx["extr"]), cu(fx["near"]), cu(fx["far"])
    for static in (0, 1):
        o, d, v = Render.get_rays_in_world(foc, 8, ext, static_viewdirs=bool(static))
        assert maxdiff(o.cpu(), fx[f"rays_o_{static}"]) < 1e-7
        assert maxdiff(d.cpu(), fx[f"rays_d_{static}"]) < 2e-6
        assert maxdiff(v.cpu(), fx[f"viewdirs_{static}"]) < 2e-6
    _, d, _ = Render.get_rays_in_world(foc, 8, ext)
    for N in (4, 24):
        z = Render.get_z_vals(near, far, d, N, perturb=False)
        assert z.shape == fx[f"z_{N}"].shape and maxdiff(z.cpu(), fx[f"z_{N}"]) < 2e-6
        zp = Render.get_z_vals(near, far, d, N, perturb=True, perturb_u=cu(fx[f"u_{N}"]))
        assert maxdiff(zp.cpu(), fx[f"zp_{N}"]) < 2e-6
    pts, rd, vd, zz = Render.prepare_nerf_inputs(foc, 8, ext, near, far, N_samples=6, perturb=False)
    assert pts.shape == fx["pts_6"].shape and maxdiff(pts.cpu(), fx["pts_6"]) < 2e-6
    assert maxdiff(Render.normalize_points(pts, near, far).cpu(), fx["pts_n_6"]) < 2e-5


def test_stratified_samples_fixture_on_hip(golden):
    """Render.get_z_vals(offset_sampling=False) -- the classic stratified branch `mlp_init_pass` uses (nerf_utils.py:98-117) --
    on the HIP path against the reference's fixture, with and without the per-sample jitter."""
    from cips_3dplusplus_amd.nerf_utils import Render
    fx = golden("rays_stratified")
    near, far = cu(fx["near"]), cu(fx["far"])
    d = torch.zeros(2, 8, 8, 3, device=DEV)
    for N in (1, 5, 24):
        z = Render.get_z_vals(near, far, d, N, perturb=False, offset_sampling=False)
        assert z.shape == fx[f"zs_{N}"].shape and maxdiff(z.cpu(), fx[f"zs_{N}"]) < 2e-6
        zp = Render.get_z_vals(near, far, d, N, perturb=True, offset_sampling=False, perturb_u=cu(fx[f"t_{N}"]))
        assert maxdiff(zp.cpu(), fx[f"zsp_{N}"]) < 2e-6
        zr = Render.get_z_vals(near, far, d, N, perturb=True, offset_sampling=False)          # own draw: inside the strata
        if N > 1:
            base = fx[f"zs_{N}"]
            mids = 0.5 * (base[..., 1:] + base[..., :-1])
            lo = torch.cat([base[..., :1], mids], -1)
            hi = torch.cat([mids, base[..., -1:]], -1)
            assert bool(((zr.cpu() >= lo - 1e-6) & (zr.cpu() <= hi + 1e-6)).all())


def _siren_renderer(fx):
    ren = pkg.VolumeFeatureRenderer(N_layers_renderer=2, input_dim=3, hidden_dim=32, style_dim=32, view_dim=3,
                                    with_sdf=True, output_features=True)
    sd = fx.sub("sd.")
    sd["sigmoid_beta"] = fx["vi_beta"].reshape(1)
    ren.load_state_dict(sd, strict=True)
    return ren.to(DEV).requires_grad_(False), sd


def test_siren_fixture_on_hip(golden):
    """volume_renderer.py:39-160 (FiLMSiren, SirenGenerator.points_forward with per-point view directions) and
    nerf_utils.py:230-338 (volume_integration on the reference's edge rays) through the HIP modules."""
    from cips_3dplusplus_amd.nerf_utils import Render
    fx = golden("siren")
    ren, _ = _siren_renderer(fx)
    net = ren.network
    styles = cu(fx["styles"])
    for i, (name, layer) in enumerate((("first", net.pts_linears[0]), ("hid", net.pts_linears[1]), ("view", net.views_linears))):
        y = layer(cu(fx[f"film_{name}_in"]), styles[:, i].contiguous())
        assert y.shape == fx[f"film_{name}_out"].shape
        assert maxdiff(y.cpu(), fx[f"film_{name}_out"]) < 2e-5, name
    rgb, sdf, feat = net(cu(fx["x"]), styles)
    for a, k in ((rgb, "rgb"), (sdf, "sdf"), (feat, "feat")):
        assert a.shape == fx[k].shape and maxdiff(a.cpu(), fx[k]) < 5e-5, k
    # compositing on the reference's own edge rays: [0,0] alpha -> 1 at the first sample, [0,1] empty space (all weight on
    # the last, 1e10-long interval), [1,2] sign flip mid-ray
    out = Render.volume_integration(cu(fx["rgb"]), cu(fx["vi_sdf"]), cu(fx["feat"]), cu(fx["vi_z"]), cu(fx["vi_rays_d"]),
                                    cu(fx["vi_pts"]), sigmoid_beta=cu(fx["vi_beta"]))
    for a, k in zip(out[:4], ("vi_rgb_map", "vi_feature_map", "vi_xyz", "vi_mask")):
        assert a.shape == fx[k].shape, k
        assert maxdiff(a.cpu(), fx[k]) < 2e-5 * max(1.0, float(fx[k].abs().max())), k


@pytest.mark.parametrize("tag,with_sdf,fb", [("raw", False, False), ("fb", True, True), ("raw_fb", False, True)])
def test_volume_integration_unused_branches_on_hip(golden, tag, with_sdf, fb):
    """Render.volume_integration with with_sdf=False (softplus of the raw density, incl. values above torch's threshold) and
    force_background=True (nerf_utils.py:288-310) on the HIP op, against the reference's outputs."""
    from cips_3dplusplus_amd.nerf_utils import Render
    fx = golden("vi_branches")
    out = Render.volume_integration(cu(fx["rgb"]), cu(fx["sdf"] if with_sdf else fx["raw"]), cu(fx["feat"]), cu(fx["z"]),
                                    cu(fx["rays_d"]), cu(fx["pts"]), with_sdf=with_sdf,
                                    sigmoid_beta=cu(fx["beta"]) if with_sdf else None, force_background=fb)
    for a, k in zip(out[:4], ("rgb_map", "feature_map", "xyz", "mask")):
        ref = fx[f"{tag}_{k}"]
        assert a.shape == ref.shape, k
        assert maxdiff(a.cpu(), ref) < 2e-5 * max(1.0, float(ref.abs().max())), k


@pytest.mark.parametrize("tag", ["seeded", "b20", "bm4"])
def test_renderer_with_raw_density_fixture_on_hip(golden, tag):
    """VolumeFeatureRenderer(with_sdf=False): the fused render kernel's raw-density branch (alpha = 1 - exp(-softplus(raw) *
    delta), nerf_utils.py:288-297) against the reference's own outputs, through the reference entry (explicit sample points).
    `b20` has raw values on both sides of softplus's threshold.  The differentiable path takes the branch as well (its
    gradients: test_gpu_backward.py::test_raw_density_renderer_bwd_vs_oracle)."""
    fx = golden("renderer_raw")
    ren = pkg.VolumeFeatureRenderer(N_layers_renderer=2, input_dim=3, hidden_dim=32, style_dim=32, view_dim=3, with_sdf=False,
                                    output_features=True)
    sd = fx.sub("sd.")
    sd["network.sigma_linear.bias"] = fx[f"{tag}_bias"]
    ren.load_state_dict(sd, strict=True)
    ren = ren.to(DEV).requires_grad_(False)
    out = ren(cu(fx["pts"]), cu(fx["rays_d"]), cu(fx["viewdirs"]), cu(fx["z"]), cu(fx["near"]), cu(fx["far"]),
              styles=cu(fx["styles"]))
    for a, k in zip(out[:5], ("rgb_map", "feature_map", "raw", "mask", "xyz")):
        ref = fx[f"{tag}_{k}"]
        assert a.shape == ref.shape, k
        assert maxdiff(a.cpu(), ref) < 3e-5 * max(1.0, float(ref.abs().max())), k
    # the sdf interpretation of the same numbers gives a different image: the flag is what is tested
    ren.with_sdf = True
    other = ren(cu(fx["pts"]), cu(fx["rays_d"]), cu(fx["viewdirs"]), cu(fx["z"]), cu(fx["near"]), cu(fx["far"]),
                styles=cu(fx["styles"]))
    assert maxdiff(other[0].cpu(), fx[f"{tag}_rgb_map"]) > 1e-3
    ren.with_sdf = False
    from cips_3dplusplus_amd import autograd as ag
    cam = O.camera_params(torch.tensor([[0.3, 0.1], [-0.6, -0.1]]), 8, 6, 0.12)
    film = torch.zeros(2, 3, 2, 32, device=DEV, requires_grad=True)
    f, t, _, _ = ag.NerfRenderFn.apply(ren, cu(cam[0]), cu(cam[1]), cu(cam[2]), cu(cam[3]), film, None, 8, 6, False)
    (f.sum() + t.sum()).backward()
    assert film.grad is not None and bool(torch.isfinite(film.grad).all())


@pytest.mark.parametrize("hidden,D", [(32, 2), (256, 2)])
def test_fused_render_on_edge_rays(hidden, D):
    """The compositing edge regimes, forced INSIDE the fused render kernel: with the sdf head's weights zeroed the sdf of every
    point equals the head's bias, and (bias, sigmoid_beta) pick the regime: `saturated` (sigma = 1000: alpha = 1 at the
    first sample, transmittance 1e-10 after it), `void` (sigma = 0 everywhere: no weight at all, rgb_map = -1),
    `background` (thin medium: nearly all weight on the last, 1e10-long interval: mask ~ 1), `soft` (the regime of random
    weights).  Compared with the oracle on the same weights (nerf_utils.py:264-338)."""
    cfg = configs.tiny_G_cfg(hidden, D, 1) if hidden < 256 else configs.ffhq_G_cfg(256, D)
    G = pkg.build_generator(cfg, DEV, seed=3)
    S, N, B = 12, 8, 2
    cam = O.camera_params(torch.tensor([[0.3, 0.1], [-0.6, -0.1]]), S, 6, 0.12)
    styles = weights.det_normal("edge.styles", (B, D + 1, cfg["mapping_renderer_cfg"]["style_dim"]), 0.5, 1)
    for bias, beta, what in ((-0.05, 1e-3, "saturated"), (0.05, 1e-3, "void"), (0.5, 0.1, "background"), (0.004, 0.1, "soft")):
        with torch.no_grad():
            G.renderer.network.sigma_linear.weight.zero_()
            G.renderer.network.sigma_linear.bias.fill_(bias)
            G.renderer.sigmoid_beta.fill_(beta)
        sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
        thumb, feats, sdf, mask, xyz = G.renderer.render(cu(cam[0]), cu(cam[1]), cu(cam[2]), cu(cam[3]), cu(styles), S, N,
                                                         return_sdf=True)
        ro, rd, vd = O.rays_in_world(cam[1], S, cam[0], False)
        z = O.z_vals(cam[2], cam[3], B, S, S, N)
        pts = O.ray_points(ro, rd, z)
        R = S * S
        ref = O.renderer_forward(sd, "renderer", pts.reshape(B, R, N, 3), rd.reshape(B, R, 3), vd.reshape(B, R, 3),
                                 z.reshape(B, R, N), cam[2], cam[3], styles, D)       # rgb, feat, sdf, mask, xyz
        to_img = lambda t: t.reshape(B, S, S, -1).permute(0, 3, 1, 2)               # noqa: E731
        assert maxdiff(sdf.cpu().reshape(B, R, N), ref[2].reshape(B, R, N)) < 1e-5, what
        tol = 1e-4 if hidden == 256 else 3e-5
        assert maxdiff(thumb.cpu(), to_img(ref[0])) < tol, what
        assert maxdiff(feats.cpu(), to_img(ref[1])) < tol * max(1.0, float(ref[1].abs().max())), what
        assert maxdiff(mask.cpu(), to_img(ref[3])) < tol and maxdiff(xyz.cpu(), to_img(ref[4])) < tol, what
        bg = mask[:, 0].cpu()
        if what == "saturated":
            assert float(bg.abs().max()) < 1e-6
            # all weight on sample 0: xyz is the first sample's position
            assert maxdiff(xyz.cpu(), to_img(pts.reshape(B, R, N, 3)[:, :, 0])) < 1e-5
        if what == "void":
            assert float(bg.abs().max()) < 1e-6 and float((thumb.cpu() + 1).abs().max()) < 1e-6
            assert float(feats.abs().max()) < 1e-6
        if what == "background":
            assert float(bg.min()) > 0.95


def _psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10 * math.log10(float(b.abs().max()) ** 2 / mse)


def test_config3_at_stated_size():
    """BASELINE config 3 as stated: FFHQ 1024^2 full generator, batch 4, bf16 decoder (both forms: operands only, and operands +
    bf16 storage of the up-sampling stages' activations, the form bench.py reports).  Bounds: PSNR of the bf16-decoder
    image against the exact fp32 image of the same inputs >= 55 dB (measured ~60 dB: a 20 dB regression cannot pass), the fp32 NeRF
    outputs unchanged, and
    batch independence: every view of the batch-4 call equals the batch-1 call on that view's inputs up to summation
    order (the ray-chunk count of the render kernel depends on the batch): 1e-4 of the range in fp32 mode, > 50 dB in
    bf16 mode (a last-bit change of a feature can flip a bf16 rounding)."""
    cfg = configs.ffhq_G_cfg(1024, 2)
    G = pkg.build_generator(cfg, DEV, seed=0)
    B = 4
    zs = [cu(weights.det_normal("c3.z0", (B, 256), 1.0, 3)), cu(weights.det_normal("c3.z1", (B, 256), 1.0, 3))]
    locs = cu(weights.det_normal("c3.locs", (B, 2), 1.0, 3) * torch.tensor([0.3, 0.15]))
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=locs)
    _, nb, _ = weights.synth_inputs(cfg, seed=3)
    nb = [cu(b) for b in nb]
    ncfg = dict(N_samples=24, perturb=False, static_viewdirs=False)
    kw = dict(cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=nb, nerf_cfg=ncfg, return_xyz=True)
    r32 = G(zs=zs, **kw)
    rgb32, thumb32 = r32["rgb"].clone(), r32["thumb_rgb"].clone()
    G.set_decoder_precision("bf16")
    r16 = G(zs=zs, **kw)
    assert r16["rgb"].shape == (B, 3, 1024, 1024) and bool(torch.isfinite(r16["rgb"]).all())
    psnr = _psnr(r16["rgb"], rgb32)
    print(f"config 3 (1024^2, B=4, bf16 decoder) vs fp32: PSNR {psnr:.1f} dB, max-abs {maxdiff(r16['rgb'], rgb32):.3e} "
          f"on range {float(rgb32.abs().max()):.2f}")
    assert psnr >= 55.0 and not torch.equal(r16["rgb"], rgb32)
    assert torch.equal(r16["thumb_rgb"], thumb32)                        # the renderer stays fp32
    G.set_decoder_precision("bf16_storage")
    r16s = G(zs=zs, **kw)
    psnr_s = _psnr(r16s["rgb"], rgb32)
    print(f"config 3 with bf16 storage of the up-sampling stages: PSNR {psnr_s:.1f} dB, max-abs {maxdiff(r16s['rgb'], rgb32):.3e}")
    assert psnr_s >= 55.0 and bool(torch.isfinite(r16s["rgb"]).all()) and torch.equal(r16s["thumb_rgb"], thumb32)
    assert not torch.equal(r16s["rgb"], r16["rgb"])
    rng = float(rgb32.abs().max())
    for prec, full in (("bf16", r16["rgb"]), ("bf16_storage", r16s["rgb"]), ("fp32", rgb32)):
        G.set_decoder_precision(prec)
        for b in (0, 3):
            one = G(zs=[z[b:b + 1].contiguous() for z in zs], cam_poses=e[b:b + 1].contiguous(), focals=f[b:b + 1].contiguous(),
                    img_size=64, near=n[b:b + 1].contiguous(), far=fa[b:b + 1].contiguous(), noise_bufs=nb, nerf_cfg=ncfg)
            d = maxdiff(one["rgb"][0], full[b])
            print(f"batch independence, {prec}, view {b}: max-abs {d:.3e} on range {rng:.2f}")
            if prec == "fp32":
                assert d < 1e-4 * rng, f"view {b} of the batch differs from its batch-1 render"
            else:
                assert _psnr(one["rgb"][0], full[b]) > 50.0


def test_config5_at_stated_size_backward(golden):
    """BASELINE config 5 as stated: CompCars camera, 256^2 output, D = 6, 64x64 rays x 24 samples, static view directions,
    batch 2 (image + mirrored view): loss, outputs and every gradient of one flip-inversion step against the REFERENCE's
    (tests/golden/config5.npz, strided).  Bounds: outputs 1e-3 max-abs (north_star); each gradient within
    max(2e-3 of its own max-abs, 20 x the reference's own fp32-vs-fp64 difference for that gradient, stored in the fixture):
    the noise-buffer gradients of the late layers are heavily cancelled sums (max-abs 2e-5) on which the reference itself
    carries ~1 % of fp32 noise."""
    fx = golden("config5")
    res, D, N = 256, 6, 24
    cfg = configs.ffhq_G_cfg(res, D)
    G = pkg.build_generator(cfg, DEV, seed=2)
    sd_cpu = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    assert weights.state_dict_checksum(sd_cpu) == int(fx["sd_checksum"])
    G.decoder.requires_grad_(True)
    locs, w_r, w_d, nb, t_rgb, t_thumb = weights.synth_inversion_inputs(cfg, res)
    leaf = lambda t: cu(t).requires_grad_(True)       # noqa: E731
    locs, w_r, w_d, nb = leaf(locs), leaf(w_r), leaf(w_d), [leaf(b) for b in nb]
    cam_cfg = configs.COMPCARS_CAM_CFG
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=locs, fov_ang=cam_cfg["fov_ang"],
                                                   dist_radius=cam_cfg["dist_radius"])
    r = G(zs=[None, None], style_render=w_r, style_decoder=w_d, cam_poses=e, focals=f, img_size=64, near=n, far=fa,
          noise_bufs=nb, nerf_cfg=dict(N_samples=N, perturb=False, static_viewdirs=True), renderer_detach=False,
          return_xyz=True)
    st = int(fx["stride"])
    assert maxdiff(r["rgb"].detach().flatten()[::st].cpu(), fx["rgb_s"]) < 1e-3
    for k, fk in (("thumb_rgb", "thumb"), ("xyz", "xyz"), ("mask", "mask"), ("depth", "depth")):
        assert maxdiff(r[k].detach().cpu(), fx[fk]) < 1e-4, k
    loss = ((r["rgb"] - cu(t_rgb)) ** 2).mean() + 50 * ((r["thumb_rgb"] - cu(t_thumb)) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(fx["loss"])) < 1e-4 * float(fx["loss"])

    def close(a, key, strided):
        ref = fx[f"g.{key}_s" if strided else f"g.{key}"]
        a = a.detach().flatten()[::st] if strided else a.detach()
        scale, floor = float(fx[f"g.{key}_absmax"]), float(fx[f"g.{key}_floor"])
        err = float((a.cpu().reshape(ref.shape) - ref).abs().max())
        assert err <= max(2e-3 * scale, 20 * floor) + 1e-12, f"d{key}: err {err:.3e}, max-abs {scale:.3e}, reference fp32 floor {floor:.3e}"
        return err / (scale + 1e-30)

    worst = max(close(locs.grad, "locs", False), close(w_r.grad, "w_r", False), close(w_d.grad, "w_d", False))
    for i, b in enumerate(nb):
        worst = max(worst, close(b.grad, f"noise{i}", True))
    n_checked = 0
    for name, p in G.decoder.named_parameters():
        if f"g.dec.{name}_s" in fx:
            worst = max(worst, close(p.grad, f"dec.{name}", True))
            n_checked += 1
        else:
            assert p.grad is None, name
    assert n_checked > 60
    print(f"config 5 at stated size: loss {float(loss):.4f}, worst gradient error {worst:.2e} of its max-abs "
          f"({n_checked} decoder parameters)")


def _headline_inputs(B, seed=21):
    from cips_3dplusplus_amd.camera import Camera
    cfg = configs.ffhq_G_cfg(1024, 2)
    G = pkg.build_generator(cfg, DEV, seed=0)
    zs, _, _ = weights.synth_inputs(cfg, batch=B, seed=seed)
    zs = [cu(z) for z in zs]
    locs = cu(torch.tensor([[0.25, 0.1], [-0.3, -0.05], [0.1, 0.12], [0.0, 0.0]])[:B])
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=locs, fov_ang=6, dist_radius=0.12)
    nb = G.create_noise_bufs(64, DEV)
    u = cu(weights.det_unit_uniform("prop.u", (B, 64, 64, 1), 3))
    return G, zs, (e, f, n, fa), nb, u


def test_headline_forward_is_deterministic_and_batch_independent():
    """Size-independent properties at the bench's own size (FFHQ 1024^2, D=2, 64x64 rays x 24 samples): the same inputs give
    the same bits on every call (no atomics, no run-dependent reduction order anywhere on the forward path), and a view's
    image does not depend on what else is in the batch (every kernel of the path works per sample; only the workgroup ->
    tile order of the planes GEMM differs between batch sizes, so this is exact up to its fp32 accumulation order)."""
    G, zs, (e, f, n, fa), nb, u = _headline_inputs(2)
    kw = dict(img_size=64, noise_bufs=nb, nerf_cfg=dict(N_samples=24, perturb=True, static_viewdirs=False))
    run = lambda sl: G(zs=[z[sl] for z in zs], cam_poses=e[sl], focals=f[sl], near=n[sl], far=fa[sl], perturb_u=u[sl], **kw)
    both = run(slice(0, 2))
    again = run(slice(0, 2))
    for k in ("rgb", "thumb_rgb", "mask", "depth"):
        assert torch.equal(both[k], again[k]), k
    for i in range(2):
        one = run(slice(i, i + 1))
        assert maxdiff(one["rgb"].cpu(), both["rgb"][i:i + 1].cpu()) < 2e-5 * float(both["rgb"].abs().max()), i
        assert maxdiff(one["thumb_rgb"].cpu(), both["thumb_rgb"][i:i + 1].cpu()) < 1e-6, i


def test_config1_at_stated_shape():
    """BASELINE config 1 on the HIP path at its stated shape: FFHQ generator with a 64x64 output (upsample_list = [], the
    `test__rendering_time` CPU case), D = 8 renderer layers, 64x64 rays x 24 samples, one view, against the oracle on the same
    inputs -- with fixed and with jittered sample depths.  This is the configuration whose planes run ends in a different
    place: nine ToRGBs at 64^2, one more than a fold holds (test_planes_run_ends_in_fp32_when_nothing_up_samples walks the
    plan; here the numbers are held to the path's parity bar, 1e-3 max-abs, and to what is measured)."""
    cfg = configs.ffhq_G_cfg(64, 8)
    G = pkg.build_generator(cfg, DEV, seed=1)
    zs, nb, _ = weights.synth_inputs(cfg, seed=9)
    loc = torch.tensor([[0.25, -0.1]])
    e, f, n, fa, _ = Camera.generate_camera_params(64, DEV, locations=cu(loc))
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    u = weights.det_unit_uniform("c1.u", (1, 64, 64, 1), 5)
    for perturb in (False, True):
        ncfg = dict(N_samples=24, perturb=perturb, static_viewdirs=False)
        r = G(zs=[cu(z) for z in zs], cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=[cu(b) for b in nb], nerf_cfg=ncfg,
              perturb_u=cu(u) if perturb else None, return_xyz=True)
        assert r["rgb"].shape == (1, 3, 64, 64) and r["thumb_rgb"].shape == (1, 3, 64, 64)
        plan = list(G._plans.values())[0].plan
        assert plan.nerf.depth == 8 and plan.nerf.n_samples == 24 and plan.nerf.img_size == 64
        ref = O.generator_forward(sd, cfg, zs, e.cpu(), f.cpu(), 64, n.cpu(), fa.cpu(), ncfg, nb, perturb_u=u if perturb else None,
                                  return_xyz=True)
        for key, bar in (("rgb", 3e-4), ("thumb_rgb", 5e-5), ("mask", 5e-5), ("xyz", 5e-5)):
            if key not in ref or key not in r:
                continue
            d = maxdiff(r[key].cpu(), ref[key])
            print(f"config 1 (64^2 output, D=8, N=24, perturb={perturb}) {key}: max-abs {d:.2e} on range {float(ref[key].abs().max()):.2f}")
            assert d < 1e-3 and d < bar * max(1.0, float(ref[key].abs().max())), key
