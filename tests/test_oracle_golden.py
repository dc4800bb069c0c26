"""Pins the CPU oracle (oracle/path.py) to golden vectors produced by the reference
(tests/golden/make_golden.py).  CPU only; runs everywhere."""
import math

import numpy as np
import pytest
import torch

from cips_3dplusplus_amd import configs, weights
from conftest import maxdiff
from oracle import path as O

TOL = 2e-6


def test_camera(golden):
    fx = golden("camera")
    for tag, cam in (("ffhq", configs.FFHQ_CAM_CFG), ("cars", configs.COMPCARS_CAM_CFG)):
        e, f, n, fa, vp = O.camera_params(fx["locs"], 64, cam["fov_ang"], cam["dist_radius"])
        assert maxdiff(e, fx[f"{tag}_extr"]) < TOL
        assert maxdiff(f, fx[f"{tag}_focal"]) < 1e-3 * 1e-2   # focal ~ 304: relative 3e-8
        assert maxdiff(n, fx[f"{tag}_near"]) == 0 and maxdiff(fa, fx[f"{tag}_far"]) == 0
        assert maxdiff(vp, fx[f"{tag}_vp"]) == 0
    e, f, *_ = O.camera_params(fx["locs"], 64, fx["fovt"], 0.12)
    assert maxdiff(e, fx["fovt_extr"]) < TOL and maxdiff(f, fx["fovt_focal"]) < 1e-4
    e, f, *_ = O.camera_params(torch.zeros(8, 2), 64, 6, 0.12, up=fx["roll_ups"])
    assert maxdiff(e, fx["roll_extr"]) < TOL
    e, *_ = O.camera_params(fx["deg_locs"], 64, 6, 0.12)
    assert maxdiff(e, fx["deg_extr"]) < TOL


def test_frontal_camera_constants():
    # locations = 0 -> focal 304.4597, near .88, far 1.12 (FFHQ cam_cfg)
    e, f, n, fa, _ = O.camera_params(torch.zeros(1, 2), 64, 6, 0.12)
    assert abs(float(f) - 304.4597) < 1e-3 and abs(float(n) - 0.88) < 1e-7 and abs(float(fa) - 1.12) < 1e-7
    assert maxdiff(e[0], torch.tensor([[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1]])) < 1e-7


def test_stratified_samples(golden):
    """the classic stratified branch of get_z_vals (offset_sampling=False) against the reference's fixture"""
    fx = golden("rays_stratified")
    for N in (1, 5, 24):
        assert maxdiff(O.z_vals_stratified(fx["near"], fx["far"], 2, 8, 8, N), fx[f"zs_{N}"]) < TOL
        assert maxdiff(O.z_vals_stratified(fx["near"], fx["far"], 2, 8, 8, N, fx[f"t_{N}"]), fx[f"zsp_{N}"]) < TOL


def test_rays_and_samples(golden):
    fx = golden("rays")
    for static in (0, 1):
        o, d, v = O.rays_in_world(fx["focal"], 8, fx["extr"], bool(static))
        assert maxdiff(o, fx[f"rays_o_{static}"]) == 0
        assert maxdiff(d, fx[f"rays_d_{static}"]) < TOL
        assert maxdiff(v, fx[f"viewdirs_{static}"]) < TOL
    for N in (4, 24):
        assert maxdiff(O.z_vals(fx["near"], fx["far"], 2, 8, 8, N), fx[f"z_{N}"]) < TOL
        assert maxdiff(O.z_vals(fx["near"], fx["far"], 2, 8, 8, N, fx[f"u_{N}"]), fx[f"zp_{N}"]) < TOL
    o, d, _ = O.rays_in_world(fx["focal"], 8, fx["extr"])
    pts = O.ray_points(o, d, O.z_vals(fx["near"], fx["far"], 2, 8, 8, 6))
    assert maxdiff(pts, fx["pts_6"]) < TOL
    assert maxdiff(O.normalize_points(pts, fx["near"], fx["far"]), fx["pts_n_6"]) < 2e-5


def test_film_siren_and_compositing(golden):
    fx = golden("siren")
    sd = fx.sub("sd.")
    styles = fx["styles"]
    for i, (name, prefix) in enumerate((("first", "network.pts_linears.0"), ("hid", "network.pts_linears.1"),
                                        ("view", "network.views_linears"))):
        y = O.film_siren(sd, prefix, fx[f"film_{name}_in"], styles[:, i])
        assert maxdiff(y, fx[f"film_{name}_out"]) < 5e-6
    x = fx["x"]
    # the reference net takes per-point view dirs; the oracle takes per-ray dirs, so evaluate
    # sample-by-sample (one direction per ray per call)
    outs = [O.siren_points(sd, "network", x[:, :, n:n + 1, :3], x[:, :, n, 3:], styles, 2) for n in range(x.shape[2])]
    rgb = torch.cat([o[0] for o in outs], dim=2)
    sdf = torch.cat([o[1] for o in outs], dim=2)
    feat = torch.cat([o[2] for o in outs], dim=2)
    assert maxdiff(rgb, fx["rgb"]) < 5e-6 and maxdiff(sdf, fx["sdf"]) < 5e-6 and maxdiff(feat, fx["feat"]) < 5e-6
    rm, fm, xyz, mask = O.volume_integration(fx["rgb"], fx["vi_sdf"], fx["feat"], fx["vi_z"], fx["vi_rays_d"],
                                             fx["vi_pts"], fx["vi_beta"])
    assert maxdiff(rm, fx["vi_rgb_map"]) < TOL and maxdiff(fm, fx["vi_feature_map"]) < TOL
    assert maxdiff(xyz, fx["vi_xyz"]) < TOL and maxdiff(mask, fx["vi_mask"]) < TOL


@pytest.mark.parametrize("tag,with_sdf,fb", [("raw", False, False), ("fb", True, True), ("raw_fb", False, True)])
def test_volume_integration_unused_branches(golden, tag, with_sdf, fb):
    """Raw-density (softplus) and force_background branches of Render.volume_integration (nerf_utils.py:288-310) against the
    reference's own outputs (tests/golden/vi_branches.npz)."""
    fx = golden("vi_branches")
    rm, fm, xyz, mask = O.volume_integration(fx["rgb"], fx["sdf"] if with_sdf else fx["raw"], fx["feat"], fx["z"], fx["rays_d"],
                                             fx["pts"], fx["beta"], with_sdf=with_sdf, force_background=fb)
    for a, k in ((rm, "rgb_map"), (fm, "feature_map"), (xyz, "xyz"), (mask, "mask")):
        assert maxdiff(a, fx[f"{tag}_{k}"]) < TOL, k


@pytest.mark.parametrize("tag", ["seeded", "b20", "bm4"])
def test_renderer_with_raw_density(golden, tag):
    """VolumeFeatureRenderer(with_sdf=False).forward (volume_renderer.py:192-303 + nerf_utils.py:288-297) on explicit sample
    points, against the reference's outputs (tests/golden/renderer_raw.npz); `b20` straddles softplus's threshold."""
    fx = golden("renderer_raw")
    sd = {"renderer." + k: v for k, v in fx.sub("sd.").items()}
    sd["renderer.network.sigma_linear.bias"] = fx[f"{tag}_bias"]
    rm, fm, raw, mask, xyz = O.renderer_forward(sd, "renderer", fx["pts"], fx["rays_d"], fx["viewdirs"], fx["z"], fx["near"],
                                                fx["far"], fx["styles"], 2, with_sdf=False)
    assert maxdiff(raw, fx[f"{tag}_raw"]) < 2e-5
    for a, k in ((rm, "rgb_map"), (fm, "feature_map"), (xyz, "xyz"), (mask, "mask")):
        assert a.shape == fx[f"{tag}_{k}"].shape and maxdiff(a, fx[f"{tag}_{k}"]) < 2e-5, k


def test_ops(golden):
    fx = golden("ops")
    for name in fx["ufd_names"]:
        up, down, p0, p1 = [int(v) for v in fx[f"ufd_{name}_cfg"]]
        y = O.upfirdn2d(fx[f"ufd_{name}_x"], fx[f"ufd_{name}_k"], up, down, (p0, p1))
        assert y.shape == fx[f"ufd_{name}_y"].shape, name
        assert maxdiff(y, fx[f"ufd_{name}_y"]) < TOL, name
    for name in ("2d_g1", "2d_gs", "4d", "4d_nob", "3d"):
        b = fx[f"flr_{name}_b"] if f"flr_{name}_b" in fx else None
        y = O.fused_leaky_relu(fx[f"flr_{name}_x"], b, scale=float(fx[f"flr_{name}_scale"]))
        assert maxdiff(y, fx[f"flr_{name}_y"]) < TOL, name


def test_modulated_conv_blocks(golden):
    fx = golden("modconv")
    for tag in fx["mc_names"]:
        sd = {"m." + k: v for k, v in fx.sub(f"mc_{tag}.sd.").items()}
        y = O.modulated_conv2d(sd, "m", fx[f"mc_{tag}.x"], fx[f"mc_{tag}.style"],
                               demodulate="_d1" in str(tag), upsample="_up1" in str(tag))
        assert maxdiff(y, fx[f"mc_{tag}.y"]) < 2e-5, tag
    for tag in ("up0", "up1"):
        sd = {"m." + k: v for k, v in fx.sub(f"sc_{tag}.sd.").items()}
        y = O.styled_conv(sd, "m", fx[f"sc_{tag}.x"], fx[f"sc_{tag}.style"], fx[f"sc_{tag}.noise"], upsample=tag == "up1")
        assert maxdiff(y, fx[f"sc_{tag}.y"]) < 2e-5
        sd = {"m." + k: v for k, v in fx.sub(f"rgb_{tag}.sd.").items()}
        y = O.to_rgb(sd, "m", fx[f"rgb_{tag}.x"], fx[f"rgb_{tag}.style"], fx[f"rgb_{tag}.skip"], upsample=tag == "up1")
        assert maxdiff(y, fx[f"rgb_{tag}.y"]) < 2e-5
        y = O.to_rgb(sd, "m", fx[f"rgb_{tag}.x"], fx[f"rgb_{tag}.style"])
        assert maxdiff(y, fx[f"rgb_{tag}.y_noskip"]) < 2e-5


def test_k1_upsample_identity(golden):
    """k=1 up-sampling conv == per-sample GEMM then upfirdn2d(up=2, pad=(2,1), kernel*4).
    The HIP decoder is built on this identity, so pin it against the reference's output."""
    fx = golden("modconv")
    tag = "k1_up1_d1"
    sd = {"m." + k: v for k, v in fx.sub(f"mc_{tag}.sd.").items()}
    lo = O.modulated_conv2d(sd, "m", fx[f"mc_{tag}.x"], fx[f"mc_{tag}.style"], demodulate=True, upsample=False)
    y = O.upfirdn2d(lo, sd["m.blur.kernel"], up=2, pad=(2, 1))
    assert maxdiff(y, fx[f"mc_{tag}.y"]) < 2e-5


def _tiny_cfg(tag):
    hidden = 32
    D = 3 if "d3" in tag else 2
    return configs.tiny_G_cfg(hidden=hidden, N_layers_renderer=D, kernel_size=3 if "k3" in tag else 1)


@pytest.mark.parametrize("tag", ["h32_d2", "h32_d3", "h32_d2_k3"])
def test_tiny_generator_end_to_end(golden, tag):
    fx = golden("tiny_generator")
    cfg = _tiny_cfg(tag)
    sd = fx.sub(f"{tag}.sd.")
    assert list(fx[f"{tag}.keys"]) == list(sd.keys())
    zs = [fx[f"{tag}.z0"], fx[f"{tag}.z1"]]
    cam = O.camera_params(fx[f"{tag}.locs"], 8, 6, 0.12)
    nb = [fx[f"{tag}.noise{i}"] for i in range(O.decoder_layout(cfg)["num_layers"])]
    means = (fx[f"{tag}.mean_r"], fx[f"{tag}.mean_d"])
    runs = (("a", dict(N_samples=6, perturb=False, static_viewdirs=False), 1.0, None),
            ("b", dict(N_samples=5, perturb=False, static_viewdirs=True), 0.5, None),
            ("c", dict(N_samples=6, perturb=True, static_viewdirs=False), 1.0, fx[f"{tag}.c.u"]))
    for vtag, ncfg, trunc, u in runs:
        r = O.generator_forward(sd, cfg, zs, cam[0], cam[1], 8, cam[2], cam[3], ncfg, nb, truncation=trunc,
                                style_render_mean=means[0], style_decoder_mean=means[1], perturb_u=u,
                                return_sdf=True, return_xyz=True)
        for k in ("rgb", "thumb_rgb", "sdf", "xyz", "mask", "depth"):
            assert r[k].shape == fx[f"{tag}.{vtag}.{k}"].shape
            assert maxdiff(r[k], fx[f"{tag}.{vtag}.{k}"]) < 5e-5, (tag, vtag, k)
    mr, md = O.mean_latents(sd, cfg, fx[f"{tag}.ml_zr"], fx[f"{tag}.ml_zd"])
    assert maxdiff(mr, fx[f"{tag}.ml_r"]) < 1e-5 and maxdiff(md, fx[f"{tag}.ml_d"]) < 1e-5


def test_noise_buf_shapes():
    cfg = configs.ffhq_G_cfg(1024, 2)
    sizes = [b.shape[-1] for b in O.create_noise_bufs(cfg, 64)]
    assert sizes == [64] * 9 + [128] * 2 + [256] * 2 + [512] * 2 + [1024] * 2
    lay = O.decoder_layout(cfg)
    assert lay["n_latent"] == 18 and lay["num_layers"] == 17


@pytest.mark.parametrize("tag,D,static", [("h32_d2", 2, True), ("h32_d3", 3, False)])
def test_oracle_autograd_matches_reference_gradients(golden, tag, D, static):
    """The oracle under torch autograd reproduces loss and gradients of the imported reference for one inversion-like
    step (tests/golden/make_golden.py:g_backward) -- this is what the GPU backward tests lean on."""
    fx = golden("backward")
    tiny = golden("tiny_generator")
    cfg = configs.tiny_G_cfg(32, D, 1)
    shapes = {k[len(f"{tag}.sd."):]: tuple(tiny[k].shape) for k in tiny.keys() if k.startswith(f"{tag}.sd.")}
    sd = weights.synth_state_dict(shapes, seed=7)
    sd = {k: (v.clone().requires_grad_(True) if k.startswith("decoder.") and v.is_floating_point() and "kernel" not in k
              else v) for k, v in sd.items()}
    locs = fx[f"{tag}.locs"].clone().requires_grad_(True)
    w_r = fx[f"{tag}.w_r"].clone().requires_grad_(True)
    w_d = fx[f"{tag}.w_d"].clone().requires_grad_(True)
    n_noise = len([k for k in fx.keys() if k.startswith(f"{tag}.noise")])
    nb = [fx[f"{tag}.noise{i}"].clone().requires_grad_(True) for i in range(n_noise)]
    cam = O.camera_params(locs, 8, 6, 0.12)
    r = O.generator_forward(sd, cfg, [None, None], cam[0], cam[1], 8, cam[2], cam[3],
                            dict(N_samples=6, perturb=False, static_viewdirs=static), nb, style_render=w_r, style_decoder=w_d)
    loss = ((r["rgb"] - fx[f"{tag}.t_rgb"]) ** 2).mean() + 50 * ((r["thumb_rgb"] - fx[f"{tag}.t_thumb"]) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(fx[f"{tag}.loss"])) < 1e-4 * float(fx[f"{tag}.loss"])

    def close(a, b, what):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7, what

    close(locs.grad, fx[f"{tag}.g.locs"], "locs"); close(w_r.grad, fx[f"{tag}.g.w_r"], "w_r")
    close(w_d.grad, fx[f"{tag}.g.w_d"], "w_d")
    for i in range(n_noise):
        close(nb[i].grad, fx[f"{tag}.g.noise{i}"], f"noise{i}")
    n = 0
    for k in fx.keys():
        if k.startswith(f"{tag}.g.dec."):
            close(sd["decoder." + k[len(f"{tag}.g.dec."):]].grad, fx[k], k)
            n += 1
    assert n > 30


def test_oracle_config5_at_stated_size(golden):
    """BASELINE config 5 at its stated size (CompCars camera, 256^2, D = 6, batch 2, forward + backward of the surrogate
    loss): the oracle under autograd against the imported reference (tests/golden/config5.npz) -- outputs, loss and the
    gradients of the leaves the inversion loop optimises.  This pins what tests/test_gpu_reference_fixtures.py leans on."""
    fx = golden("config5")
    res, D, N = 256, 6, 24
    cfg = configs.ffhq_G_cfg(res, D)
    import cips_3dplusplus_amd as pkg
    shapes = {k: tuple(v.shape) for k, v in pkg.Generator(**cfg).state_dict().items()}
    sd = weights.synth_state_dict(shapes, seed=2)
    assert weights.state_dict_checksum(sd) == fx["sd_checksum"]
    locs, w_r, w_d, nb, t_rgb, t_thumb = weights.synth_inversion_inputs(cfg, res)
    locs.requires_grad_(True); w_r.requires_grad_(True); w_d.requires_grad_(True)
    cam = O.camera_params(locs, 64, configs.COMPCARS_CAM_CFG["fov_ang"], configs.COMPCARS_CAM_CFG["dist_radius"])
    r = O.generator_forward(sd, cfg, [None, None], cam[0], cam[1], 64, cam[2], cam[3],
                            dict(N_samples=N, perturb=False, static_viewdirs=True), nb, style_render=w_r, style_decoder=w_d,
                            return_xyz=True)
    st = int(fx["stride"])
    assert maxdiff(r["rgb"].detach().flatten()[::st], fx["rgb_s"]) < 2e-4
    assert maxdiff(r["thumb_rgb"].detach(), fx["thumb"]) < 2e-6 and maxdiff(r["xyz"].detach(), fx["xyz"]) < 2e-6
    loss = ((r["rgb"] - t_rgb) ** 2).mean() + 50 * ((r["thumb_rgb"] - t_thumb) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(fx["loss"])) < 1e-5 * float(fx["loss"])
    for a, k in ((locs.grad, "g.locs"), (w_r.grad, "g.w_r"), (w_d.grad, "g.w_d")):
        assert float((a - fx[k]).abs().max()) <= 2e-4 * float(fx[k].abs().max()) + 1e-8, k


def test_k3_upsample_identity(golden):
    """The identity csrc/conv3x3.hip builds its up-sampling branch on: conv_transpose2d(x, w, stride 2) followed by
    Blur(4x4 taps x 4, pad (1, 1)) (models/model_v3.py:280-291) == a valid 3x3 correlation with the 180-degree rotated
    taps of  Z = upfirdn2d(x, taps, up=2, pad=(3, 2))  applied per INPUT channel.  Checked on the reference's own
    k = 3 up-sampling fixtures (tests/golden/modconv.npz: mc_k3_up1_d1 / _d0)."""
    import torch.nn.functional as F
    fx = golden("modconv")
    for tag in ("k3_up1_d1", "k3_up1_d0"):
        sd = fx.sub(f"mc_{tag}.sd.")
        x, style, y_ref = fx[f"mc_{tag}.x"], fx[f"mc_{tag}.style"], fx[f"mc_{tag}.y"]
        B, Cin, H, W = x.shape
        weight = sd["weight"]
        Cout = weight.shape[1]
        s = O._equal_linear(sd, "modulation", style).view(B, 1, Cin, 1, 1)
        w = (1 / math.sqrt(Cin * 9)) * weight * s
        if tag.endswith("d1"):
            w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).view(B, Cout, 1, 1, 1)
        z = O.upfirdn2d(x, sd["blur.kernel"], up=2, pad=(3, 2))
        assert z.shape[-2:] == (2 * H + 2, 2 * W + 2)
        y = F.conv2d(z.reshape(1, B * Cin, 2 * H + 2, 2 * W + 2), torch.flip(w, [3, 4]).reshape(B * Cout, Cin, 3, 3), groups=B)
        y = y.view(B, Cout, 2 * H, 2 * W)
        assert y.shape == y_ref.shape and maxdiff(y, y_ref) < 2e-5 * max(1.0, float(y_ref.abs().max()))
