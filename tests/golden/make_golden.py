"""Generate golden vectors by running the REFERENCE on CPU (this container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--full]

Writes tests/golden/*.npz: inputs + expected outputs (data only, no reference source).
`--full` also produces the full-size (256^2 / 1024^2) strided-sample fixtures (minutes of CPU).
Every fixture records which reference entry point produced it (file:line in `_src`).
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(os.path.dirname(HERE))]
from _ref_import import import_reference  # noqa: E402

ref = import_reference()
from exp.cips3d import nerf_utils as ref_nerf  # noqa: E402
from exp.cips3d import volume_renderer as ref_vr  # noqa: E402
import op as ref_op  # noqa: E402

from cips_3dplusplus_amd import configs, weights  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def npy(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **d):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **npy(d))
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


# ---------------------------------------------------------------- 1. cameras
def g_camera():
    out = {"_src": "cips3d/nerf_utils.py:344-436,466-564"}
    locs = torch.tensor([[0.0, 0.0], [0.4, -0.1], [-0.77, 0.0], [3.0, 0.1]])
    for tag, cam in (("ffhq", configs.FFHQ_CAM_CFG), ("cars", configs.COMPCARS_CAM_CFG)):
        e, f, n, fa, vp = ref_nerf.Camera.generate_camera_params(
            img_size=64, device="cpu", locations=locs, fov_ang=cam["fov_ang"], dist_radius=cam["dist_radius"])
        out.update({f"{tag}_extr": e, f"{tag}_focal": f, f"{tag}_near": n, f"{tag}_far": fa, f"{tag}_vp": vp})
    out["locs"] = locs
    fov_t = torch.tensor([[6.0], [6.5], [7.0], [8.0]])
    e, f, n, fa, _ = ref_nerf.Camera.generate_camera_params(
        img_size=64, device="cpu", locations=locs, fov_ang=fov_t, dist_radius=0.12)
    out.update(fovt=fov_t, fovt_extr=e, fovt_focal=f)
    # in-plane roll trajectory of the translate_rotate demo (render_video_web_v10.py:1625-1641)
    t = torch.linspace(0, 1, 8)
    alpha = t * 2 * torch.pi + 0.5 * torch.pi
    ups = torch.stack([torch.cos(alpha), torch.sin(alpha), torch.zeros(8)], dim=1)
    e, f, n, fa, _ = ref_nerf.Camera.generate_camera_params_v1(
        img_size=64, device="cpu", locations=torch.zeros(8, 2), up=ups, fov_ang=6, dist_radius=0.12)
    out.update(roll_ups=ups, roll_extr=e, roll_focal=f)
    # straight-up camera: exercises the degenerate-x replacement branch
    locs_deg = torch.tensor([[0.0, 1.5707963], [0.3, 0.0]])
    e, f, n, fa, _ = ref_nerf.Camera.generate_camera_params(
        img_size=64, device="cpu", locations=locs_deg, fov_ang=6, dist_radius=0.12)
    out.update(deg_locs=locs_deg, deg_extr=e)
    save("camera", **out)


# ---------------------------------------------------------------- 2. rays / z
def g_rays():
    out = {"_src": "cips3d/nerf_utils.py:18-121,136-170"}
    locs = torch.tensor([[0.25, -0.1], [-0.5, 0.12]])
    e, f, n, fa, _ = ref_nerf.Camera.generate_camera_params(
        img_size=8, device="cpu", locations=locs, fov_ang=6, dist_radius=0.12)
    out.update(extr=e, focal=f, near=n, far=fa)
    for static in (False, True):
        o, d, v = ref_nerf.Render.get_rays_in_world(f, 8, e, static_viewdirs=static)
        out.update({f"rays_o_{int(static)}": o.contiguous(), f"rays_d_{int(static)}": d, f"viewdirs_{int(static)}": v})
    _, d, _ = ref_nerf.Render.get_rays_in_world(f, 8, e)
    for N in (4, 24):
        out[f"z_{N}"] = ref_nerf.Render.get_z_vals(n, fa, d, N, perturb=False)
        torch.manual_seed(100 + N)
        out[f"zp_{N}"] = ref_nerf.Render.get_z_vals(n, fa, d, N, perturb=True)
        torch.manual_seed(100 + N)
        out[f"u_{N}"] = torch.rand(2, 8, 8, 1)
    pts, rd, vd, zz = ref_nerf.Render.prepare_nerf_inputs(f, 8, e, n, fa, N_samples=6, perturb=False)
    out.update(pts_6=pts, pts_n_6=ref_nerf.Render.normalize_points(pts, n, fa))
    save("rays", **out)


def g_rays_stratified():
    """the classic stratified branch of Render.get_z_vals (offset_sampling=False: `mlp_init_pass`)"""
    out = {"_src": "cips3d/nerf_utils.py:69-121 (offset_sampling=False branch :98-117)"}
    locs = torch.tensor([[0.25, -0.1], [-0.5, 0.12]])
    e, f, n, fa, _ = ref_nerf.Camera.generate_camera_params(
        img_size=8, device="cpu", locations=locs, fov_ang=6, dist_radius=0.12)
    out.update(near=n, far=fa)
    _, d, _ = ref_nerf.Render.get_rays_in_world(f, 8, e)
    for N in (1, 5, 24):
        out[f"zs_{N}"] = ref_nerf.Render.get_z_vals(n, fa, d, N, perturb=False, offset_sampling=False)
        torch.manual_seed(200 + N)
        out[f"zsp_{N}"] = ref_nerf.Render.get_z_vals(n, fa, d, N, perturb=True, offset_sampling=False)
        torch.manual_seed(200 + N)
        out[f"t_{N}"] = torch.rand(2, 8, 8, N)
    save("rays_stratified", **out)


# ---------------------------------------------------------------- 3. FiLM-SIREN + compositing
def g_siren():
    out = {"_src": "cips3d/volume_renderer.py:15-160; cips3d/nerf_utils.py:230-338"}
    torch.manual_seed(3)
    W = 32
    net = ref_vr.SirenGenerator(D=2, W=W, style_dim=W, input_ch=3, input_ch_views=3)
    sd = {"network." + k: v for k, v in net.state_dict().items()}
    out.update({"sd." + k: v for k, v in sd.items()})
    B, R, N = 2, 16, 8
    x = torch.rand(B, R, N, 6) * 2 - 1
    styles = torch.randn(B, 3, W)
    rgb, sdf, feat = net(x, styles)
    out.update(x=x, styles=styles, rgb=rgb, sdf=sdf, feat=feat)
    for i, (cin, name) in enumerate(((3, "first"), (W, "hid"), (W + 3, "view"))):
        layer = [net.pts_linears[0], net.pts_linears[1], net.views_linears][i]
        xin = torch.rand(B, R, N, cin) * 2 - 1
        out[f"film_{name}_in"] = xin
        out[f"film_{name}_out"] = layer(xin, styles[:, i])
    # compositing, including a ray that saturates early and one that is background only
    z = torch.sort(torch.rand(B, R, N) * 0.24 + 0.88, dim=-1)[0]
    rays_d = torch.randn(B, R, 3)
    pts = torch.randn(B, R, N, 3)
    sdf2 = torch.randn(B, R, N, 1) * 0.05
    sdf2[0, 0] = -3.0      # alpha -> 1 at the first sample
    sdf2[0, 1] = +3.0      # empty space: everything lands on the last (1e10) interval
    sdf2[1, 2, :4] = 0.5
    sdf2[1, 2, 4:] = -0.5
    beta = torch.tensor([0.1])
    rgb_map, fmap, xyz, mask, _ = ref_nerf.Render.volume_integration(
        rgb=rgb, sdf=sdf2, features=feat, z_vals=z, rays_d=rays_d, pts=pts, sigmoid_beta=beta)
    out.update(vi_z=z, vi_rays_d=rays_d, vi_pts=pts, vi_sdf=sdf2, vi_beta=beta,
               vi_rgb_map=rgb_map, vi_feature_map=fmap, vi_xyz=xyz, vi_mask=mask)
    save("siren", **out)
    # the branches no released config takes: raw density (with_sdf=False, softplus) and force_background, separately and together
    raw = torch.randn(B, R, N, 1) * 2.0
    raw[0, 0] = 25.0       # above softplus's threshold
    br = {"_src": "cips3d/nerf_utils.py:288-310", "rgb": rgb, "feat": feat, "z": z, "rays_d": rays_d, "pts": pts, "raw": raw,
          "sdf": sdf2, "beta": beta}
    for tag, kw in (("raw", dict(sdf=raw, with_sdf=False)), ("fb", dict(sdf=sdf2, sigmoid_beta=beta, force_background=True)),
                    ("raw_fb", dict(sdf=raw, with_sdf=False, force_background=True))):
        rm, fm, xz, mk, _ = ref_nerf.Render.volume_integration(rgb=rgb, features=feat, z_vals=z, rays_d=rays_d, pts=pts, **kw)
        br.update({f"{tag}_rgb_map": rm, f"{tag}_feature_map": fm, f"{tag}_xyz": xz, f"{tag}_mask": mk})
    save("vi_branches", **br)


def g_renderer_raw():
    """VolumeFeatureRenderer(with_sdf=False).forward on explicit sample points: the sigma head's output is a raw density
    (softplus).  Three sigma-head biases: the seeded one, 20 (the raw values straddle softplus's threshold), -4 (thin medium)."""
    out = {"_src": "cips3d/volume_renderer.py:163-303 with with_sdf=False; cips3d/nerf_utils.py:288-297"}
    torch.manual_seed(31)
    W, D = 32, 2
    ren = ref_vr.VolumeFeatureRenderer(N_layers_renderer=D, input_dim=3, hidden_dim=W, style_dim=W, view_dim=3, with_sdf=False,
                                       output_features=True)
    out.update({"sd." + k: v.clone() for k, v in ren.state_dict().items()})
    B, R, N = 2, 24, 8
    near, far = torch.full((B, 1, 1), 0.88), torch.full((B, 1, 1), 1.12)
    rays_d = torch.nn.functional.normalize(torch.randn(B, R, 3) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1) * 1.1
    viewdirs = torch.nn.functional.normalize(rays_d, dim=-1)
    rays_o = torch.tensor([0.0, 0.0, 1.0]).expand(B, R, 3) + 0.05 * torch.randn(B, R, 3)
    z = ref_nerf.Render.get_z_vals(near, far, rays_d.view(B, R, 1, 3), N, perturb=True).view(B, R, N)
    pts = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
    styles = torch.randn(B, D + 1, W) * 0.5
    out.update(pts=pts, rays_d=rays_d, viewdirs=viewdirs, z=z, near=near, far=far, styles=styles)
    for tag, bias in (("seeded", None), ("b20", 20.0), ("bm4", -4.0)):
        if bias is not None:
            ren.network.sigma_linear.bias.fill_(bias)
        out[f"{tag}_bias"] = ren.network.sigma_linear.bias.clone()
        rgb_map, fmap, raw, mask, xyz, _ = ren(pts.clone(), rays_d, viewdirs, z, near, far, styles=styles)
        out.update({f"{tag}_rgb_map": rgb_map, f"{tag}_feature_map": fmap, f"{tag}_raw": raw, f"{tag}_mask": mask,
                    f"{tag}_xyz": xyz})
    assert float((out["b20_raw"] > 20).float().mean()) not in (0.0, 1.0)          # both sides of the threshold
    save("renderer_raw", **out)


# ---------------------------------------------------------------- 4/5. ops
def g_ops():
    out = {"_src": "op/upfirdn2d.py:146-201; op/fused_act.py:87-119"}
    torch.manual_seed(4)
    k4 = ref.make_kernel([1, 3, 3, 1])
    k3 = ref.make_kernel([1, 2, 1])
    k2 = ref.make_kernel([1, 1])
    cases = [
        ("blur_up1", (1, 2, 7, 7), k4 * 4, 1, 1, (2, 2)),
        ("up2", (1, 3, 8, 8), k4 * 4, 2, 1, (2, 1)),
        ("down2", (1, 1, 9, 9), k4, 1, 2, (1, 1)),
        ("k3", (2, 2, 6, 5), k3, 1, 1, (1, 1)),
        ("k2_up2", (1, 2, 5, 6), k2 * 4, 2, 1, (1, 0)),
        ("crop", (1, 1, 10, 10), k4, 1, 1, (-1, 2)),
        ("big_up2", (2, 3, 33, 35), k4 * 4, 2, 1, (2, 1)),
        ("big_blur", (1, 5, 63, 63), k4 * 4, 1, 1, (2, 2)),
        ("down2_odd", (1, 2, 17, 16), k4, 1, 2, (2, 1)),
    ]
    names = []
    for name, shp, k, up, down, pad in cases:
        x = torch.randn(*shp)
        out[f"ufd_{name}_x"] = x
        out[f"ufd_{name}_k"] = k
        out[f"ufd_{name}_cfg"] = np.array([up, down, pad[0], pad[1]])
        out[f"ufd_{name}_y"] = ref_op.upfirdn2d(x, k, up=up, down=down, pad=pad)
        names.append(name)
    out["ufd_names"] = np.array(names)
    for name, shp, scale, use_b in (("2d_g1", (4, 32), 1.0, True), ("2d_gs", (4, 32), 2 ** 0.5, True),
                                    ("4d", (2, 8, 5, 5), 2 ** 0.5, True), ("4d_nob", (2, 8, 5, 5), 2 ** 0.5, False),
                                    ("3d", (2, 6, 7), 1.0, True)):
        x = torch.randn(*shp)
        b = torch.randn(shp[1]) if use_b else None
        out[f"flr_{name}_x"] = x
        if use_b:
            out[f"flr_{name}_b"] = b
        out[f"flr_{name}_scale"] = np.float32(scale)
        out[f"flr_{name}_y"] = ref_op.fused_leaky_relu(x, b, scale=scale)
    save("ops", **out)


# ---------------------------------------------------------------- 6/7. decoder blocks
def g_modconv():
    out = {"_src": "models/model_v3.py:218-314,418-482"}
    torch.manual_seed(6)
    B, Cin, Cout, H, S = 2, 8, 12, 6, 16
    names = []
    for k in (1, 3):
        for up in (False, True):
            for demod in (True, False):
                tag = f"k{k}_up{int(up)}_d{int(demod)}"
                m = ref.ModulatedConv2d(Cin, Cout, k, S, demodulate=demod, upsample=up)
                m.modulation.bias.data += torch.randn(Cin) * 0.2
                x = torch.randn(B, Cin, H, H)
                st = torch.randn(B, S)
                out.update({f"mc_{tag}.sd.{n}": v for n, v in m.state_dict().items()})
                out.update({f"mc_{tag}.x": x, f"mc_{tag}.style": st, f"mc_{tag}.y": m(x, st)})
                names.append(tag)
    out["mc_names"] = np.array(names)
    for up in (False, True):
        tag = f"up{int(up)}"
        sc = ref.StyledConv(Cin, Cout, 1, S, upsample=up)
        sc.noise.weight.data.fill_(0.3)
        sc.activate.bias.data = torch.randn(Cout) * 0.2
        x = torch.randn(B, Cin, H, H)
        st = torch.randn(B, S)
        Ho = H * 2 if up else H
        nz = torch.randn(1, 1, Ho, Ho)
        out.update({f"sc_{tag}.sd.{n}": v for n, v in sc.state_dict().items()})
        out.update({f"sc_{tag}.x": x, f"sc_{tag}.style": st, f"sc_{tag}.noise": nz, f"sc_{tag}.y": sc(x, st, noise=nz)})
        tr = ref.ToRGB(Cout, S, upsample=up)
        tr.bias.data = torch.randn(1, 3, 1, 1) * 0.1
        xin = torch.randn(B, Cout, Ho, Ho)
        skip = torch.randn(B, 3, H, H)
        out.update({f"rgb_{tag}.sd.{n}": v for n, v in tr.state_dict().items()})
        out.update({f"rgb_{tag}.x": xin, f"rgb_{tag}.style": st, f"rgb_{tag}.skip": skip,
                    f"rgb_{tag}.y": tr(xin, st, skip=skip), f"rgb_{tag}.y_noskip": tr(xin, st)})
    save("modconv", **out)


# ---------------------------------------------------------------- 8. tiny generator end-to-end
def _run_ref(G, zs, cam, img_size, nerf_cfg, noise_bufs, truncation=1.0, means=None, **kw):
    if means is not None:
        G.style_render_mean, G.style_decoder_mean = means
    r = G(zs=zs, cam_poses=cam[0], focals=cam[1], img_size=img_size, near=cam[2], far=cam[3],
          noise_bufs=noise_bufs, truncation=truncation, nerf_cfg=nerf_cfg,
          return_xyz=True, return_sdf=True, **kw)
    return {k: v for k, v in r.items() if v is not None}


def g_tiny_generator():
    out = {"_src": "models/model_v3.py:875-1042 (tiny G_cfg, SURVEY Appendix D)"}
    for tag, hidden, D, ks in (("h32_d2", 32, 2, 1), ("h32_d3", 32, 3, 1), ("h32_d2_k3", 32, 2, 3)):
        cfg = configs.tiny_G_cfg(hidden=hidden, N_layers_renderer=D, kernel_size=ks)
        G = ref.Generator(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        sd = weights.synth_state_dict(shapes, seed=7)
        G.load_state_dict(sd, strict=True)
        out[f"{tag}.keys"] = np.array(list(shapes.keys()))
        out.update({f"{tag}.sd.{k}": v for k, v in sd.items()})
        g = torch.Generator().manual_seed(11)
        zs = [torch.randn(2, hidden, generator=g), torch.randn(2, hidden, generator=g)]
        locs = torch.tensor([[0.2, -0.05], [-0.6, 0.1]])
        cam = ref_nerf.Camera.generate_camera_params(img_size=8, device="cpu", locations=locs,
                                                     fov_ang=6, dist_radius=0.12)
        nb = [torch.randn(*b.shape, generator=g) for b in G.create_noise_bufs(8, "cpu")]
        means = (torch.randn(1, hidden, generator=g) * 0.3, torch.randn(1, 32, generator=g) * 0.3)
        out.update({f"{tag}.z0": zs[0], f"{tag}.z1": zs[1], f"{tag}.locs": locs,
                    f"{tag}.mean_r": means[0], f"{tag}.mean_d": means[1]})
        out.update({f"{tag}.noise{i}": b for i, b in enumerate(nb)})
        for vtag, ncfg, trunc in (("a", dict(N_samples=6, perturb=False, static_viewdirs=False), 1.0),
                                  ("b", dict(N_samples=5, perturb=False, static_viewdirs=True), 0.5)):
            r = _run_ref(G, zs, cam, 8, ncfg, nb, truncation=trunc, means=means)
            out.update({f"{tag}.{vtag}.{k}": v for k, v in r.items()})
        # perturbed run with the uniform captured
        torch.manual_seed(5)
        r = _run_ref(G, zs, cam, 8, dict(N_samples=6, perturb=True, static_viewdirs=False), nb)
        torch.manual_seed(5)
        out[f"{tag}.c.u"] = torch.rand(2, 8, 8, 1)
        out.update({f"{tag}.c.{k}": v for k, v in r.items()})
        # mean latents with the z's captured (get_mean_latent, model_v3.py:1285-1297)
        torch.manual_seed(9)
        mr, md = G.get_mean_latent(64, "cpu")
        torch.manual_seed(9)
        out[f"{tag}.ml_zr"] = torch.randn(64, hidden)
        out[f"{tag}.ml_zd"] = torch.randn(64, hidden)
        out.update({f"{tag}.ml_r": mr, f"{tag}.ml_d": md})
    save("tiny_generator", **out)


# ---------------------------------------------------------------- 9. full size, strided samples
FULL_CASES = [
    # tag, resolution, D, N, static_viewdirs, truncation
    ("r256_d2_n24", 256, 2, 24, False, 1.0),
    ("r256_d8_n24", 256, 8, 24, False, 1.0),
    ("r1024_d2_n24", 1024, 2, 24, False, 1.0),
    ("r256_d6_n64_static_trunc", 256, 6, 64, True, 0.5),
]
STRIDE = 37


def full_inputs(cfg, seed=12345):
    """Inputs shared by make_golden and the GPU parity test: closed form (weights.synth_inputs), no RNG."""
    zs, nb, means = weights.synth_inputs(cfg, batch=1, seed=seed)
    locs = torch.tensor([[0.31, -0.08]])
    return zs, locs, nb, means


def g_full():
    out = {"_src": "models/model_v3.py:875-1042 at release shapes; strided samples (every 37th)",
           "stride": np.int64(STRIDE)}
    for tag, res, D, N, static, trunc in FULL_CASES:
        cfg = configs.ffhq_G_cfg(resolution=res, N_layers_renderer=D)
        G = ref.Generator(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        sd = weights.synth_state_dict(shapes, seed=1)
        G.load_state_dict(sd, strict=True)
        zs, locs, nb, means = full_inputs(cfg)
        cam = ref_nerf.Camera.generate_camera_params(img_size=64, device="cpu", locations=locs,
                                                     fov_ang=6, dist_radius=0.12)
        ncfg = dict(N_samples=N, perturb=False, static_viewdirs=static)
        r = _run_ref(G, zs, cam, 64, ncfg, nb, truncation=trunc, means=means)
        out[f"{tag}.sd_checksum"] = np.uint64(weights.state_dict_checksum(sd))
        out[f"{tag}.nkeys"] = np.int64(len(shapes))
        out[f"{tag}.rgb_s"] = r["rgb"].flatten()[::STRIDE]
        out[f"{tag}.rgb_absmax"] = r["rgb"].abs().max()
        out[f"{tag}.thumb_rgb"] = r["thumb_rgb"]
        out[f"{tag}.mask"] = r["mask"]
        out[f"{tag}.depth"] = r["depth"]
        out[f"{tag}.xyz"] = r["xyz"]
        out[f"{tag}.sdf_s"] = r["sdf"].flatten()[::STRIDE]
        # fp64 rerun: the reference's own fp32 noise floor for this case
        G64 = G.double()
        r64 = _run_ref(G64, [z.double() for z in zs], [c.double() for c in cam[:4]], 64, ncfg,
                       [b.double() for b in nb], truncation=trunc, means=tuple(m.double() for m in means))
        out[f"{tag}.noise_floor_rgb"] = (r64["rgb"].float() - r["rgb"]).abs().max()
        out[f"{tag}.noise_floor_thumb"] = (r64["thumb_rgb"].float() - r["thumb_rgb"]).abs().max()
        print(tag, "rgb absmax", float(out[f"{tag}.rgb_absmax"]), "fp32 noise floor rgb",
              float(out[f"{tag}.noise_floor_rgb"]), "thumb", float(out[f"{tag}.noise_floor_thumb"]))
    save("full_size", **out)


# ---------------------------------------------------------------- 9b. one inversion-like step: loss + gradients
def g_backward():
    """Forward + backward of the reference generator as the flip-inversion loop drives it
    (models/projector_v10.py:211-277: camera from `locations`, W+ styles passed in, renderer_detach=False, batch = image +
    flip) with a surrogate loss (VGG weights are not obtainable): mean((rgb-t)^2) + 50 mean((thumb-t_thumb)^2)."""
    out = {"_src": "models/projector_v10.py:211-277 + models/model_v3.py:875-1042 under autograd (tiny G_cfg)"}
    for tag, hidden, D, static in (("h32_d2", 32, 2, True), ("h32_d3", 32, 3, False)):
        cfg = configs.tiny_G_cfg(hidden=hidden, N_layers_renderer=D, kernel_size=1)
        G = ref.Generator(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        G.load_state_dict(weights.synth_state_dict(shapes, seed=7), strict=True)
        G.requires_grad_(False)
        G.decoder.requires_grad_(True)
        g = torch.Generator().manual_seed(31)
        locs = torch.tensor([[0.3, 0.1], [-0.3, 0.1]]).requires_grad_(True)
        w_r = (0.5 * torch.randn(2, D + 1, hidden, generator=g)).requires_grad_(True)
        w_d = (0.5 * torch.randn(2, G.decoder.n_latent, 32, generator=g)).requires_grad_(True)
        nb = [torch.randn(*b.shape, generator=g).requires_grad_(True) for b in G.create_noise_bufs(8, "cpu")]
        t_rgb = torch.randn(2, 3, 32, 32, generator=g)
        t_thumb = torch.randn(2, 3, 8, 8, generator=g)
        ncfg = dict(N_samples=6, perturb=False, static_viewdirs=static)
        with torch.enable_grad():
            cam = ref_nerf.Camera.generate_camera_params(img_size=8, device="cpu", locations=locs, fov_ang=6, dist_radius=0.12)
            r = G(zs=[None, None], style_render=w_r, style_decoder=w_d, cam_poses=cam[0], focals=cam[1], img_size=8,
                  near=cam[2], far=cam[3], noise_bufs=nb, nerf_cfg=ncfg, renderer_detach=False)
            loss = ((r["rgb"] - t_rgb) ** 2).mean() + 50 * ((r["thumb_rgb"] - t_thumb) ** 2).mean()
            loss.backward()
        out.update({f"{tag}.locs": locs.detach(), f"{tag}.w_r": w_r.detach(), f"{tag}.w_d": w_d.detach(),
                    f"{tag}.t_rgb": t_rgb, f"{tag}.t_thumb": t_thumb, f"{tag}.loss": loss.detach(),
                    f"{tag}.rgb": r["rgb"].detach(), f"{tag}.thumb": r["thumb_rgb"].detach(),
                    f"{tag}.g.locs": locs.grad, f"{tag}.g.w_r": w_r.grad, f"{tag}.g.w_d": w_d.grad})
        out.update({f"{tag}.noise{i}": b.detach() for i, b in enumerate(nb)})
        out.update({f"{tag}.g.noise{i}": b.grad for i, b in enumerate(nb)})
        for name, p in G.decoder.named_parameters():
            if p.grad is not None:
                out[f"{tag}.g.dec.{name}"] = p.grad
        print(tag, "loss", float(loss), "|dlocs|", locs.grad.abs().max().item(), "|dw_r|", w_r.grad.abs().max().item())
    save("backward", **out)


# ---------------------------------------------------------------- 9c. BASELINE config 5 at its stated size
CONFIG5_STRIDE = 53


def g_config5():
    """One flip-inversion step of the reference at the size BASELINE config 5 states: CompCars camera (fov 15, radius 0.3),
    256^2 output, D = 6 NeRF layers, 64x64 rays x 24 samples, static view directions, batch 2 (image + mirrored view),
    forward + backward of the surrogate loss (models/projector_v10.py:211-277 call pattern).  Stores the loss, strided
    outputs and strided gradients of every leaf the loop optimises."""
    res, D, N = 256, 6, 24
    out = {"_src": "models/projector_v10.py:211-277 + models/model_v3.py:875-1042 under autograd; CompCars 256^2, D=6, B=2",
           "stride": np.int64(CONFIG5_STRIDE)}
    cfg = configs.ffhq_G_cfg(resolution=res, N_layers_renderer=D)
    G = ref.Generator(**cfg).eval()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    sd = weights.synth_state_dict(shapes, seed=2)
    G.load_state_dict(sd, strict=True)
    out["sd_checksum"] = np.uint64(weights.state_dict_checksum(sd))
    G.requires_grad_(False)
    G.decoder.requires_grad_(True)
    ncfg = dict(N_samples=N, perturb=False, static_viewdirs=True)
    cam_cfg = configs.COMPCARS_CAM_CFG

    def one_step(G, dt):
        locs, w_r, w_d, nb, t_rgb, t_thumb = (weights.synth_inversion_inputs(cfg, res))
        w_r, w_d, t_rgb, t_thumb = (t.to(dt) for t in (w_r, w_d, t_rgb, t_thumb))     # the camera construction is fp32-only
        nb = [b.to(dt).requires_grad_(True) for b in nb]
        locs.requires_grad_(True); w_r.requires_grad_(True); w_d.requires_grad_(True)
        for p in G.parameters():
            p.grad = None
        with torch.enable_grad():
            cam = ref_nerf.Camera.generate_camera_params(img_size=64, device="cpu", locations=locs, fov_ang=cam_cfg["fov_ang"],
                                                         dist_radius=cam_cfg["dist_radius"])
            cam = [c.to(dt) for c in cam[:4]]
            r = G(zs=[None, None], style_render=w_r, style_decoder=w_d, cam_poses=cam[0], focals=cam[1], img_size=64,
                  near=cam[2], far=cam[3], noise_bufs=nb, nerf_cfg=ncfg, renderer_detach=False, return_xyz=True)
            loss = ((r["rgb"] - t_rgb) ** 2).mean() + 50 * ((r["thumb_rgb"] - t_thumb) ** 2).mean()
            loss.backward()
        grads = {"locs": locs.grad, "w_r": w_r.grad, "w_d": w_d.grad}
        grads.update({f"noise{i}": b.grad for i, b in enumerate(nb)})
        grads.update({f"dec.{name}": p.grad for name, p in G.decoder.named_parameters() if p.grad is not None})
        return loss.detach(), {k: v.detach() for k, v in r.items() if v is not None}, grads

    loss, r, g32 = one_step(G, torch.float32)
    assert g32["w_d"].shape[1] == G.decoder.n_latent
    _, r64, g64 = one_step(G.double(), torch.float64)       # the reference's own fp32 noise floor, per gradient
    st = CONFIG5_STRIDE
    out.update(loss=loss, rgb_s=r["rgb"].flatten()[::st], rgb_absmax=r["rgb"].abs().max(), thumb=r["thumb_rgb"], xyz=r["xyz"],
               mask=r["mask"], depth=r["depth"], noise_floor_rgb=(r64["rgb"].float() - r["rgb"]).abs().max())
    worst = 0.0
    for k, g in g32.items():
        small = k in ("locs", "w_r", "w_d")
        out[f"g.{k}" if small else f"g.{k}_s"] = g if small else g.flatten()[::st]
        out[f"g.{k}_absmax"] = g.abs().max()
        out[f"g.{k}_floor"] = (g64[k].float() - g).abs().max()
        worst = max(worst, float(out[f"g.{k}_floor"]) / (float(out[f"g.{k}_absmax"]) + 1e-30))
    print("config5 loss", float(loss), "|dlocs|", g32["locs"].abs().max().item(), "|dw_r|", g32["w_r"].abs().max().item(),
          "|dw_d|", g32["w_d"].abs().max().item(), "worst fp32 noise floor / absmax over all gradients", worst)
    save("config5", **out)


# ---------------------------------------------------------------- 10. on-disk formats
def g_ckpt_tiny():
    """A checkpoint directory and an inversion file as the reference's writers lay them out, holding the
    reference generator's own tensors: G_ema.pth = state_dict of the imported reference Generator
    (scripts/train_v10.py:496-523 via tl2 save_models(save_module=False)); w.pth = the dict of
    models/projector_v10.py:1044-1055.  config_command.yaml is written by hand in the structure the loaders read
    (`list(load_yaml(...).values())[0].G_cfg`, tests/test_cips3dpp.py:705-706): tl2's dumper is not vendored."""
    import yaml
    d = os.path.join(HERE, "ckpt_tiny")
    os.makedirs(d, exist_ok=True)
    cfg = configs.tiny_G_cfg(hidden=32, N_layers_renderer=2, kernel_size=1)
    G = ref.Generator(**cfg).eval()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    G.load_state_dict(weights.synth_state_dict(shapes, seed=7), strict=True)
    torch.save(G.state_dict(), os.path.join(d, "G_ema.pth"))
    block = {"G_cfg": {"register_modules": ["exp.cips3d.models.model_v3"],
                       "name": "exp.cips3d.models.model_v3.Generator", **cfg},
             "G_kwargs": {"cam_cfg": dict(configs.FFHQ_CAM_CFG), "nerf_cfg": dict(configs.TRAIN_NERF_CFG)}}
    with open(os.path.join(d, "config_command.yaml"), "w") as f:
        yaml.safe_dump({"train_cips3d_ffhq_v10": block}, f, sort_keys=False)
    g = torch.Generator().manual_seed(21)
    w_dec = torch.zeros(2, G.decoder.n_latent, 32)
    w_dec[1] = 1.0
    torch.save({
        "azim": torch.tensor([0.25, -0.25]), "elev": torch.tensor([0.05, 0.05]),
        "w_render_opt": torch.zeros(1, G.N_layers_renderer + 1, 32), "w_decoder_opt": w_dec,
        "render_state_dict": G.renderer.state_dict(), "decoder_state_dict": G.decoder.state_dict(),
        "noise_bufs": [torch.randn(*b.shape, generator=g) for b in G.create_noise_bufs(8, "cpu")],
        "padding": 0,
    }, os.path.join(d, "w.pth"))
    for fn in sorted(os.listdir(d)):
        print(f"ckpt_tiny/{fn}: {os.path.getsize(os.path.join(d, fn)) / 1024:.1f} KB")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    jobs = dict(camera=g_camera, rays=g_rays, rays_stratified=g_rays_stratified, siren=g_siren, renderer_raw=g_renderer_raw, ops=g_ops, modconv=g_modconv,
                tiny_generator=g_tiny_generator, ckpt_tiny=g_ckpt_tiny, backward=g_backward)
    if a.full:
        jobs["full_size"] = g_full
        jobs["config5"] = g_config5
    for name, fn in jobs.items():
        if a.only is None or a.only == name:
            fn()
