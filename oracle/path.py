"""Functional CPU restatement of the reference generator forward (oracle).

Every function works on plain tensors and a flat ``state_dict`` (the
reference's parameter names), in whatever dtype the
inputs carry (fp32 for parity, fp64 for noise-floor studies).  No nn.Module,
no autograd requirements, no device code: this is the checker, not the product.

All ``file:line`` citations are relative to /root/reference/exp/.
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    "camera_params", "rays_in_world", "z_vals", "ray_points", "normalize_points",
    "film_siren", "siren_points", "volume_integration", "renderer_forward",
    "fused_leaky_relu", "upfirdn2d", "make_blur_kernel", "pixel_norm",
    "mapping_renderer", "mapping_decoder", "modulated_conv2d", "styled_conv",
    "to_rgb", "decoder_forward", "decoder_layout", "create_noise_bufs",
    "mean_latents", "generator_forward",
]


# --------------------------------------------------------------------------- camera
def _unit(v, eps):
    # F.normalize semantics: v / max(||v||, eps)
    return v / v.norm(dim=-1, keepdim=True).clamp_min(eps)


def camera_params(locations, img_size, fov_ang=6, dist_radius=0.12, up=None):
    """cips3d/nerf_utils.py:344-436 (`locations` branch) and :466-564 (custom `up`).

    locations (B,2) = (azim, elev) -> extrinsics (B,3,4), focal/near/far (B,1,1), viewpoint (B,2)
    """
    azim = locations[:, 0:1]
    elev = locations[:, 1:2]
    B = azim.shape[0]
    dt = locations.dtype
    dist = torch.ones(B, 1, dtype=dt)
    near = (dist - dist_radius).unsqueeze(-1)
    far = (dist + dist_radius).unsqueeze(-1)
    fov = fov_ang * torch.ones(B, 1, dtype=dt) * math.pi / 180
    focal = 0.5 * img_size / torch.tan(fov).unsqueeze(-1)
    cam_dir = torch.cat([torch.cos(elev) * torch.sin(azim), torch.sin(elev),
                         torch.cos(elev) * torch.cos(azim)], dim=1)
    cam_loc = dist * cam_dir
    if up is None:
        up = torch.tensor([[0.0, 1.0, 0.0]], dtype=dt).expand(B, 3)
    z_ax = _unit(cam_dir, 1e-5)
    x_ax = _unit(torch.linalg.cross(up, z_ax, dim=1), 1e-5)
    y_ax = _unit(torch.linalg.cross(z_ax, x_ax, dim=1), 1e-5)
    degenerate = torch.isclose(x_ax, torch.zeros((), dtype=dt), atol=5e-3).all(dim=1, keepdim=True)
    if degenerate.any():
        x_ax = torch.where(degenerate, _unit(torch.linalg.cross(y_ax, z_ax, dim=1), 1e-5), x_ax)
    rot_t = torch.stack([x_ax, y_ax, z_ax], dim=2)  # columns are the camera axes == R^T
    extr = torch.cat([rot_t, cam_loc.unsqueeze(-1)], dim=-1)
    return extr, focal, near, far, torch.cat([azim, elev], dim=1)


# --------------------------------------------------------------------------- rays / samples
def rays_in_world(focal, img_size, c2w, static_viewdirs=False):
    """cips3d/nerf_utils.py:18-66.  Returns rays_o, rays_d, viewdirs, each (B,S,S,3)."""
    S = img_size
    dt = focal.dtype
    centres = torch.linspace(0.5, S - 0.5, S, dtype=dt)
    y = centres.view(1, S, 1).expand(1, S, S)   # rows
    x = centres.view(1, 1, S).expand(1, S, S)   # cols
    B = focal.shape[0]
    d_cam = torch.stack([(x - S * 0.5) / focal, -(y - S * 0.5) / focal,
                         -torch.ones(B, S, S, dtype=dt)], dim=-1)
    rays_d = (d_cam.unsqueeze(-2) * c2w[:, None, None, :3, :3]).sum(-1)
    rays_o = c2w[:, None, None, :3, 3].expand_as(rays_d)
    v = d_cam if static_viewdirs else rays_d
    return rays_o, rays_d, _unit(v, 1e-12)


def z_vals(near, far, B, H, W, N, perturb_u=None):
    """cips3d/nerf_utils.py:69-121, offset-sampling branch.

    perturb_u: None (perturb=False) or the injected per-ray uniform (B,H,W,1).
    """
    dt = near.dtype
    near = near.view(B, 1, 1, 1).expand(B, H, W, 1)
    far = far.view(B, 1, 1, 1).expand(B, H, W, 1)
    t = torch.linspace(0.0, 1.0 - 1.0 / N, N, dtype=dt).view(1, 1, 1, N)
    z = near * (1.0 - t) + far * t
    if perturb_u is not None:
        upper = torch.cat([z[..., 1:], far], dim=-1)
        z = z + (upper - z) * perturb_u
    return z


def z_vals_stratified(near, far, B, H, W, N, perturb_t=None):
    """cips3d/nerf_utils.py:69-121, classic stratified branch (offset_sampling=False; :98-117).

    perturb_t: None (perturb=False) or the injected per-sample uniform (B,H,W,N).
    """
    dt = near.dtype
    near = near.view(B, 1, 1, 1).expand(B, H, W, 1)
    far = far.view(B, 1, 1, 1).expand(B, H, W, 1)
    t = torch.linspace(0.0, 1.0, N, dtype=dt).view(1, 1, 1, N)
    z = near * (1.0 - t) + far * t
    if perturb_t is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], dim=-1)
        lower = torch.cat([z[..., :1], mids], dim=-1)
        z = lower + (upper - lower) * perturb_t
    return z


def ray_points(rays_o, rays_d, z):
    """cips3d/nerf_utils.py:136-170."""
    return rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)


def normalize_points(pts, near, far):
    """cips3d/nerf_utils.py:124-133."""
    shape = [-1] + [1] * (pts.dim() - 1)
    return pts * 2 / (far - near).view(*shape)


# --------------------------------------------------------------------------- FiLM-SIREN
def _affine(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def film_siren(sd, prefix, x, style):
    """cips3d/volume_renderer.py:39-85 with LinearLayer :15-35 (gamma: 15x+30, beta: 0.25x)."""
    B = style.shape[0]
    pre = _affine(sd, prefix, x)
    bshape = [B] + [1] * (pre.dim() - 2) + [-1]
    gamma = (15.0 * _affine(sd, prefix + ".gamma", style) + 30.0).view(*bshape)
    beta = (0.25 * _affine(sd, prefix + ".beta", style)).view(*bshape)
    return torch.sin(gamma * pre + beta)


def siren_points(sd, prefix, pts_n, viewdirs, styles, D):
    """cips3d/volume_renderer.py:133-160.  pts_n (B,R,N,3), viewdirs (B,R,3), styles (B,D+1,S)."""
    h = pts_n
    for i in range(D):
        h = film_siren(sd, f"{prefix}.pts_linears.{i}", h, styles[:, i])
    sdf = _affine(sd, prefix + ".sigma_linear", h)
    dirs = viewdirs.unsqueeze(-2).expand(*h.shape[:-1], viewdirs.shape[-1])
    feat = film_siren(sd, prefix + ".views_linears", torch.cat([h, dirs], dim=-1), styles[:, -1])
    rgb = _affine(sd, prefix + ".rgb_linear", feat)
    return rgb, sdf, feat


def volume_integration(rgb, sdf, feat, z, rays_d, pts, sigmoid_beta, with_sdf=True, force_background=False):
    """cips3d/nerf_utils.py:230-338: with_sdf branch (:276-286) or raw density through softplus (:288-297, noise-free);
    force_background (:309-310).

    rgb (..,N,3) sdf (..,N,1) feat (..,N,C) z (..,N) rays_d (..,3) pts (..,N,3)
    -> rgb_map (..,3), feature_map (..,C), xyz (..,3), mask (..,2) = [w_last, -|xyz|]
    """
    dnorm = rays_d.norm(dim=-1, keepdim=True)
    delta = torch.cat([z[..., 1:] - z[..., :-1],
                       torch.full_like(dnorm, 1e10)], dim=-1) * dnorm
    sigma = torch.sigmoid(-sdf / sigmoid_beta) / sigmoid_beta if with_sdf else F.softplus(sdf)
    alpha = 1 - torch.exp(-sigma * delta.unsqueeze(-1))
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1, :]), 1.0 - alpha + 1e-10],
                                    dim=-2), dim=-2)[..., :-1, :]
    w = alpha * trans
    if force_background:
        w = torch.cat([w[..., :-1, :], 1 - w[..., :-1, :].sum(dim=-2, keepdim=True)], dim=-2)
    rgb_map = -1 + 2 * (w * torch.sigmoid(rgb)).sum(-2)
    feature_map = (w * feat).sum(-2)
    xyz = (w * pts).sum(-2)
    mask = torch.cat([w[..., -1, :], -xyz.norm(dim=-1, keepdim=True)], dim=-1)
    return rgb_map, feature_map, xyz, mask


def renderer_forward(sd, prefix, pts, rays_d, viewdirs, z, near, far, styles, D, with_sdf=True):
    """cips3d/volume_renderer.py:192-303 (no ray chunking: chunking does not change values); with_sdf=False: the sigma
    head's output is a raw density (nerf_utils.py:288-297)."""
    pts_n = normalize_points(pts, near, far)
    rgb, sdf, feat = siren_points(sd, prefix + ".network", pts_n, viewdirs, styles, D)
    rgb_map, feature_map, xyz, mask = volume_integration(
        rgb, sdf, feat, z, rays_d, pts, sd[prefix + ".sigmoid_beta"], with_sdf=with_sdf)
    return rgb_map, feature_map, sdf, mask, xyz


# --------------------------------------------------------------------------- op-level
def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    """op/fused_act.py:105-116 (CPU branch; the slope argument is ignored there, 0.2 is used)."""
    if bias is not None:
        x = x + bias.view(1, -1, *([1] * (x.dim() - 2)))
    return F.leaky_relu(x, 0.2) * scale


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:146-201: zero-insert by `up`, pad/crop, true convolution with `kernel`,
    keep every `down`-th sample.  x (B,C,H,W), kernel (kh,kw)."""
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    y = x.new_zeros(B * C, 1, H * up, W * up)
    y[:, :, ::up, ::up] = x.reshape(B * C, 1, H, W)
    y = F.pad(y, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    y = y[:, :, max(-p0, 0): y.shape[2] - max(-p1, 0), max(-p0, 0): y.shape[3] - max(-p1, 0)]
    y = F.conv2d(y, kernel.flip(0, 1).view(1, 1, kh, kw).to(y.dtype))
    y = y[:, :, ::down, ::down]
    return y.reshape(B, C, y.shape[2], y.shape[3])


def make_blur_kernel(taps=(1, 3, 3, 1), gain=1.0, dtype=torch.float32):
    """models/model_v3.py:73-81."""
    k = torch.tensor(taps, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return (k / k.sum() * gain).to(dtype)


# --------------------------------------------------------------------------- mapping nets
def pixel_norm(x):
    """models/model_v3.py:32-37."""
    return x * torch.rsqrt((x ** 2).mean(dim=1, keepdim=True) + 1e-8)


def _equal_linear(sd, prefix, x, lr_mul=1.0, activate=False):
    """models/model_v3.py:183-210."""
    w = sd[prefix + ".weight"]
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    if activate:
        return fused_leaky_relu(F.linear(x, w * scale), sd[prefix + ".bias"] * lr_mul)
    return F.linear(x, w * scale, sd[prefix + ".bias"] * lr_mul)


def _nerf_style(sd, z, n_layers):
    """models/model_v3.py:40-65,1420-1433: n x [linear (no bias) -> lrelu(x+b)*1]."""
    h = z
    for i in range(n_layers):
        h = fused_leaky_relu(F.linear(h, sd[f"style.{i}.weight"]), sd[f"style.{i}.bias"], scale=1)
    return h


def _decoder_style(sd, z, n_layers, lr_mul):
    """models/model_v3.py:1380-1399."""
    h = pixel_norm(z)
    for i in range(1, n_layers + 1):
        h = _equal_linear(sd, f"style_decoder.{i}", h, lr_mul=lr_mul, activate=True)
    return h


def mapping_renderer(sd, cfg, z, truncation=1.0, mean=None):
    """models/model_v3.py:1402-1418 -> (B, D+1, style_dim)."""
    w = _nerf_style(sd, z, cfg["mapping_renderer_cfg"]["N_layers"])
    if truncation < 1:
        w = mean + truncation * (w - mean)
    D = cfg["renderer_cfg"]["N_layers_renderer"]
    return w.unsqueeze(1).repeat(1, D + 1, 1)


def mapping_decoder(sd, cfg, z, truncation=1.0, mean=None):
    """models/model_v3.py:1350-1378 -> (B, n_latent, style_dim)."""
    mc = cfg["mapping_decoder_cfg"]
    w = _decoder_style(sd, z, mc["N_layers"], mc["lr_mul_mapping"])
    if truncation < 1:
        w = mean + truncation * (w - mean)
    return w.unsqueeze(1).repeat(1, decoder_layout(cfg)["n_latent"], 1)


def mean_latents(sd, cfg, z_render, z_decoder):
    """models/model_v3.py:1285-1297 with the 10 000 random z's injected."""
    mc = cfg["mapping_decoder_cfg"]
    return (_nerf_style(sd, z_render, cfg["mapping_renderer_cfg"]["N_layers"]).mean(0, keepdim=True),
            _decoder_style(sd, z_decoder, mc["N_layers"], mc["lr_mul_mapping"]).mean(0, keepdim=True))


# --------------------------------------------------------------------------- decoder
def _bf16_round(t):
    """round-to-nearest-even to bfloat16 and back (the operand rounding of the bf16 compute mode, BASELINE config 3)"""
    return t.to(torch.bfloat16).to(t.dtype)


def modulated_conv2d(sd, prefix, x, style, demodulate=True, upsample=False, bf16_gemm=False, bf16_store=False):
    """models/model_v3.py:218-314 (plain and up-sampling branches).  bf16_gemm (not in the reference): both GEMM operands
    are rounded to bf16, accumulation stays fp32 -- the restatement the bf16 decoder mode is checked against."""
    B, Cin, H, W = x.shape
    weight = sd[prefix + ".weight"]            # (1, Cout, Cin, k, k)
    Cout, k = weight.shape[1], weight.shape[3]
    s = _equal_linear(sd, prefix + ".modulation", style).view(B, 1, Cin, 1, 1)
    w = (1 / math.sqrt(Cin * k * k)) * weight * s
    if demodulate:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).view(B, Cout, 1, 1, 1)
    if bf16_gemm:
        w, x = _bf16_round(w), _bf16_round(x)
    if upsample:
        wt = w.transpose(1, 2).reshape(B * Cin, Cout, k, k)
        y = F.conv_transpose2d(x.reshape(1, B * Cin, H, W), wt, padding=0, stride=2, groups=B)
        y = y.view(B, Cout, y.shape[2], y.shape[3])
        if bf16_store:   # (not in the reference) the pre-FIR result kept as bf16: for k = 1 the transposed conv's output is
            y = _bf16_round(y)     # the low-resolution GEMM result on the even lattice and zeros elsewhere
        p = (4 - 2) - (k - 1)
        return upfirdn2d(y, sd[prefix + ".blur.kernel"], pad=((p + 1) // 2 + 1, p // 2 + 1))
    y = F.conv2d(x.reshape(1, B * Cin, H, W), w.view(B * Cout, Cin, k, k), padding=k // 2, groups=B)
    y = y.view(B, Cout, y.shape[2], y.shape[3])
    return _bf16_round(y) if bf16_store else y      # (not in the reference: see decoder_forward)


def styled_conv(sd, prefix, x, style, noise, upsample=False, bf16_gemm=False, bf16_store=False):
    """models/model_v3.py:418-454 with NoiseInjection :327-341 (explicit noise only)."""
    y = modulated_conv2d(sd, prefix + ".conv", x, style, demodulate=True, upsample=upsample, bf16_gemm=bf16_gemm,
                         bf16_store=bf16_store)
    y = y + sd[prefix + ".noise.weight"] * noise
    return fused_leaky_relu(y, sd[prefix + ".activate.bias"])


def to_rgb(sd, prefix, x, style, skip=None, upsample=False):
    """models/model_v3.py:457-482."""
    y = modulated_conv2d(sd, prefix + ".conv", x, style, demodulate=False) + sd[prefix + ".bias"]
    if skip is not None:
        if upsample:
            skip = upfirdn2d(skip, sd[prefix + ".upsample.kernel"], up=2, pad=(2, 1))
        y = y + skip
    return y


def decoder_layout(cfg):
    """Stage table of models/model_v3.py:564-574,668-729: per nominal stage (Cin, Cout, up)."""
    dc = cfg["decoder_cfg"]
    m = dc["channel_multiplier"]
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * m, 128: 128 * m, 256: 64 * m,
          512: 32 * m, 1024: 16 * m}
    lo, hi = int(math.log2(dc["size_start"])), int(math.log2(dc["size_end"]))
    stages, cin = [], ch[dc["size_start"]]
    for i in range(lo + 1, hi + 1):
        stages.append(dict(cin=cin, cout=ch[2 ** i], up=(2 ** i) in dc["upsample_list"]))
        cin = ch[2 ** i]
    return dict(c0=ch[dc["size_start"]], stages=stages,
                num_layers=(hi - lo) * 2 + 1, n_latent=(hi - lo) * 2 + 2)


def create_noise_bufs(cfg, start_size, generator=None, dtype=torch.float32):
    """models/model_v3.py:639-666 (shapes only; values from `generator`)."""
    sizes, cur = [start_size], start_size
    for st in decoder_layout(cfg)["stages"]:
        if st["up"]:
            cur *= 2
        sizes += [cur, cur]
    return [torch.randn(1, 1, s, s, generator=generator, dtype=dtype) for s in sizes]


def decoder_forward(sd, cfg, features, styles, noise, prefix="decoder", bf16_gemm=False, bf16_store=False):
    """models/model_v3.py:592-637."""
    lay = decoder_layout(cfg)
    out = styled_conv(sd, prefix + ".conv1", features, styles[:, 0], noise[0], bf16_gemm=bf16_gemm)
    skip = to_rgb(sd, prefix + ".to_rgb1", out, styles[:, 1])
    i = 1
    for s, st in enumerate(lay["stages"]):
        # bf16_store (not in the reference; the product's "bf16_storage" mode): the conv result that a block's first StyledConv hands
        # to its FIR / activation is kept as bf16 -- in every up-sampling block, and in every block above the NeRF resolution that
        # does not up-sample (the 512 / 1024 blocks of a 256^2 generator: the product runs them as flat stages of the same kernel)
        store = bf16_store and (st["up"] or out.shape[-1] > features.shape[-1])
        out = styled_conv(sd, f"{prefix}.convs.{2 * s}", out, styles[:, i], noise[2 * s + 1],
                          upsample=st["up"], bf16_gemm=bf16_gemm, bf16_store=store)
        out = styled_conv(sd, f"{prefix}.convs.{2 * s + 1}", out, styles[:, i + 1], noise[2 * s + 2], bf16_gemm=bf16_gemm)
        skip = to_rgb(sd, f"{prefix}.to_rgbs.{s}", out, styles[:, i + 2], skip, upsample=st["up"])
        i += 2
    return skip


# --------------------------------------------------------------------------- generator
def generator_forward(sd, cfg, zs, cam_poses, focals, img_size, near, far, nerf_cfg,
                      noise_bufs, truncation=1.0, style_render=None, style_decoder=None,
                      style_render_mean=None, style_decoder_mean=None, perturb_u=None,
                      return_sdf=False, return_xyz=False, bf16_decoder=False):
    """models/model_v3.py:875-1042 for the inference configuration (explicit noise_bufs,
    injected perturbation/means).  Returns the ret_maps dict.  bf16_decoder (not in the reference): True = the StyledConv
    GEMM operands rounded to bf16; "storage" = additionally the pre-FIR result of every up-sampling conv rounded to bf16."""
    D = cfg["renderer_cfg"]["N_layers_renderer"]
    if style_render is None or style_decoder is None:
        style_render = mapping_renderer(sd, cfg, zs[0], truncation, style_render_mean)
        style_decoder = mapping_decoder(sd, cfg, zs[1], truncation, style_decoder_mean)
    B, S, N = cam_poses.shape[0], img_size, nerf_cfg["N_samples"]
    rays_o, rays_d, viewdirs = rays_in_world(focals, S, cam_poses, nerf_cfg.get("static_viewdirs", False))
    if nerf_cfg.get("perturb", False) and perturb_u is None:
        raise ValueError("oracle needs the per-ray uniform injected when perturb=True")
    z = z_vals(near, far, B, S, S, N, perturb_u if nerf_cfg.get("perturb", False) else None)
    pts = ray_points(rays_o, rays_d, z)
    R = S * S
    thumb, feat, sdf, mask, xyz = renderer_forward(
        sd, "renderer", pts.reshape(B, R, N, 3), rays_d.reshape(B, R, 3), viewdirs.reshape(B, R, 3),
        z.reshape(B, R, N), near, far, style_render, D, with_sdf=cfg["renderer_cfg"].get("with_sdf", True))
    to_img = lambda t: t.transpose(1, 2).reshape(B, t.shape[-1], S, S).contiguous()
    features = to_img(feat)
    rgb = decoder_forward(sd, cfg, features, style_decoder, noise_bufs, bf16_gemm=bool(bf16_decoder),
                          bf16_store=bf16_decoder == "storage")
    mask_img = to_img(mask)
    return {
        "rgb": rgb, "thumb_rgb": to_img(thumb), "style_decoder": None, "eikonal_term": None,
        "sdf": sdf.reshape(B, S, S, N, 1) if return_sdf else None,
        "xyz": to_img(xyz) if return_xyz else None,
        "mask": mask_img[:, 0:1], "depth": mask_img[:, 1:2],
        "_features": features,
    }
