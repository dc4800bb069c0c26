"""GPU parity of the backward path: HIP backward kernels, chained by cips_3dplusplus_amd.autograd,
vs torch autograd through the CPU oracle on the same seeded inputs, and vs gradients of the imported reference
(tests/golden/backward.npz).  Gradient tolerance: 2e-4 of the gradient's own max-abs (+1e-6), fp32 accumulation order
differs (fp32 atomics in the reductions)."""
import math

import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import autograd as AG
from cips_3dplusplus_amd import _lib, configs, hip
from oracle import path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(t):
    return t.to(DEV).contiguous()


def close(a, b, rel=2e-4, what=""):
    a = a.detach().cpu().reshape(b.shape)
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= rel * scale + 1e-6, f"{what}: err {err:.3e} vs scale {scale:.3e}"


def leaf(t):
    return t.clone().requires_grad_(True)


def test_linear_bwd():
    g = torch.Generator().manual_seed(1)
    for (B, i, o, lrelu) in ((2, 64, 48, False), (3, 512, 256, True), (1, 32, 512, False)):
        x, W, b, dy = torch.randn(B, i, generator=g), torch.randn(o, i, generator=g), torch.randn(o, generator=g), \
            torch.randn(B, o, generator=g)
        ws, bs, gain, osc, osh = 0.37, 0.01, 2 ** 0.5, 15.0, 30.0
        if lrelu:
            osc, osh = 1.0, 0.0
        xr, Wr, br = leaf(x), leaf(W), leaf(b)
        pre = xr @ (Wr * ws).t() + br * bs
        y = (torch.nn.functional.leaky_relu(pre, 0.2) * gain if lrelu else pre) * osc + osh
        y.backward(dy)
        xg, Wg, bg = leaf(cu(x)), leaf(cu(W)), leaf(cu(b))
        yg = AG.linear(xg, Wg, bg, w_scale=ws, b_scale=bs, lrelu=lrelu, act_gain=gain, out_scale=osc, out_shift=osh)
        close(yg, y.detach(), 1e-5, "y")
        yg.backward(cu(dy))
        close(xg.grad, xr.grad, what="dx"); close(Wg.grad, Wr.grad, what="dW"); close(bg.grad, br.grad, what="db")


@pytest.mark.parametrize("in_dim,B", [(512, 2), (36, 3), (256, 1)])
def test_linear_table_bwd(in_dim, B):
    """cips3d_linear_table_bwd: heads of different heights over shared input slots, against torch autograd."""
    g = torch.Generator().manual_seed(in_dim)
    n_slots, outs, slots = 4, (48, 3, 130, 64, 7), (0, 1, 1, 3, 0)          # two heads share slot 1, two share slot 0
    x = torch.randn(B, n_slots, in_dim, generator=g)
    Ws = [torch.randn(o, in_dim, generator=g) for o in outs]
    bs = [torch.randn(o, generator=g) for o in outs]
    dys = [torch.randn(B, o, generator=g) for o in outs]
    ws, bsc, osc, osh = 0.21, 0.5, 3.0, 1.5
    xr = leaf(x)
    Wr, br = [leaf(W) for W in Ws], [leaf(b) for b in bs]
    tot = sum(((xr[:, s] @ (W * ws).t() + b * bsc) * osc + osh) .mul(dy).sum() for W, b, s, dy in zip(Wr, br, slots, dys))
    tot.backward()
    xd = cu(x)
    out = torch.empty(B * sum(outs), device=DEV)
    tab = hip.LinearTable(DEV)
    keep, off = [], 0
    for W, b, s_, o in zip(Ws, bs, slots, outs):
        Wd, bd = cu(W), cu(b)
        keep += [Wd, bd]
        tab.add(Wd, bd, xd, n_slots * in_dim, out, o, w_scale=ws, b_scale=bsc, out_scale=osc, out_shift=osh,
                x_offset=s_ * in_dim, out_offset=off)
        off += B * o
    tab.run(B)
    off = 0
    for W, b, s_, o in zip(Ws, bs, slots, outs):
        ref = (x[:, s_] @ (W * ws).t() + b * bsc) * osc + osh
        close(out[off:off + B * o].view(B, o), ref, 1e-5, "forward")
        off += B * o
    dy = torch.cat([cu(d).reshape(-1) for d in dys])
    dx = torch.zeros_like(xd)
    dW, woffs, db = tab.backward(B, out, dy, xd, dx)
    close(dx, xr.grad, what="dx")
    row = 0
    for i, o in enumerate(outs):
        close(dW[woffs[i]:woffs[i] + o * in_dim].view(o, in_dim), Wr[i].grad, what=f"dW{i}")
        close(db[row:row + o], br[i].grad, what=f"db{i}")
        row += o


@pytest.mark.parametrize("cin,cout,hw,up,B,per_sample_noise", [
    (32, 32, 16, False, 2, False), (64, 32, 16, True, 1, False), (128, 64, 8, True, 2, True), (96, 160, 6, False, 2, False),
    (256, 512, 12, False, 1, False)])
def test_styled_conv_bwd_vs_oracle(cin, cout, hw, up, B, per_sample_noise):
    import cips_3dplusplus_amd.decoder as dec
    torch.manual_seed(cin + cout + hw)
    sc = dec.StyledConv(cin, cout, 1, 64, upsample=up)
    sc.noise.weight.data.fill_(0.3)
    sc.activate.bias.data = torch.randn(cout) * 0.2
    sd = {"m." + k: leaf(v) if v.is_floating_point() and "kernel" not in k else v.clone() for k, v in sc.state_dict().items()}
    x, st = torch.randn(B, cin, hw, hw), torch.randn(B, 64)
    ho = 2 * hw if up else hw
    nz = torch.randn(B if per_sample_noise else 1, 1, ho, ho)
    dy = torch.randn(B, cout, ho, ho)
    xr, sr, nr = leaf(x), leaf(st), leaf(nz)
    ref = O.styled_conv(sd, "m", xr, sr, nr, upsample=up)
    ref.backward(dy)
    sc = sc.to(DEV)
    xg, sg, ng = leaf(cu(x)), leaf(cu(st)), leaf(cu(nz))
    y = AG.styled_conv(sc, xg, sg, ng)
    close(y, ref.detach(), 3e-5, "y")
    y.backward(cu(dy))
    close(xg.grad, xr.grad, what="dx"); close(sg.grad, sr.grad, what="dstyle"); close(ng.grad, nr.grad, what="dnoise")
    for name, p in sc.named_parameters():
        if name == "bias":            # present in checkpoints, unused in forward (model_v3.py:440)
            assert p.grad is None
            continue
        close(p.grad, sd["m." + name].grad, what=name)


@pytest.mark.parametrize("cin,cout,hw,up,B", [(32, 64, 8, False, 2), (64, 32, 8, True, 2), (32, 32, 12, True, 1), (96, 64, 6, False, 3)])
def test_styled_conv_3x3_bwd_vs_oracle(cin, cout, hw, up, B):
    """kernel_size = 3 under autograd (it raised before): the plain branch and the stride-2 transposed + blur branch of
    ModulatedConv2d (models/model_v3.py:280-312) with noise, bias and activation, every gradient against torch autograd through
    the CPU oracle."""
    import cips_3dplusplus_amd.decoder as dec
    torch.manual_seed(cin + cout + hw + up)
    sc = dec.StyledConv(cin, cout, 3, 64, upsample=up)
    sc.noise.weight.data.fill_(0.3)
    sc.activate.bias.data = torch.randn(cout) * 0.2
    sd = {"m." + k: leaf(v) if v.is_floating_point() and "kernel" not in k else v.clone() for k, v in sc.state_dict().items()}
    x, st = torch.randn(B, cin, hw, hw), torch.randn(B, 64)
    ho = 2 * hw if up else hw
    nz = torch.randn(1, 1, ho, ho)
    dy = torch.randn(B, cout, ho, ho)
    xr, sr, nr = leaf(x), leaf(st), leaf(nz)
    ref = O.styled_conv(sd, "m", xr, sr, nr, upsample=up)
    ref.backward(dy)
    sc = sc.to(DEV)
    xg, sg, ng = leaf(cu(x)), leaf(cu(st)), leaf(cu(nz))
    y = AG.styled_conv(sc, xg, sg, ng)
    close(y, ref.detach(), 3e-5, "y")
    y.backward(cu(dy))
    close(xg.grad, xr.grad, what="dx"); close(sg.grad, sr.grad, what="dstyle"); close(ng.grad, nr.grad, what="dnoise")
    for name, p in sc.named_parameters():
        if name == "bias":
            assert p.grad is None
            continue
        close(p.grad, sd["m." + name].grad, what=name)


def test_decoder_3x3_bwd_vs_oracle():
    """Decoder.forward with decoder_cfg.kernel_size = 3 under autograd: image and the gradients of features, styles and every
    parameter against the oracle."""
    cfg = configs.tiny_G_cfg(32, 2, 3)
    G = pkg.build_generator(cfg, DEV, seed=6)
    for sc in [G.decoder.conv1] + list(G.decoder.convs):
        sc.noise.weight.data.fill_(0.2)
    sd = {k: (leaf(v.cpu()) if v.is_floating_point() and "kernel" not in k else v.cpu().clone())
          for k, v in G.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    B, S = 2, 8
    feat = torch.randn(B, 32, S, S, generator=g)
    styles = torch.randn(B, G.decoder.n_latent, 32, generator=g)
    noise = [torch.randn(*b.shape, generator=g) for b in G.decoder.create_noise_bufs(S, "cpu")]
    fr, sr = leaf(feat), leaf(styles)
    ref = O.decoder_forward(sd, cfg, fr, sr, noise)
    tgt = torch.randn(ref.shape, generator=g)
    ((ref - tgt) ** 2).mean().backward()
    fg, sg = leaf(cu(feat)), leaf(cu(styles))
    for p in G.decoder.parameters():
        p.requires_grad_(True)
    out = AG.decoder_forward(G.decoder, fg, sg, [cu(n) for n in noise])
    close(out, ref.detach(), 1e-4, "rgb")
    ((out - cu(tgt)) ** 2).mean().backward()
    close(fg.grad, fr.grad, 5e-4, "dfeatures"); close(sg.grad, sr.grad, 5e-4, "dstyles")
    n = 0
    for name, p in G.decoder.named_parameters():
        ref_g = sd["decoder." + name].grad
        if ref_g is None:
            assert p.grad is None, name
            continue
        close(p.grad, ref_g, 5e-4, name)
        n += 1
    assert n >= 30


@pytest.mark.parametrize("up", [False, True])
def test_to_rgb_bwd_vs_oracle(up):
    import cips_3dplusplus_amd.decoder as dec
    torch.manual_seed(3 + up)
    C, hw, B = 64, 16, 2
    tr = dec.ToRGB(C, 32, upsample=up)
    tr.bias.data = torch.randn(1, 3, 1, 1)
    sd = {"m." + k: leaf(v) if "kernel" not in k else v.clone() for k, v in tr.state_dict().items()}
    x, st = torch.randn(B, C, hw, hw), torch.randn(B, 32)
    skip = torch.randn(B, 3, hw // 2 if up else hw, hw // 2 if up else hw)
    dy = torch.randn(B, 3, hw, hw)
    xr, sr, kr = leaf(x), leaf(st), leaf(skip)
    ref = O.to_rgb(sd, "m", xr, sr, kr, upsample=up)
    ref.backward(dy)
    tr = tr.to(DEV)
    xg, sg, kg = leaf(cu(x)), leaf(cu(st)), leaf(cu(skip))
    y = AG.to_rgb(tr, xg, sg, kg)
    close(y, ref.detach(), 3e-5, "rgb")
    y.backward(cu(dy))
    close(xg.grad, xr.grad, what="dx"); close(sg.grad, sr.grad, what="dstyle"); close(kg.grad, kr.grad, what="dskip")
    for name, p in tr.named_parameters():
        close(p.grad, sd["m." + name].grad, what=name)


def test_decoder_bwd_vs_oracle():
    cfg = configs.tiny_G_cfg(32, 2, 1)
    G = pkg.build_generator(cfg, DEV, seed=5)
    for sc in [G.decoder.conv1] + list(G.decoder.convs):
        sc.noise.weight.data.fill_(0.2)
    sd = {k: (leaf(v.cpu()) if v.is_floating_point() and "kernel" not in k else v.cpu().clone())
          for k, v in G.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    B, S = 2, 8
    feat = torch.randn(B, 32, S, S, generator=g)
    styles = torch.randn(B, G.decoder.n_latent, 32, generator=g)
    noise = [torch.randn(*b.shape, generator=g) for b in G.decoder.create_noise_bufs(S, "cpu")]
    fr, sr = leaf(feat), leaf(styles)
    nr = [leaf(n) for n in noise]
    ref = O.decoder_forward(sd, cfg, fr, sr, nr)
    tgt = torch.randn(ref.shape, generator=g)
    ((ref - tgt) ** 2).mean().backward()
    fg, sg = leaf(cu(feat)), leaf(cu(styles))
    ng = [leaf(cu(n)) for n in noise]
    for p in G.decoder.parameters():
        p.requires_grad_(True)
    out = AG.decoder_forward(G.decoder, fg, sg, ng)
    close(out, ref.detach(), 1e-4, "rgb")
    ((out - cu(tgt)) ** 2).mean().backward()
    close(fg.grad, fr.grad, what="dfeatures"); close(sg.grad, sr.grad, what="dstyles")
    for i in range(len(ng)):
        close(ng[i].grad, nr[i].grad, what=f"dnoise{i}")
    checked = 0
    for name, p in G.decoder.named_parameters():
        r = sd["decoder." + name].grad
        if r is None:
            assert p.grad is None, name
            continue
        close(p.grad, r, what=name)
        checked += 1
    assert checked > 30


# ------------------------------------------------------------------------------------------ camera, NeRF, generator
def test_camera_bwd_vs_oracle():
    from cips_3dplusplus_amd.camera import Camera
    locs = torch.tensor([[0.3, 0.1], [-0.7, -0.2], [0.0, 0.0], [2.5, 0.6]])
    g = torch.Generator().manual_seed(4)
    dext = torch.randn(4, 3, 4, generator=g)
    lr = leaf(locs)
    O.camera_params(lr, 64, 6, 0.12)[0].backward(dext)
    lg = leaf(cu(locs))
    e = Camera.generate_camera_params(64, DEV, locations=lg, fov_ang=6, dist_radius=0.12)
    assert e[0].requires_grad and not e[1].requires_grad
    e[0].backward(cu(dext))
    close(lg.grad, lr.grad, 1e-4, "dlocations")


def _renderer_sd(G):
    return {k: v.detach().cpu() for k, v in G.state_dict().items()}


# (N = 36: more samples than the one-sample-per-thread compositing kernel takes -- the per-ray loop of composite_kernel)
@pytest.mark.parametrize("D,static,perturb,N,S", [(2, False, False, 6, 8), (3, True, True, 5, 8), (2, False, True, 8, 16),
                                                 (2, False, True, 36, 8)])
def test_nerf_render_bwd_vs_oracle(D, static, perturb, N, S):
    cfg = configs.tiny_G_cfg(32, D, 1)
    G = pkg.build_generator(cfg, DEV, seed=3)
    sd = _renderer_sd(G)
    g = torch.Generator().manual_seed(D + N)
    B, R, H = 2, S * S, 32
    locs = torch.tensor([[0.25, 0.1], [-0.4, -0.05]])
    cam = O.camera_params(locs, S, 6, 0.12)
    styles = 0.5 * torch.randn(B, D + 1, 32, generator=g)
    u = torch.rand(B, S, S, 1, generator=g) if perturb else None
    tF, tT = torch.randn(B, H, S, S, generator=g), torch.randn(B, 3, S, S, generator=g)
    # oracle
    cr, sr = leaf(cam[0]), leaf(styles)
    rays_o, rays_d, viewdirs = O.rays_in_world(cam[1], S, cr, static)
    z = O.z_vals(cam[2], cam[3], B, S, S, N, u)
    pts = O.ray_points(rays_o, rays_d, z)
    thumb, feat, sdf, mask, xyz = O.renderer_forward(sd, "renderer", pts.reshape(B, R, N, 3), rays_d.reshape(B, R, 3),
                                                     viewdirs.reshape(B, R, 3), z.reshape(B, R, N), cam[2], cam[3], sr, D)
    to_img = lambda t: t.transpose(1, 2).reshape(B, t.shape[-1], S, S)
    loss = (to_img(feat) * tF).sum() + 3.0 * (to_img(thumb) * tT).sum()
    loss.backward()
    # HIP
    cg, sg = leaf(cu(cam[0])), leaf(cu(styles))
    film = AG.film_table(G.renderer, sg)
    f_g, t_g, xyz_g, mask_g = AG.NerfRenderFn.apply(G.renderer, cg, cu(cam[1]), cu(cam[2]), cu(cam[3]), film,
                                                    None if u is None else cu(u), S, N, static)
    close(f_g, to_img(feat).detach(), 1e-4, "features"); close(t_g, to_img(thumb).detach(), 1e-4, "thumb")
    ((f_g * cu(tF)).sum() + 3.0 * (t_g * cu(tT)).sum()).backward()
    close(sg.grad, sr.grad, 3e-4, "dstyles")
    close(cg.grad, cr.grad, 3e-4, "dcam_poses")


@pytest.mark.parametrize("D,static,perturb,N,S", [(2, False, False, 8, 8), (3, True, True, 8, 12)])
def test_renderer_weight_gradients_vs_oracle(D, static, perturb, N, S):
    """`optim_render_params` (projector_v10.py:848-872, 968): gradients of EVERY renderer parameter -- layer weights and biases,
    the view layer's direction columns, the rgb / sigma heads, sigmoid_beta (NerfRenderFn, materialised backward) and the
    gamma / beta heads (film_table) -- against torch autograd through the CPU oracle on the same inputs."""
    cfg = configs.tiny_G_cfg(32, D, 1)
    G = pkg.build_generator(cfg, DEV, seed=4)
    g = torch.Generator().manual_seed(D + N)
    with torch.no_grad():
        for p in G.renderer.parameters():                  # a zero bias hides a wrong bias gradient
            if p.abs().max() == 0:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    sd = {k: leaf(v.detach().cpu()) if k.startswith("renderer.") else v.detach().cpu() for k, v in G.state_dict().items()}
    B, R, H = 2, S * S, 32
    locs = torch.tensor([[0.25, 0.1], [-0.4, -0.05]])
    cam = O.camera_params(locs, S, 6, 0.12)
    styles = 0.5 * torch.randn(B, D + 1, 32, generator=g)
    u = torch.rand(B, S, S, 1, generator=g) if perturb else None
    tF, tT = torch.randn(B, H, S, S, generator=g), torch.randn(B, 3, S, S, generator=g)
    rays_o, rays_d, viewdirs = O.rays_in_world(cam[1], S, cam[0], static)
    z = O.z_vals(cam[2], cam[3], B, S, S, N, u)
    pts = O.ray_points(rays_o, rays_d, z)
    thumb, feat, sdf, mask, xyz = O.renderer_forward(sd, "renderer", pts.reshape(B, R, N, 3), rays_d.reshape(B, R, 3),
                                                     viewdirs.reshape(B, R, 3), z.reshape(B, R, N), cam[2], cam[3], styles, D)
    to_img = lambda t: t.transpose(1, 2).reshape(B, t.shape[-1], S, S)
    ((to_img(feat) * tF).sum() + 3.0 * (to_img(thumb) * tT).sum()).backward()
    # HIP
    G.requires_grad_(False)
    G.renderer.requires_grad_(True)
    film = AG.film_table(G.renderer, cu(styles))
    rp = [p for _, p in AG.nerf_named_parameters(G.renderer)]
    f_g, t_g, _, _ = AG.NerfRenderFn.apply(G.renderer, cu(cam[0]), cu(cam[1]), cu(cam[2]), cu(cam[3]), film,
                                           None if u is None else cu(u), S, N, static, *rp)
    close(f_g, to_img(feat).detach(), 1e-4, "features")
    ((f_g * cu(tF)).sum() + 3.0 * (t_g * cu(tT)).sum()).backward()
    n = 0
    for name, p in G.renderer.named_parameters():
        ref = sd["renderer." + name].grad
        assert ref is not None and p.grad is not None, name
        close(p.grad, ref, 5e-4, name)
        n += 1
    assert n == 5 + 6 * (D + 1)


@pytest.mark.parametrize("hidden,S,N", [(32, 8, 6), (64, 16, 8)])
def test_raw_density_renderer_bwd_vs_oracle(hidden, S, N):
    """renderer_cfg.with_sdf = False (sigma = softplus of the head's output, cips3d/nerf_utils.py:288-297) under autograd:
    camera / style gradients on the route NerfRenderFn picks for the shape (fused where it tiles), and every renderer
    parameter's gradient on the materialised route, against torch autograd through the CPU oracle."""
    D = 2
    cfg = configs.tiny_G_cfg(hidden, D, 1)
    cfg["renderer_cfg"] = dict(cfg["renderer_cfg"], with_sdf=False)
    G = pkg.build_generator(cfg, DEV, seed=6)
    assert G.renderer.with_sdf is False
    g = torch.Generator().manual_seed(hidden + N)
    with torch.no_grad():
        for q in G.renderer.parameters():
            if q.abs().max() == 0:
                q.copy_(0.05 * torch.randn(q.shape, generator=g))
    sd = {k: leaf(v.detach().cpu()) if k.startswith("renderer.") else v.detach().cpu() for k, v in G.state_dict().items()}
    B, R, H = 2, S * S, hidden
    locs = torch.tensor([[0.25, 0.1], [-0.4, -0.05]])
    cam = O.camera_params(locs, S, 6, 0.12)
    sdim = G.renderer.style_dim
    styles = 0.5 * torch.randn(B, D + 1, sdim, generator=g)
    u = torch.rand(B, S, S, 1, generator=g)
    tF, tT = torch.randn(B, H, S, S, generator=g), torch.randn(B, 3, S, S, generator=g)
    cr, sr = leaf(cam[0]), leaf(styles)
    rays_o, rays_d, viewdirs = O.rays_in_world(cam[1], S, cr, False)
    z = O.z_vals(cam[2], cam[3], B, S, S, N, u)
    pts = O.ray_points(rays_o, rays_d, z)
    thumb, feat, sdf, mask, xyz = O.renderer_forward(sd, "renderer", pts.reshape(B, R, N, 3), rays_d.reshape(B, R, 3),
                                                     viewdirs.reshape(B, R, 3), z.reshape(B, R, N), cam[2], cam[3], sr, D,
                                                     with_sdf=False)
    to_img = lambda t: t.transpose(1, 2).reshape(B, t.shape[-1], S, S)
    ((to_img(feat) * tF).sum() + 3.0 * (to_img(thumb) * tT).sum()).backward()
    # camera / styles: the route of the shape
    G.requires_grad_(False)
    cg, sg = leaf(cu(cam[0])), leaf(cu(styles))
    film = AG.film_table(G.renderer, sg)
    f_g, t_g, _, _ = AG.NerfRenderFn.apply(G.renderer, cg, cu(cam[1]), cu(cam[2]), cu(cam[3]), film, cu(u), S, N, False)
    close(f_g, to_img(feat).detach(), 1e-4, "features"); close(t_g, to_img(thumb).detach(), 1e-4, "thumb")
    ((f_g * cu(tF)).sum() + 3.0 * (t_g * cu(tT)).sum()).backward()
    close(sg.grad, sr.grad, 3e-4, "dstyles")
    close(cg.grad, cr.grad, 3e-4, "dcam_poses")
    # the renderer's own parameters: materialised route
    G.renderer.requires_grad_(True)
    film = AG.film_table(G.renderer, cu(styles))
    rp = [q for _, q in AG.nerf_named_parameters(G.renderer)]
    f_g, t_g, _, _ = AG.NerfRenderFn.apply(G.renderer, cu(cam[0]), cu(cam[1]), cu(cam[2]), cu(cam[3]), film, cu(u), S, N, False, *rp)
    ((f_g * cu(tF)).sum() + 3.0 * (t_g * cu(tT)).sum()).backward()
    n = 0
    for name, q in G.renderer.named_parameters():
        ref = sd["renderer." + name].grad
        if name == "sigmoid_beta":              # unused on this branch
            assert ref is None and (q.grad is None or float(q.grad.abs().max()) == 0.0)
            continue
        assert ref is not None and q.grad is not None, name
        close(q.grad, ref, 5e-4, name)
        n += 1
    assert n == 4 + 6 * (D + 1)


def test_generator_optimises_renderer_parameters():
    """Generator.forward accepts renderer parameters that require gradients (it raised before) and the projector's
    `optim_render_params` switch moves them."""
    from cips_3dplusplus_amd.projector import FlipProjector, surrogate_loss
    G = pkg.build_generator(configs.tiny_G_cfg(32, 2, 1), DEV, seed=2)
    g = torch.Generator(device=DEV).manual_seed(0)
    t_rgb = torch.randn(2, 3, 32, 32, device=DEV, generator=g).clamp(-1, 1)
    t_thumb = torch.randn(2, 3, 8, 8, device=DEV, generator=g).clamp(-1, 1)
    before = {k: v.detach().clone() for k, v in G.renderer.state_dict().items()}
    out = FlipProjector(G, DEV).project_wplus({"img_size": 8, "fov_ang": 6, "dist_radius": 0.12},
                                             {"N_samples": 6, "perturb": False, "static_viewdirs": True},
                                             surrogate_loss(t_rgb, t_thumb), N_steps_pose=6, N_steps_app=0, w_avg_samples=64,
                                             optim_render_params=True)
    moved = [k for k, v in out["render_state_dict"].items() if not torch.equal(v, before[k])]
    assert len(moved) == len(before), sorted(set(before) - set(moved))
    assert all(bool(torch.isfinite(v).all()) for v in out["render_state_dict"].values())


def test_optimised_renderer_parameters_are_the_ones_the_forward_uses():
    """HipAdam writes parameters through raw pointers; the renderer's packed hidden weights and stacked biases are cached on
    (data_ptr, _version).  Without a version bump every step after the first rendered with the INITIAL hidden weights and
    biases while the backward recomputed with live ones.  (a) every step must change the cache key and the cached buffers must
    equal a fresh packing of the live weights; (b) after K steps of the projector (`optim_render_params`) the optimised module
    must render exactly what a fresh generator loaded from its state_dict renders.  (A loss-trajectory comparison with
    torch.optim.Adam is not a test: Adam's first steps are +-lr by the SIGN of each gradient, and the backward's fp32 atomics
    reorder sums, so entries with noise-level gradients legitimately take different signs from run to run.)"""
    from cips_3dplusplus_amd.camera import Camera
    from cips_3dplusplus_amd.optim import HipAdam
    from cips_3dplusplus_amd.projector import FlipProjector, surrogate_loss
    g = torch.Generator(device=DEV).manual_seed(0)
    # (a) the optimiser alone
    G = pkg.build_generator(configs.tiny_G_cfg(32, 2, 1), DEV, seed=2)
    ren = G.renderer
    ren.requires_grad_(True)
    opt = HipAdam(list(ren.parameters()), lr=1e-2)
    packed0, bias0 = (t.clone() for t in ren._derived_buffers())
    key0 = ren._weights_key()
    for step in range(3):
        for p in ren.parameters():
            p.grad = torch.randn(p.shape, device=DEV, generator=g)
        key_before = ren._weights_key()
        opt.step()
        assert ren._weights_key() != key_before, "a HipAdam step must bump the parameters' versions"
        packed, bias = ren._derived_buffers()
        net = ren.network
        fresh_bias = torch.stack([l.bias for l in net.pts_linears] + [net.views_linears.bias]).detach()
        fresh_packed = hip.nerf_pack_weights(torch.stack([l.weight for l in net.pts_linears[1:]]).detach().contiguous(),
                                             net.views_linears.weight.detach().contiguous(), ren.hidden_dim, ren.N_layers_renderer)
        # (bit patterns: the packed stream holds fp16 pairs; its tail beyond the per-layer scales is never written)
        n_used = ren.N_layers_renderer * ren.hidden_dim ** 2 + 2 * ren.N_layers_renderer
        bits = lambda t: t.reshape(-1)[:n_used].view(torch.int32)      # noqa: E731
        assert torch.equal(bias, fresh_bias) and torch.equal(bits(packed), bits(fresh_packed)), step
    assert ren._weights_key() != key0 and not torch.equal(bits(packed), bits(packed0)) and not torch.equal(bias, bias0)
    # (b) through the projector
    t_rgb = torch.randn(2, 3, 32, 32, device=DEV, generator=g).clamp(-1, 1)
    t_thumb = torch.randn(2, 3, 8, 8, device=DEV, generator=g).clamp(-1, 1)
    G = pkg.build_generator(configs.tiny_G_cfg(32, 2, 1), DEV, seed=2)
    out = FlipProjector(G, DEV).project_wplus({"img_size": 8, "fov_ang": 6, "dist_radius": 0.12},
                                             {"N_samples": 6, "perturb": False, "static_viewdirs": True},
                                             surrogate_loss(t_rgb, t_thumb), N_steps_pose=8, N_steps_app=0, w_avg_samples=64,
                                             optim_render_params=True)
    Gopt = out["G"]
    hist = out["loss_history"]
    assert bool(torch.isfinite(hist).all()) and float(hist[-1]) < float(hist[0])
    fresh = pkg.build_generator(configs.tiny_G_cfg(32, 2, 1), DEV, state_dict={k: v.detach().clone() for k, v in Gopt.state_dict().items()})
    e, f, n, fa, _ = Camera.generate_camera_params(8, DEV, locations=torch.tensor([[0.2, -0.1]], device=DEV), fov_ang=6, dist_radius=0.12)
    styles = torch.randn(1, 3, Gopt.renderer.style_dim, device=DEV, generator=g)
    with torch.no_grad():
        a = Gopt.renderer.render(e, f, n, fa, styles, 8, 6)
        b = fresh.renderer.render(e, f, n, fa, styles, 8, 6)
    for x, y in zip(a, b):
        if x is not None:
            assert torch.equal(x, y)


@pytest.mark.parametrize("tag,D,static", [("h32_d2", 2, True), ("h32_d3", 3, False)])
def test_generator_backward_golden(golden, tag, D, static):
    """One inversion-like step: loss and every gradient vs the imported reference (tests/golden/backward.npz)."""
    from cips_3dplusplus_amd.camera import Camera
    fx, tiny = golden("backward"), golden("tiny_generator")
    cfg = configs.tiny_G_cfg(32, D, 1)
    G = pkg.build_generator(cfg, DEV, state_dict=tiny.sub(f"{tag}.sd."))
    G.requires_grad_(False)
    G.decoder.requires_grad_(True)
    locs, w_r, w_d = leaf(cu(fx[f"{tag}.locs"])), leaf(cu(fx[f"{tag}.w_r"])), leaf(cu(fx[f"{tag}.w_d"]))
    nb = [leaf(cu(fx[f"{tag}.noise{i}"])) for i in range(G.decoder.num_layers)]
    e, f, n, fa, _ = Camera.generate_camera_params(8, DEV, locations=locs, fov_ang=6, dist_radius=0.12)
    r = G(zs=[None, None], style_render=w_r, style_decoder=w_d, cam_poses=e, focals=f, img_size=8, near=n, far=fa,
          noise_bufs=nb, nerf_cfg=dict(N_samples=6, perturb=False, static_viewdirs=static), renderer_detach=False)
    close(r["rgb"], fx[f"{tag}.rgb"], 1e-4, "rgb"); close(r["thumb_rgb"], fx[f"{tag}.thumb"], 1e-4, "thumb")
    loss = ((r["rgb"] - cu(fx[f"{tag}.t_rgb"])) ** 2).mean() + 50 * ((r["thumb_rgb"] - cu(fx[f"{tag}.t_thumb"])) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(fx[f"{tag}.loss"])) < 1e-4 * float(fx[f"{tag}.loss"])
    close(locs.grad, fx[f"{tag}.g.locs"], 5e-4, "dlocs"); close(w_r.grad, fx[f"{tag}.g.w_r"], 5e-4, "dw_render")
    close(w_d.grad, fx[f"{tag}.g.w_d"], 5e-4, "dw_decoder")
    for i in range(len(nb)):
        close(nb[i].grad, fx[f"{tag}.g.noise{i}"], 5e-4, f"dnoise{i}")
    n_checked = 0
    for name, p in G.decoder.named_parameters():
        key = f"{tag}.g.dec.{name}"
        if key in fx:
            close(p.grad, fx[key], 5e-4, name)
            n_checked += 1
        else:
            assert p.grad is None, name
    assert n_checked > 30
    assert all(p.grad is None for p in G.renderer.parameters())


def test_path_selection_by_grad_requirements():
    """Plain inference (frozen handle, or torch.no_grad()) runs the fused no-graph path; decoder parameters that require
    grad select the differentiable path even when no input does; so do renderer parameters; mapping-network parameters raise."""
    from cips_3dplusplus_amd.camera import Camera
    G = pkg.build_generator(configs.tiny_G_cfg(32, 2, 1), DEV, seed=1)
    assert not any(p.requires_grad for p in G.parameters())
    e, f, n, fa, _ = Camera.generate_camera_params(8, DEV, locations=torch.zeros(1, 2, device=DEV))
    zs = [torch.randn(1, 32, device=DEV), torch.randn(1, 32, device=DEV)]
    kw = dict(zs=zs, cam_poses=e, focals=f, img_size=8, near=n, far=fa, nerf_cfg=dict(N_samples=6, perturb=False),
              noise_bufs=G.create_noise_bufs(8, DEV))
    r = G(**kw)
    assert not r["rgb"].requires_grad
    G.decoder.requires_grad_(True)                      # flip steps / optim_cam=False: only decoder parameters are leaves
    with torch.no_grad():
        r0 = G(**kw)
    assert not r0["rgb"].requires_grad
    r1 = G(**kw)
    assert r1["rgb"].requires_grad
    close(r1["rgb"], r["rgb"].cpu(), 1e-5, "rgb on the two paths")
    r1["rgb"].square().mean().backward()
    assert G.decoder.conv1.conv.weight.grad is not None and float(G.decoder.conv1.conv.weight.grad.abs().max()) > 0
    G.renderer.requires_grad_(True)                     # optim_render_params: the renderer's weights get gradients too
    r2 = G(**kw)
    (r2["rgb"].square().mean() + r2["thumb_rgb"].square().mean()).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in G.renderer.parameters())
    assert float(G.renderer.network.pts_linears[0].weight.grad.abs().max()) > 0
    G.style.requires_grad_(True)                        # the mapping networks stay constants
    with pytest.raises(NotImplementedError, match="mapping-network"):
        G(**kw)
    with torch.no_grad():
        G(**kw)                                          # inference is unaffected


def test_exact_fp32_mode_is_inference_only():
    """set_precision("fp32_exact") covers the whole inference forward; a differentiable forward in that mode must raise instead of
    silently running the split-fp16 stash kernel (gradients against another forward than inference runs)."""
    from cips_3dplusplus_amd.camera import Camera
    G = pkg.build_generator(configs.ffhq_G_cfg(256, 2), DEV, seed=1)
    e, f, n, fa, _ = Camera.generate_camera_params(16, DEV, locations=torch.zeros(1, 2, device=DEV))
    zs = [torch.randn(1, 256, device=DEV), torch.randn(1, 256, device=DEV)]
    kw = dict(zs=zs, cam_poses=e, focals=f, img_size=16, near=n, far=fa, nerf_cfg=dict(N_samples=8, perturb=False),
              noise_bufs=G.create_noise_bufs(16, DEV))
    with torch.no_grad():
        r = G(**kw)["rgb"].cpu()
    G.set_precision("fp32_exact")
    with torch.no_grad():
        close(G(**kw)["rgb"], r, 1e-4, "exact-fp32 inference next to the split forward")
    G.decoder.requires_grad_(True)
    with pytest.raises(NotImplementedError, match="inference-only"):
        G(**kw)
    G.set_precision("fp32")
    assert G(**kw)["rgb"].requires_grad


def test_flip_inversion_loop_reduces_loss():
    """The optimisation loop of projector_v10.py:915-1280 over the HIP forward/backward: a target rendered at a known
    pose / style is approached (pose phase: camera + NeRF style, decoder frozen; appearance phase: decoder too)."""
    import copy
    from cips_3dplusplus_amd.camera import Camera
    from cips_3dplusplus_amd.projector import FlipProjector, surrogate_loss, cur_lr
    assert cur_lr(0, 200) == 0.0 and abs(cur_lr(100, 200) - 1.0) < 1e-12 and cur_lr(199, 200) < 0.01
    cfg = configs.tiny_G_cfg(32, 2, 1)
    G = pkg.build_generator(cfg, DEV, seed=11)
    G2 = copy.deepcopy(G)                                  # copies must not share plan / table pointers
    cam_cfg = {"img_size": 8, "fov_ang": 6, "dist_radius": 0.12}
    ncfg = {"N_samples": 6, "perturb": False, "static_viewdirs": True}
    with torch.no_grad():
        mr, md = G.get_mean_latent(512, DEV)
        torch.manual_seed(0)
        w_r = mr.reshape(1, 1, -1).repeat(2, 3, 1) + 0.3 * torch.randn(1, 3, 32, device=DEV)
        w_d = md.reshape(1, 1, -1).repeat(2, G.decoder.n_latent, 1)
        loc = torch.tensor([[0.35, 0.1], [-0.35, 0.1]], device=DEV)
        e, f, n, fa, _ = Camera.generate_camera_params(8, DEV, locations=loc, fov_ang=6, dist_radius=0.12)
        nb0 = [torch.zeros_like(b) for b in G.create_noise_bufs(8, DEV)]
        tgt = G2(zs=[None, None], style_render=w_r, style_decoder=w_d, cam_poses=e, focals=f, img_size=8, near=n, far=fa,
                 noise_bufs=nb0, nerf_cfg=ncfg)
    proj = FlipProjector(G, DEV)
    out = proj.project_wplus(cam_cfg, ncfg, surrogate_loss(tgt["rgb"], tgt["thumb_rgb"]), N_steps_pose=60, N_steps_app=30,
                             lr_cam=0.02, lr_render_w=0.01, w_avg_samples=512, azim_init=(0.15, -0.15))
    h = out["loss_history"]
    assert float(h[-1]) < 0.35 * float(h[1]), (float(h[1]), float(h[-1]))
    assert abs(float(out["azim"][0]) - 0.35) < 0.2 and abs(float(out["azim"][1]) + 0.35) < 0.2   # from +-0.15 towards +-0.35
    assert set(out) >= {"azim", "elev", "w_render_opt", "w_decoder_opt", "render_state_dict", "decoder_state_dict",
                        "noise_bufs", "padding"}
    # the pose phase must not have touched the decoder; the appearance phase must have
    d0, d1 = G.decoder.state_dict(), out["decoder_state_dict"]
    assert any(not torch.equal(d0[k], d1[k]) for k in d0 if d0[k].is_floating_point())
    assert all(torch.equal(a, b) for a, b in zip(G.renderer.state_dict().values(), out["render_state_dict"].values()))


def test_pose_phase_trajectory_matches_the_oracle_loop():
    """Eight steps of the pose phase (projector_v10.py:915-1216: camera angles + NeRF W+ under Adam with the ramped learning rate,
    decoder at lr 0) on the HIP path -- one-call decoder node, fused NeRF backward, fused Adam -- against the same loop written
    with torch autograd over the CPU oracle and torch's default Adam: the loss and the camera angles of every step."""
    from cips_3dplusplus_amd.projector import FlipProjector, surrogate_loss, cur_lr
    cfg = configs.tiny_G_cfg(32, 2, 1)
    G = pkg.build_generator(cfg, DEV, seed=13)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    mr, md = 0.3 * torch.randn(1, 32, generator=g), 0.3 * torch.randn(1, 32, generator=g)
    t_rgb, t_thumb = torch.randn(2, 3, 32, 32, generator=g).clamp(-1, 1), torch.randn(2, 3, 8, 8, generator=g).clamp(-1, 1)
    N, lr_cam, lr_w, az0 = 8, 0.02, 0.01, (0.3, -0.2)
    ncfg = {"N_samples": 6, "perturb": False, "static_viewdirs": True}
    G.get_mean_latent = lambda n, dev: (cu(mr), cu(md))        # (the projector draws its means on the device otherwise)
    traj = []
    FlipProjector(G, DEV).project_wplus({"img_size": 8, "fov_ang": 6, "dist_radius": 0.12}, ncfg,
                                        surrogate_loss(cu(t_rgb), cu(t_thumb)), N_steps_pose=N, N_steps_app=0, lr_cam=lr_cam,
                                        lr_render_w=lr_w, azim_init=az0, w_avg_samples=8,
                                        on_step=lambda s_, l, a, e: traj.append((float(l.detach()), a.detach().cpu().clone(),
                                                                                 e.detach().cpu().clone())))
    # the same loop on the CPU oracle
    azim = torch.tensor([[az0[0]], [az0[1]]], requires_grad=True)
    elev = torch.zeros(2, 1, requires_grad=True)
    w_r = mr.reshape(1, 1, -1).repeat(1, 3, 1).clone().requires_grad_(True)
    w_d = md.reshape(1, 1, -1).repeat(2, G.decoder.n_latent, 1)
    nb = [torch.zeros(*b.shape) for b in G.create_noise_bufs(8, "cpu")]
    o_cam = torch.optim.Adam([{"params": [azim, elev], "lr": lr_cam, "betas": (0.9, 0.999)}])
    o_w = torch.optim.Adam([{"params": [w_r], "lr": lr_w, "betas": (0.9, 0.999)}])
    for step in range(N):
        m = cur_lr(step, N)
        o_cam.param_groups[0]["lr"], o_w.param_groups[0]["lr"] = lr_cam * m, lr_w * m
        cam = O.camera_params(torch.cat([azim, elev], 1), 8, 6, 0.12)
        r = O.generator_forward(sd, cfg, [None, None], cam[0], cam[1], 8, cam[2], cam[3], ncfg, nb, style_render=w_r.repeat(2, 1, 1),
                                style_decoder=w_d)
        loss = ((r["rgb"] - t_rgb) ** 2).mean() + 50.0 * ((r["thumb_rgb"] - t_thumb) ** 2).mean()
        o_cam.zero_grad(); o_w.zero_grad()
        loss.backward()
        o_cam.step(); o_w.step()
        l_hip, a_hip, e_hip = traj[step]
        assert abs(l_hip - float(loss.detach())) < 2e-4 * abs(float(loss.detach())), (step, l_hip)
        assert float((a_hip - azim.detach()).abs().max()) < 2e-4 and float((e_hip - elev.detach()).abs().max()) < 2e-4, step
    assert len(traj) == N and abs(float(azim.detach()[0]) - az0[0]) > 1e-3            # the camera moved


def test_modulate_transpose_packing_equals_pack_weights():
    """CIPS3D_MOD_TRANSPOSE: the modulate kernel writes the A fragments of wm^T directly -- bit for bit what
    cips3d_pack_weights(transpose = 1) makes from the plain modulated matrix (the data-gradient GEMM's operand)."""
    g = torch.Generator().manual_seed(3)
    B, cout, cin = 2, 64, 96
    W = cu(torch.randn(1, cout, cin, 1, 1, generator=g))
    s = cu(1.0 + 0.3 * torch.randn(B, cin, generator=g))
    for demod in (True, False):
        plain = hip.modulate_weights(W, s, cin, B, cout, cin, 1, 1.0 / math.sqrt(cin), demod, packed=False)
        ref = hip.pack_weights(plain.view(B, cout, cin), transpose=True)
        lib = _lib.load()
        out = torch.empty(B * cout * cin, device=DEV)
        flags = (hip.MOD_DEMODULATE if demod else 0) | hip.MOD_PACKED | hip.MOD_TRANSPOSE
        _lib.check(lib.cips3d_modulate_weights(W.data_ptr(), s.data_ptr(), cin, out.data_ptr(), B, cout, cin, 1,
                                               1.0 / math.sqrt(cin), flags, torch.cuda.current_stream().cuda_stream), "mod")
        assert torch.equal(out, ref.reshape(-1))
