"""FiLM-SIREN volume renderer modules (parameter containers + the fused HIP render call).

Class names, constructor arguments, parameter names/shapes and initialisation follow
/root/reference/exp/cips3d/volume_renderer.py:15-190 so that a reference `state_dict` loads
unchanged (`renderer.sigmoid_beta`, `renderer.network.pts_linears.{i}.{weight,bias,gamma.*,beta.*}`,
`views_linears.*`, `rgb_linear.*`, `sigma_linear.*`).  The arithmetic itself lives in
csrc/nerf.hip: `VolumeFeatureRenderer.render` launches linear_table (FiLM heads) -> nerf_render ->
nerf_finish; per-point activations never reach HBM.
"""
import math

import torch
from torch import nn

from . import hip


def _uniform(shape, bound):
    return torch.empty(*shape).uniform_(-bound, bound)


class LinearLayer(nn.Module):
    """volume_renderer.py:15-35: out = std_init * (x W^T + b) + bias_init (constants applied at run time)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, std_init=1, freq_init=False, is_first=False):
        super().__init__()
        if is_first:
            w = _uniform((out_dim, in_dim), 1 / in_dim)
        elif freq_init:
            w = _uniform((out_dim, in_dim), math.sqrt(6 / in_dim) / 25)
        else:
            w = 0.25 * nn.init.kaiming_normal_(torch.randn(out_dim, in_dim), a=0.2, mode="fan_in",
                                               nonlinearity="leaky_relu")
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(_uniform((out_dim,), math.sqrt(1 / in_dim)))
        self.bias_init = bias_init
        self.std_init = std_init

    def forward(self, input):
        if input.numel() // input.shape[-1] > 64:       # point tensors (b, ..., n, c): MFMA path
            return hip.points_linear(input, self.weight, self.bias, out_scale=float(self.std_init),
                                     out_shift=float(self.bias_init))
        x = input.reshape(-1, input.shape[-1]).contiguous()
        y = hip.linear(x, self.weight, self.bias, out_scale=float(self.std_init), out_shift=float(self.bias_init))
        return y.view(*input.shape[:-1], -1)


class FiLMSiren(nn.Module):
    """volume_renderer.py:39-85: sin(gamma(style) * (x W^T + b) + beta(style))."""

    def __init__(self, in_channel, out_channel, style_dim, is_first=False):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        bound = 1 / 3 if is_first else math.sqrt(6 / in_channel) / 25
        self.weight = nn.Parameter(_uniform((out_channel, in_channel), bound))
        self.bias = nn.Parameter(_uniform((out_channel,), math.sqrt(1 / in_channel)))
        self.gamma = LinearLayer(style_dim, out_channel, bias_init=30, std_init=15)
        self.beta = LinearLayer(style_dim, out_channel, bias_init=0, std_init=0.25)

    @torch.no_grad()
    def forward(self, input, style):
        """volume_renderer.py:70-85: input (b, ..., in), style (b, style_dim) -> sin(gamma * (x W^T + b) + beta).
        (The generator itself never materialises per-point activations: csrc/nerf.hip fuses all layers.)"""
        film = torch.stack([self.gamma(style), self.beta(style)], 1).contiguous()       # [B, 2, out]
        return hip.points_linear(input, self.weight, self.bias, film=film)


class SirenGenerator(nn.Module):
    """volume_renderer.py:89-116 (construction order matters for seeded init parity)."""

    def __init__(self, D=8, W=256, style_dim=256, input_ch=3, input_ch_views=3, output_features=True, **kwargs):
        super().__init__()
        self.D, self.W, self.style_dim = D, W, style_dim
        self.input_ch, self.input_ch_views, self.output_features = input_ch, input_ch_views, output_features
        self.pts_linears = nn.ModuleList([FiLMSiren(3, W, style_dim=style_dim, is_first=True)] +
                                         [FiLMSiren(W, W, style_dim=style_dim) for _ in range(D - 1)])
        self.views_linears = FiLMSiren(input_ch_views + W, W, style_dim=style_dim)
        self.rgb_linear = LinearLayer(W, 3, freq_init=True)
        self.sigma_linear = LinearLayer(W, 1, freq_init=True)

    @torch.no_grad()
    def points_forward(self, x, styles):
        """volume_renderer.py:133-160: x (b, ..., n, 3 + 3) = [normalised points, view dirs], styles (b, D+1, style_dim)
        -> rgb (.., 3), sdf (.., 1), features (.., W), all per point (layer by layer, materialised)."""
        pts, views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = pts.contiguous()
        for i, layer in enumerate(self.pts_linears):
            h = layer(h, styles[:, i])
        sdf = self.sigma_linear(h)
        feat = self.views_linears(torch.cat([h, views], -1), styles[:, -1])
        return self.rgb_linear(feat), sdf, feat

    def forward(self, x, styles, forward_points=None):
        return self.points_forward(x=x, styles=styles)


class VolumeFeatureRenderer(nn.Module):
    """volume_renderer.py:163-190 container; `render` is the fused replacement of
    prepare_nerf_inputs + forward + volume_integration (nerf_utils.py:173-338, volume_renderer.py:192-303)."""

    def __init__(self, N_layers_renderer, input_dim, hidden_dim, style_dim, view_dim, with_sdf, output_features,
                 **kwargs):
        super().__init__()
        if input_dim != 3 or view_dim != 3:
            raise NotImplementedError("the HIP renderer implements input_dim=3, view_dim=3")
        self.N_layers_renderer = N_layers_renderer
        self.input_dim, self.hidden_dim, self.style_dim, self.view_dim = input_dim, hidden_dim, style_dim, view_dim
        self.with_sdf, self.output_features = with_sdf, output_features
        self.sigmoid_beta = nn.Parameter(0.1 * torch.ones(1))
        self.network = SirenGenerator(D=N_layers_renderer, W=hidden_dim, style_dim=style_dim, input_ch=input_dim,
                                      input_ch_views=view_dim, output_features=output_features)
        self._derived = None      # (key, packed, layer_bias)
        self._packed_t = None
        self._packed32 = None     # (key, exact-fp32 weight stream): built on first use in "fp32_exact" precision
        self.exact_fp32 = False
        self._tables = {}         # B -> (styles_buf, film_buf, LinearTable)

    # ---- derived, weight-dependent device buffers (re-made when a parameter changes) ------------
    def _weights_key(self):
        # (the parameter list is cached: walking the module tree is most of what this key costs per forward)
        ent = self.__dict__.get("_wk_params")
        w0 = self.network.views_linears._parameters["weight"]
        if ent is None or ent[0] is not w0:
            net = self.network
            ps = [l.weight for l in net.pts_linears] + [l.bias for l in net.pts_linears] + \
                 [net.views_linears.weight, net.views_linears.bias]
            ent = self.__dict__["_wk_params"] = (w0, ps)
        return tuple((p.data_ptr(), p._version) for p in ent[1])

    def set_precision(self, precision):
        """"fp32" (default): the point MLP's GEMMs as fp32-equivalent split-fp16 products (three exact fp16 products per fp32
        product, fp32 accumulate: csrc/nerf.hip); "fp32_exact": the fp32 matrix instruction on fp32 operands -- IEEE fp32
        products, the arithmetic of the reference's F.linear (cips3d/volume_renderer.py:15-35, 74-85) -- through
        the F32 instantiation of the same kernel (hidden_dim 256, inference; ~2.3x the render time)."""
        if precision not in ("fp32", "fp32_exact"):
            raise ValueError(precision)
        if precision == "fp32_exact" and self.hidden_dim != 256:
            raise NotImplementedError("the exact-fp32 render kernel is built for hidden_dim = 256")
        self.exact_fp32 = precision == "fp32_exact"
        return self

    def packed32(self):
        """The exact-fp32 weight stream (None in the default precision), re-made when a weight changes."""
        if not self.exact_fp32:
            return None
        key = self._weights_key()
        if self._packed32 is None or self._packed32[0] != key:
            net = self.network
            D, H = self.N_layers_renderer, self.hidden_dim
            with torch.no_grad():
                w_hidden = torch.stack([l.weight for l in net.pts_linears[1:]]).contiguous() if D > 1 else None
                self._packed32 = (key, hip.nerf_pack_weights32(w_hidden, net.views_linears.weight.detach().contiguous(), H, D))
        return self._packed32[1]

    def _derived_buffers(self):
        key = self._weights_key()
        if self._derived is None or self._derived[0] != key:
            net = self.network
            D, H = self.N_layers_renderer, self.hidden_dim
            with torch.no_grad():
                w_hidden = torch.stack([l.weight for l in net.pts_linears[1:]]).contiguous() if D > 1 else None
                packed = hip.nerf_pack_weights(w_hidden, net.views_linears.weight.detach().contiguous(), H, D)
                layer_bias = torch.stack([l.bias for l in net.pts_linears] + [net.views_linears.bias]).contiguous()
            self._derived = (key, packed, layer_bias)
            self._packed_t = None
        return self._derived[1], self._derived[2]

    def _packed_transposed(self):
        """The transposed weight stream of the fused backward, made on first use (inversion only)."""
        packed, _ = self._derived_buffers()
        if self._packed_t is None:
            net = self.network
            D, H = self.N_layers_renderer, self.hidden_dim
            with torch.no_grad():
                w_hidden = torch.stack([l.weight for l in net.pts_linears[1:]]).contiguous() if D > 1 else None
                self._packed_t = hip.nerf_pack_weights_t(w_hidden, net.views_linears.weight.detach().contiguous(), packed, H, D)
        return self._packed_t

    def _film_table(self, B, device, lane=0):
        net = self.network
        key = (B, net.views_linears.gamma.weight.data_ptr())
        slot = B if lane == 0 else (B, lane)
        ent = self._tables.get(slot)
        if ent is None or ent[0] != key:
            D, H, S = self.N_layers_renderer, self.hidden_dim, self.style_dim
            styles_buf = torch.empty(B, D + 1, S, device=device)
            film = torch.empty(B, D + 1, 2, H, device=device)
            tab = hip.LinearTable(device, lane)
            layers = list(net.pts_linears) + [net.views_linears]
            for l, layer in enumerate(layers):
                for j, head in enumerate((layer.gamma, layer.beta)):
                    tab.add(head.weight, head.bias, styles_buf, (D + 1) * S, film, (D + 1) * 2 * H,
                            out_scale=float(head.std_init), out_shift=float(head.bias_init),
                            x_offset=l * S, out_offset=(l * 2 + j) * H)
            ent = (key, styles_buf, film, tab)
            self._tables[slot] = ent
        return ent[1], ent[2], ent[3]

    @torch.no_grad()
    def render(self, cam_poses, focals, near, far, styles, img_size, N_samples, perturb_u=None,
               static_viewdirs=False, return_sdf=False, n_chunks=None, film=None, stash=None, zero_words=None, planar_mask=False):
        """cam_poses (B,3,4), focals/near/far (B,1,1), styles (B,D+1,style_dim)
        -> thumb_rgb (B,3,S,S), features (B,H,S,S), sdf (B,S,S,N,1)|None, mask (B,2,S,S), xyz (B,3,S,S)
        zero_words: a float32 scratch tensor the launch leaves zeroed (cips3d_nerf_params.zero_words).
        planar_mask: mask comes back as (2,B,S,S) -- the layout Generator.forward splits into its `mask` and `depth` maps."""
        B = cam_poses.shape[0]
        dev = cam_poses.device
        D, H = self.N_layers_renderer, self.hidden_dim
        net = self.network
        packed, layer_bias = self._derived_buffers()
        if film is None:
            styles_buf, film, tab = self._film_table(B, dev)
            styles_buf.copy_(styles)
            tab.run(B)
        else:                       # FiLM table computed by the caller (differentiable path: autograd.film_table)
            film = film.detach().float().contiguous()
        if stash is not None:       # differentiable forward: hip.nerf_forward_stash buffers, filled for the fused backward
            if self.exact_fp32:
                # the stash instantiation and the fused backward exist in split-fp16 arithmetic only: gradients would be taken
                # against another forward than the one inference runs in this mode
                raise NotImplementedError('set_precision("fp32_exact") is inference-only: the differentiable forward (stash + fused '
                                          'backward) has no exact-fp32 instantiation; use set_precision("fp32") for gradients')
            n_chunks = stash["n_chunks"]
        if n_chunks is None:
            n_chunks = hip.nerf_suggest_chunks(B, img_size, N_samples)
        R = img_size * img_size
        sdf = torch.empty(B, R, N_samples, device=dev) if return_sdf else None
        features, thumb, xyz, mask = hip.nerf_render_maps(cam_poses=cam_poses.float().contiguous(), focals=focals.float().reshape(B).contiguous(),
                        near_=near.float().reshape(B).contiguous(), far_=far.float().reshape(B).contiguous(),
                        perturb_u=None if perturb_u is None else perturb_u.float().reshape(B, R).contiguous(),
                        w_first=net.pts_linears[0].weight, packed=packed, w_view=net.views_linears.weight, film=film,
                        layer_bias=layer_bias, w_sigma=net.sigma_linear.weight, w_rgb=net.rgb_linear.weight,
                        b_sigma=net.sigma_linear.bias, b_rgb=net.rgb_linear.bias, sigmoid_beta=self.sigmoid_beta,
                        B=B, img_size=img_size, n_samples=N_samples, hidden=H, depth=D,
                        static_viewdirs=int(bool(static_viewdirs)), n_chunks=n_chunks, sdf=sdf,
                        raw_density=not self.with_sdf, packed32=None if stash is not None else self.packed32(),
                        stash=None if stash is None else stash["stash"], bwd_sdf=None if stash is None else stash["sdf"],
                        bwd_crgb=None if stash is None else stash["crgb"], zero_words=zero_words, planar_mask=planar_mask)
        if sdf is not None:
            sdf = sdf.view(B, img_size, img_size, N_samples, 1)
        return thumb, features, sdf, mask, xyz

    @torch.no_grad()
    def run_network(self, inputs, viewdirs, styles=None):
        """volume_renderer.py:282-303: per-point (rgb, sdf, features) for normalised points + per-ray view directions."""
        dirs = viewdirs.unsqueeze(-2).expand(inputs.shape)
        return self.network(torch.cat([inputs, dirs], -1), styles=styles)

    @torch.no_grad()
    def forward(self, pts, rays_d, viewdirs, z_vals, near, far, styles=None, return_eikonal=False, N_samples_forward=None):
        """The reference entry (volume_renderer.py:192-303): explicit sample points instead of a camera.
        pts (b h w N 3) or (b hw N 3); rays_d / viewdirs (b h w 3) | (b hw 3); z_vals (b h w N) | (b hw N); near / far
        (b 1 1); styles (b, D+1, style_dim) -> rgb_map (.., 3), feature_map (.., C), sdf (.., N, 1), mask (.., 2),
        xyz (.., 3), eikonal_term (None).  Runs the same fused kernel in its explicit-geometry instantiation;
        `N_samples_forward` (a memory bound of the reference) is accepted and ignored."""
        if return_eikonal:
            raise NotImplementedError("eikonal term needs double backward (training-only)")
        B = pts.shape[0]
        lead = pts.shape[:-2]
        N = pts.shape[-2]
        D, H = self.N_layers_renderer, self.hidden_dim
        dev = pts.device
        p = pts.float().reshape(B, -1, N, 3).contiguous()
        R = p.shape[1]
        d = rays_d.float().reshape(B, R, 3).contiguous()
        v = viewdirs.float().reshape(B, R, 3).contiguous()
        z = z_vals.float().reshape(B, R, N).contiguous()
        net = self.network
        packed, layer_bias = self._derived_buffers()
        styles_buf, film, tab = self._film_table(B, dev)
        styles_buf.copy_(styles)
        tab.run(B)
        n_chunks = max(1, min(N, hip.nerf_suggest_chunks(B, max(1, int(R ** 0.5)), N)))
        sdf = torch.empty(B, R, N, device=dev)
        features, thumb, xyz, mask = hip.nerf_render_maps(near_=near.float().reshape(B).contiguous(), far_=far.float().reshape(B).contiguous(),
                        w_first=net.pts_linears[0].weight, packed=packed, w_view=net.views_linears.weight, film=film,
                        layer_bias=layer_bias, w_sigma=net.sigma_linear.weight, w_rgb=net.rgb_linear.weight,
                        b_sigma=net.sigma_linear.bias, b_rgb=net.rgb_linear.bias, sigmoid_beta=self.sigmoid_beta,
                        B=B, img_size=1, n_samples=N, hidden=H, depth=D, static_viewdirs=0, n_chunks=n_chunks,
                        sdf=sdf, x_pts=p, x_rays_d=d, x_viewdirs=v, x_z_vals=z, n_rays=R, raw_density=not self.with_sdf,
                        packed32=self.packed32())
        to_rays = lambda t: t.view(B, t.shape[1], R).transpose(1, 2).reshape(*lead, t.shape[1]).contiguous()
        return to_rays(thumb), to_rays(features), sdf.view(*lead, N, 1), to_rays(mask), to_rays(xyz), None
