"""Generator of CIPS-3D++ behind the reference call surface
(/root/reference/exp/cips3d/models/model_v3.py:808-1490): same constructor dict arguments, same
`forward(...)` keyword arguments, same `ret_maps` keys, same `state_dict` key names.

forward = mapping networks (cips3d_linear chain) -> fused NeRF render (csrc/nerf.hip) -> decoder
(csrc/decoder.hip).  Inference path only: no autograd graph is built (the reference's training-only
switches raise, see `forward`).  RNG sites of the reference stay RNG sites here (per-ray jitter when
`perturb`, fresh decoder noise when `noise_bufs` is None, 10 000 z's in `get_mean_latent`) and are
injectable for parity runs (`perturb_u=`, `noise_bufs=`, preset `style_render_mean` /
`style_decoder_mean` attributes).
"""
import torch
from torch import nn

from .decoder import Decoder, EqualLinear, MappingLinear, PixelNorm
from .renderer import VolumeFeatureRenderer


class Generator(nn.Module):
    def __init__(self, enable_decoder, freeze_renderer, renderer_detach=True, predict_rgb_residual=False,
                 scale_factor=None, renderer_cfg={}, mapping_renderer_cfg={}, decoder_cfg={},
                 mapping_decoder_cfg={}, **kwargs):
        super().__init__()
        if not enable_decoder:
            raise NotImplementedError("enable_decoder=False raises in the reference as well (model_v3.py:1025-1026)")
        self.enable_decoder, self.freeze_renderer, self.renderer_detach = enable_decoder, freeze_renderer, renderer_detach
        self.predict_rgb_residual, self.scale_factor = predict_rgb_residual, scale_factor
        self.renderer_cfg, self.mapping_renderer_cfg = renderer_cfg, mapping_renderer_cfg
        self.decoder_cfg, self.mapping_decoder_cfg = decoder_cfg, mapping_decoder_cfg
        self.module_name_list = []

        # construction order = reference order (renderer, style, decoder, style_decoder): seeded init parity
        self.renderer = VolumeFeatureRenderer(style_dim=mapping_renderer_cfg["style_dim"], **renderer_cfg)
        self.module_name_list.append("renderer")
        self.N_layers_renderer = self.renderer.N_layers_renderer
        self.create_mapping_nerf(**mapping_renderer_cfg)
        self.z_dim = mapping_renderer_cfg["z_dim"]
        self.decoder = Decoder(style_dim=mapping_decoder_cfg["style_dim"],
                               **{**decoder_cfg, "in_channel": renderer_cfg["hidden_dim"]})
        self.module_name_list.append("decoder")
        self.create_mapping_decoder(z_dim=mapping_renderer_cfg["style_dim"], **mapping_decoder_cfg)

    # ---------------------------------------------------------------- construction helpers
    def create_mapping_nerf(self, z_dim, style_dim, N_layers):
        self.style = nn.Sequential(*[MappingLinear(z_dim, style_dim, activation="fused_lrelu") for _ in range(N_layers)])
        self.module_name_list += [f"style.{i}" for i in range(N_layers)] + ["style"]

    def create_mapping_decoder(self, z_dim, style_dim, lr_mul_mapping, N_layers):
        layers = [PixelNorm(), EqualLinear(z_dim, style_dim, lr_mul=lr_mul_mapping, activation="fused_lrelu")]
        layers += [EqualLinear(style_dim, style_dim, lr_mul=lr_mul_mapping, activation="fused_lrelu")
                   for _ in range(N_layers - 1)]
        self.style_decoder = nn.Sequential(*layers)
        self.module_name_list += [f"style_decoder.{i}" for i in range(N_layers + 1)] + ["style_decoder"]

    # ---------------------------------------------------------------- mapping networks
    def _run_style(self, z, trunc_mean=None, psi=1.0):
        h = z.float().contiguous()
        n = len(self.style)
        for i, layer in enumerate(self.style):
            last = i == n - 1
            h = layer(h, trunc_mean=trunc_mean if last else None, trunc_psi=psi)
        return h

    def _run_style_decoder(self, z, trunc_mean=None, psi=1.0):
        h = z.float().contiguous()
        layers = list(self.style_decoder)[1:]
        for i, layer in enumerate(layers):
            last = i == len(layers) - 1
            h = layer(h, pixelnorm=(i == 0), trunc_mean=trunc_mean if last else None, trunc_psi=psi)
        return h

    @torch.no_grad()
    def get_mean_latent(self, N_noises, device):
        """model_v3.py:1285-1297."""
        zr = torch.randn(N_noises, self.z_dim, device=device)
        mean_r = self._run_style(zr).mean(0, keepdim=True)
        zd = torch.randn(N_noises, self.z_dim, device=device)
        mean_d = self._run_style_decoder(zd).mean(0, keepdim=True)
        return mean_r, mean_d

    def mapping_renderer(self, zs, truncation, style_render_mean=None, **kwargs):
        """model_v3.py:1402-1418."""
        m = style_render_mean.reshape(-1).contiguous() if truncation < 1 else None
        latents = [self._run_style(z, m, float(truncation)) for z in zs]
        return latents[0].unsqueeze(1).repeat(1, self.N_layers_renderer + 1, 1), latents

    def mapping_decoder(self, latents, truncation, style_decoder_mean, inject_index=None):
        """model_v3.py:1350-1378."""
        m = style_decoder_mean.reshape(-1).contiguous() if truncation < 1 else None
        styles = [self._run_style_decoder(z, m, float(truncation)) for z in latents]
        n_latent = self.decoder.n_latent
        if len(styles) < 2:
            return styles[0].unsqueeze(1).repeat(1, n_latent, 1)
        if inject_index is None:
            inject_index = n_latent
        return torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                          styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)

    def mapping_networks(self, zs, truncation, inject_index, path_reg=False, style_render=None, style_decoder=None,
                         recompute_mean=False):
        """model_v3.py:1299-1348."""
        if style_render is not None and style_decoder is not None:
            return style_render, style_decoder
        if (style_render is None) != (style_decoder is None):
            raise NotImplementedError
        if truncation < 1:
            if recompute_mean or not hasattr(self, "style_render_mean") or not hasattr(self, "style_decoder_mean"):
                self.style_render_mean, self.style_decoder_mean = self.get_mean_latent(10000, zs[0].device)
            mean_r, mean_d = self.style_render_mean, self.style_decoder_mean
        else:
            mean_r = mean_d = None
        style_render, _ = self.mapping_renderer([zs[0]], truncation, mean_r)
        style_decoder = self.mapping_decoder([zs[1]], truncation, mean_d, inject_index)
        return style_render, style_decoder

    def get_ws(self, zs, truncation, device):
        """model_v3.py:1472-1490."""
        mean_r, mean_d = self.get_mean_latent(10000, device)
        w_r = self._run_style(zs[0], mean_r.reshape(-1).contiguous(), float(truncation))
        w_d = self._run_style_decoder(zs[1], mean_d.reshape(-1).contiguous(), float(truncation))
        return (w_r[:, None, :].repeat(1, self.N_layers_renderer + 1, 1),
                w_d[:, None, :].repeat(1, self.decoder.n_latent, 1))

    @torch.no_grad()
    def rays_forward(self, N_rays_forward, pts, rays_d, viewdirs, z_vals, near, far, style_render, style_decoder=None,
                     noise_bufs=None, eikonal_reg=False, cam_poses=None, project_noise=False, mesh_path=None,
                     renderer_detach=None, N_samples_forward=None):
        """model_v3.py:1201-1268: the NeRF half for caller-made geometry, pts (b, hw, n, 3) etc. -> thumb_rgb (b, hw, 3),
        sdf (b, hw, n, 1), mask (b, hw, 2), xyz (b, hw, 3), features (b, hw, C), eikonal_term.  The reference loops over
        `N_rays_forward`-sized ray chunks to bound memory; the fused kernel keeps no per-point activations, so one call
        covers all rays (the chunk size is accepted and ignored)."""
        if eikonal_reg:
            raise NotImplementedError("eikonal_reg needs double backward (training-only)")
        thumb, feats, sdf, mask, xyz, eik = self.renderer(pts=pts, rays_d=rays_d, viewdirs=viewdirs, z_vals=z_vals, near=near,
                                                          far=far, styles=style_render)
        return thumb, sdf, mask, xyz, feats, eik

    def init_forward(self, *args, **kwargs):
        raise NotImplementedError("init_forward / mlp_init_pass is the SDF sphere-initialisation pass of training "
                                  "(model_v3.py:1449-1470, volume_renderer.py: mlp_init_pass): out of scope")

    def set_decoder_precision(self, precision):
        """"fp32" (default: fp32-equivalent; the stand-alone decoder GEMMs use split-fp16 products, see Decoder.set_precision),
        "fp32_exact" (the fp32 MFMA everywhere in the decoder), "bf16" (BASELINE config 3: the decoder's GEMM
        operands are rounded to bf16 in registers, fp32 accumulate; the NeRF renderer stays fp32 because gamma ~ 30
        re-amplifies input error in every SIREN layer) or "bf16_storage" (bf16 + the pre-FIR activations of the fused
        up-sampling stages stored as bf16: half the activation bytes of the >= 128^2 stages)."""
        self.decoder.set_precision(precision)
        return self

    def set_precision(self, precision):
        """Arithmetic of the whole forward.  "fp32_exact": IEEE-fp32 products everywhere a GEMM computes -- the decoder on the
        fp32 matrix instruction AND the renderer's point MLP (VolumeFeatureRenderer.set_precision) -- i.e. the reference's own
        arithmetic (cips3d/volume_renderer.py:74-85, models/model_v3.py:296-312); any other value: the decoder precision of
        that name with the renderer in its default (fp32-equivalent split-fp16 products).  "fp32_exact" is an inference mode:
        the differentiable forward raises NotImplementedError in it (renderer.py: no exact-fp32 stash / fused backward)."""
        self.decoder.set_precision(precision)
        self.renderer.set_precision("fp32_exact" if precision == "fp32_exact" else "fp32")
        return self

    # ---------------------------------------------------------------- noise
    def create_noise_bufs(self, start_size, device):
        return self.decoder.create_noise_bufs(start_size=start_size, device=device)

    def get_noise_bufs(self, noise_bufs, randomize_noise):
        if noise_bufs is None:
            if not randomize_noise:
                raise NotImplementedError
            noise_bufs = [None] * self.decoder.num_layers
        return noise_bufs

    # ---------------------------------------------------------------- one-call forward
    def _forward_plan(self, B, img_size, N, static):
        """cips3d_generator_forward plan for this shape, or None when the configuration cannot be planned
        (k=3 / untiled channel counts use the per-op path below)."""
        from . import plan as _plan
        if not hasattr(self, "_plans"):
            self._plans = {}
        # Plans (workspaces, style tables) belong to a LANE = the stream the call is issued on: forwards on different streams share
        # nothing they write and may be in flight together (pipeline.ViewPipeline alternates independent views between two streams:
        # one view's latency-bound style phase runs under another's large kernels).  Lane 0 = the first stream seen (self._plans).
        lane = self._lane_of_current_stream()
        plans = self._plans if lane == 0 else self.__dict__.setdefault("_lane_plans", {}).setdefault(lane, {})
        key = (B, img_size, N, static)
        ent = plans.get(key, 0)
        if ent is None:
            return None
        if ent == 0 or ent.key != _plan.ForwardPlan.weights_key(self):
            try:
                ent = _plan.ForwardPlan(self, B, img_size, N, static, lane=lane)
            except _plan.PlanUnsupported:
                ent = None
            plans[key] = ent
        return ent

    MAX_LANES = 8

    def _lane_of_current_stream(self):
        if not torch.cuda.is_available():
            return 0
        if torch.cuda.is_current_stream_capturing():
            # a forward captured into a graph runs on lane 0's plan (capture streams come and go; building a plan inside a capture
            # is not possible): do not replay it while an eager forward of lane 0 is in flight
            return 0
        from ._lib import stream_ptr
        sid = stream_ptr()
        lanes = self.__dict__.setdefault("_stream_lanes", {})
        lane = lanes.get(sid)
        if lane is None:
            if len(lanes) >= self.MAX_LANES:
                raise RuntimeError(f"Generator.forward was called on more than {self.MAX_LANES} different streams: every stream keeps "
                                   "its own forward plans (workspaces, style tables); reuse streams")
            lane = lanes[sid] = len(lanes)
        return lane

    def _planned_forward(self, plan, zs, cam_poses, focals, near, far, perturb_u, noise_bufs, truncation, style_render,
                         style_decoder, return_sdf, return_xyz, fresh_perturb=False, styles_resident=None, rgb_out=None):
        from . import hip
        B = plan.B
        z_r = z_d = mean_r = mean_d = None
        ident = lambda t: (t.data_ptr(), t._version, tuple(t.shape))     # noqa: E731 (what a style table was computed from)
        # The plan keeps STRONG references to the stamped tensors (`refs`) until its next full run: a freed latent's storage could
        # otherwise be handed to a new tensor with the same address, shape and version 0, which would pass the check with stale
        # tables.  Not detected: a tensor whose storage is swapped through `.data = ...` onto memory at the same address (versions do
        # not move) -- do not rebind `.data` of latents, styles or weights between the frames of a resident sequence.
        # (a resident frame also promises unchanged WEIGHTS behind the tables: an optimiser step on the decoder or the mapping
        # networks between two frames bumps these versions; the renderer's are part of the plan's key)
        plist = self.__dict__.get("_stamp_params")
        if plist is None or plist[0] != id(self.decoder.conv1.conv.weight):
            plist = (id(self.decoder.conv1.conv.weight),
                     [p for m in (self.decoder, self.style, self.style_decoder) for p in m.parameters()])
            self.__dict__["_stamp_params"] = plist
        # (versions only grow: the sum changes whenever one of them does.  Walking ~150 parameters is ~30 us of host time: only calls
        # that belong to a sequence -- styles_resident given as True or False -- pay it; the default None records no stamp)
        wver = sum(p._version for p in plist[1]) if styles_resident is not None else None
        if style_render is not None and style_decoder is not None:
            stamp = ("w+", ident(style_render), ident(style_decoder), wver)
            refs = (style_render, style_decoder)
            if not styles_resident:
                plan.styles_r.copy_(style_render)       # explicit W+ styles bypass the mapping networks
                plan.styles_d.copy_(style_decoder)
        else:
            stamp = ("z", ident(zs[0]), ident(zs[1]), float(truncation), wver)
            refs = (zs[0], zs[1])
            if truncation < 1:
                stamp += (ident(self.style_render_mean), ident(self.style_decoder_mean))
                refs += (self.style_render_mean, self.style_decoder_mean)
            if not styles_resident:
                z_r, z_d = zs[0].float(), zs[1].float()
                if z_r.shape[0] != B or z_d.shape[0] != B:
                    # one latent for a batch of views (the multi-view loop with chunk > 1): the kernels read B rows
                    if z_r.shape[0] != 1 or z_d.shape[0] != 1:
                        raise ValueError(f"zs hold {z_r.shape[0]} / {z_d.shape[0]} latents for a batch of {B} views")
                    z_r, z_d = z_r.expand(B, -1), z_d.expand(B, -1)
                z_r, z_d = z_r.contiguous(), z_d.contiguous()
                if truncation < 1:
                    mean_r = self.style_render_mean.reshape(-1).contiguous()
                    mean_d = self.style_decoder_mean.reshape(-1).contiguous()
        marks, hip.DECODER_MARKS = hip.DECODER_MARKS, None
        events = None
        lst = hip.want_events("nerf_render")
        if lst is not None:
            ev = hip.event_pair()
            lst.append(ev)
            events = (ev[0].cuda_event, ev[1].cuda_event)
        rgb, thumb, xyz, mask, sdf = plan.run(
            z_r, z_d, cam_poses.float().contiguous(), focals.float().reshape(B).contiguous(),
            near.float().reshape(B).contiguous(), far.float().reshape(B).contiguous(),
            None if perturb_u is None else perturb_u.float().reshape(B, -1).contiguous(), noise_bufs,
            float(truncation), mean_r, mean_d, return_sdf, events, fresh_perturb=fresh_perturb,
            marks=None if marks is None else marks.io_fields(), styles_resident=bool(styles_resident),
            style_stamp=stamp if styles_resident is not None else None, rgb_out=rgb_out,
            style_refs=refs, views_in_flight=getattr(self, "_views_in_flight", 1))
        # mask arrives as [2,B,S,S] (plan.run): two contiguous [B,1,S,S] maps without a copy
        m2 = mask
        return {"rgb": rgb, "thumb_rgb": thumb, "style_decoder": None, "eikonal_term": None,
                "sdf": sdf if return_sdf else None, "xyz": xyz if return_xyz else None,
                "mask": m2[0].unsqueeze(1), "depth": m2[1].unsqueeze(1)}

    def can_emit_uint8(self, B, img_size, N_samples, static_viewdirs=False):
        """True when a forward of this shape can write its image as uint8 (`rgb_out` of dtype uint8)."""
        plan = self._forward_plan(B, img_size, int(N_samples), bool(static_viewdirs))
        return plan is not None and bool(getattr(plan, "u8_capable", False))

    # ---------------------------------------------------------------- forward
    def forward(self, zs, cam_poses, focals, img_size, near=0.88, far=1.12, truncation=1, inject_index=None,
                path_reg=False, style_render=None, style_decoder=None, noise_bufs=None, randomize_noise=True,
                eikonal_reg=False, return_sdf=False, return_xyz=False, N_rays_forward=None, N_rays_grad=None,
                N_samples_forward=None, nerf_cfg={}, recompute_mean=False, project_noise=False, mesh_path=None,
                renderer_detach=None, sample_idx_h=None, sample_idx_w=None, perturb_u=None, differentiable=None, **kwargs):
        """model_v3.py:875-1042.  Inference runs the fused path without an autograd graph.  The differentiable op chain of
        `autograd.py` (the inversion loop, projector_v10.py:211-277) is used when `differentiable=True`, or -- with the
        default `differentiable=None` -- when gradients are enabled and either an INPUT tensor (cam_poses, style_render,
        style_decoder, a noise buffer) or a DECODER parameter requires them.  Differentiable leaves: those inputs and the
        decoder's parameters.  The renderer's and the mapping networks' weights are constants of that path
        (`optim_render_params: false`, train_cips3d_compcars_v10.yaml:585): if one of them requires grad the call raises
        instead of silently returning no gradient -- freeze them (`build_generator` does) or call under `torch.no_grad()`."""
        kw = dict(zs=zs, cam_poses=cam_poses, focals=focals, img_size=img_size, near=near, far=far, truncation=truncation,
                  inject_index=inject_index, path_reg=path_reg, style_render=style_render, style_decoder=style_decoder,
                  noise_bufs=noise_bufs, randomize_noise=randomize_noise, eikonal_reg=eikonal_reg, return_sdf=return_sdf,
                  return_xyz=return_xyz, N_rays_forward=N_rays_forward, N_rays_grad=N_rays_grad,
                  N_samples_forward=N_samples_forward, nerf_cfg=nerf_cfg, recompute_mean=recompute_mean,
                  project_noise=project_noise, mesh_path=mesh_path, renderer_detach=renderer_detach,
                  sample_idx_h=sample_idx_h, sample_idx_w=sample_idx_w, perturb_u=perturb_u)
        if differentiable is None:
            differentiable = False
            if torch.is_grad_enabled():
                ins = [cam_poses, style_render, style_decoder] + list(noise_bufs or [])
                differentiable = (any(torch.is_tensor(t) and t.requires_grad for t in ins)
                                  or any(p.requires_grad for p in self.decoder.parameters())
                                  or any(p.requires_grad for p in self.renderer.parameters()))
        if differentiable:
            if not torch.is_grad_enabled():
                raise RuntimeError("differentiable=True under torch.no_grad()")
            if kwargs.get("styles_resident") or kwargs.get("rgb_out") is not None:
                raise NotImplementedError("styles_resident / rgb_out belong to the inference forward (call under torch.no_grad())")
            # the renderer's weights may be optimised (`optim_render_params`, projector_v10.py:848-872, 968); the two mapping
            # networks may not: the inversion loop never runs them (it optimises W+ directly)
            frozen = [n for mod in ("style", "style_decoder")
                      for n, p in getattr(self, mod).named_parameters(prefix=mod) if p.requires_grad]
            if frozen:
                raise NotImplementedError(
                    f"{len(frozen)} mapping-network parameters require grad (first: {frozen[0]}): their gradients "
                    "are not implemented (the inversion recipes optimise W+ styles, not the mapping networks); call "
                    "`G.requires_grad_(False); G.decoder.requires_grad_(True)` as projector_v10.py does, or run under "
                    "torch.no_grad()")
            return self._forward_grad(**kw)
        with torch.no_grad():
            return self._forward_infer(**kw, **kwargs)

    def _forward_grad(self, zs, cam_poses, focals, img_size, near, far, truncation, inject_index, path_reg, style_render,
                      style_decoder, noise_bufs, randomize_noise, eikonal_reg, return_sdf, return_xyz, N_rays_forward,
                      N_rays_grad, N_samples_forward, nerf_cfg, recompute_mean, project_noise, mesh_path, renderer_detach,
                      sample_idx_h, sample_idx_w, perturb_u):
        from . import autograd as AG
        assert len(zs) == 2
        if eikonal_reg or path_reg:
            raise NotImplementedError("eikonal_reg / path_reg need double backward (training-only)")
        if N_rays_grad is not None or sample_idx_h is not None or sample_idx_w is not None or project_noise:
            raise NotImplementedError("ray sub-sampling / project_noise are training-only")
        noise_bufs = self.get_noise_bufs(noise_bufs, randomize_noise)
        B, dev = cam_poses.shape[0], cam_poses.device
        N = int(nerf_cfg["N_samples"])
        if nerf_cfg.get("perturb", False) and perturb_u is None:
            perturb_u = torch.rand(B, img_size, img_size, 1, device=dev)
        if not nerf_cfg.get("perturb", False):
            perturb_u = None
        if style_render is None or style_decoder is None:
            with torch.no_grad():       # z -> W+ is a constant of the inversion (styles are its leaves)
                style_render, style_decoder = self.mapping_networks(
                    zs=zs, truncation=truncation, inject_index=inject_index, style_render=style_render,
                    style_decoder=style_decoder, recompute_mean=recompute_mean)

        def per_view(v):
            return (v if torch.is_tensor(v) else torch.full((B, 1, 1), float(v), device=dev)).detach()

        static = bool(nerf_cfg.get("static_viewdirs", False))
        # (style_render may hold ONE latent for the B views of the call: the FiLM table broadcasts it -- the inversion loop's
        # `w_render.repeat(2, 1, 1)`, projector_v10.py:1131, without the repeat launch and the sum its backward is)
        film = AG.film_table(self.renderer, style_render, batch=B)
        rparams = [p for _, p in AG.nerf_named_parameters(self.renderer)]
        if not any(p.requires_grad for p in rparams):
            rparams = []
        features, thumb, xyz, mask = AG.NerfRenderFn.apply(
            self.renderer, cam_poses.float(), per_view(focals), per_view(near), per_view(far), film,
            None if perturb_u is None else perturb_u.detach(), img_size, N, static, *rparams)
        if self.renderer_detach if renderer_detach is None else renderer_detach:
            features = features.detach()                                    # model_v3.py:1016-1017
        rgb = AG.decoder_forward(self.decoder, features, style_decoder, noise_bufs)
        sdf = None
        if return_sdf:
            with torch.no_grad():
                sdf = self.renderer.render(cam_poses.detach(), per_view(focals), per_view(near), per_view(far), None, img_size,
                                           N, perturb_u=perturb_u, static_viewdirs=static, return_sdf=True,
                                           film=film.detach())[2]
        m2 = mask                                       # [2,B,S,S] (NerfRenderFn renders with planar_mask)
        return {"rgb": rgb, "thumb_rgb": thumb, "style_decoder": None, "eikonal_term": None, "sdf": sdf,
                "xyz": xyz if return_xyz else None, "mask": m2[0].unsqueeze(1), "depth": m2[1].unsqueeze(1)}

    def _forward_infer(self, zs, cam_poses, focals, img_size, near=0.88, far=1.12, truncation=1, inject_index=None,
                       path_reg=False, style_render=None, style_decoder=None, noise_bufs=None, randomize_noise=True,
                       eikonal_reg=False, return_sdf=False, return_xyz=False, N_rays_forward=None, N_rays_grad=None,
                       N_samples_forward=None, nerf_cfg={}, recompute_mean=False, project_noise=False, mesh_path=None,
                       renderer_detach=None, sample_idx_h=None, sample_idx_w=None, perturb_u=None, styles_resident=None,
                       rgb_out=None, **kwargs):
        """styles_resident (extension of the reference's call surface; multiview.sample_multi_view uses it): True = this call is a
        frame of a sequence that renders ONE latent from many cameras (render_video_web_v10.py:1792-1824) -- the previous call of
        the same shape (and stream) already ran the mapping networks, the style heads and the modulate table for exactly these zs /
        styles / truncation / noise buffers, and this call reuses its tables (bit-identical to recomputing them; plan.run checks
        the promise).  False = the full first frame of such a sequence: it records what its tables were computed from.  None
        (default) = a call outside any sequence: nothing is recorded (the record walks every parameter's version: ~30 us of host
        time per call), and a resident call cannot follow it.  Ignored on the per-op path.
        rgb_out (extension): a preallocated contiguous [B, 3, R, R] tensor that receives `rgb` -- float32, or uint8 (the image
        leaves the last up-sampling stage as uint8: hip.rgb_to_uint8's bits without the fp32 image's round trip; planned
        forwards whose decoder ends in a fused stage, `can_emit_uint8`).  `ret["rgb"]` is that tensor."""
        assert len(zs) == 2
        if eikonal_reg or path_reg:
            raise NotImplementedError("eikonal_reg / path_reg are training-only (double backward); inference path here")
        if N_rays_grad is not None or sample_idx_h is not None or sample_idx_w is not None:
            raise NotImplementedError("ray sub-sampling is training-only (raises in the reference too, model_v3.py:954-956)")
        if project_noise:
            raise NotImplementedError("project_noise needs pytorch3d mesh rendering; unused by released configs")
        # N_rays_forward / N_samples_forward only bound activation memory in the reference; the fused kernel
        # never materialises per-point activations, so they are accepted and ignored.
        noise_bufs = self.get_noise_bufs(noise_bufs, randomize_noise)

        B = cam_poses.shape[0]
        dev = cam_poses.device
        N = int(nerf_cfg["N_samples"])
        # one jitter per ray (nerf_utils.py:110), drawn where it is consumed: the one-call path draws it together with the
        # decoder's fresh noise (plan.run), the per-op path below on its own
        fresh_perturb = bool(nerf_cfg.get("perturb", False)) and perturb_u is None
        if not nerf_cfg.get("perturb", False):
            perturb_u = None

        def per_view(v):
            return v if torch.is_tensor(v) else torch.full((B, 1, 1), float(v), device=dev)

        static = bool(nerf_cfg.get("static_viewdirs", False))
        plan = self._forward_plan(B, img_size, N, static)
        if (style_render is None) != (style_decoder is None):
            raise NotImplementedError
        if plan is not None and inject_index is None:
            if style_render is None and truncation < 1 and (
                    recompute_mean or not hasattr(self, "style_render_mean") or not hasattr(self, "style_decoder_mean")):
                self.style_render_mean, self.style_decoder_mean = self.get_mean_latent(10000, dev)
            return self._planned_forward(plan, zs, cam_poses, per_view(focals), per_view(near), per_view(far), perturb_u,
                                         noise_bufs, truncation, style_render, style_decoder, return_sdf, return_xyz,
                                         fresh_perturb=fresh_perturb, styles_resident=styles_resident, rgb_out=rgb_out)
        if rgb_out is not None:
            raise NotImplementedError("rgb_out needs the planned forward (k = 1 decoder with tiled widths, no style mixing)")
        if fresh_perturb:
            perturb_u = torch.rand(B, img_size, img_size, 1, device=dev)

        # ---- per-op path (k=3 / untiled shapes / style mixing): same kernels, launched one by one
        style_render, style_decoder = self.mapping_networks(
            zs=zs, truncation=truncation, inject_index=inject_index, style_render=style_render,
            style_decoder=style_decoder, recompute_mean=recompute_mean)
        thumb_rgb, features, sdf, mask, xyz = self.renderer.render(
            cam_poses, per_view(focals), per_view(near), per_view(far), style_render, img_size, N,
            perturb_u=perturb_u, static_viewdirs=nerf_cfg.get("static_viewdirs", False), return_sdf=return_sdf)
        rgb = self.decoder(features=features, styles=style_decoder, rgbd_in=None, noise=noise_bufs)
        m2 = mask.transpose(0, 1).contiguous()
        return {
            "rgb": rgb,
            "thumb_rgb": thumb_rgb,
            "style_decoder": None,
            "eikonal_term": None,
            "sdf": sdf if return_sdf else None,
            "xyz": xyz if return_xyz else None,
            "mask": m2[0].unsqueeze(1),
            "depth": m2[1].unsqueeze(1),
        }
