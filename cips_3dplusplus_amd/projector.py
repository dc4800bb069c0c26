"""Flip-inversion optimisation loop over the HIP forward + backward (BASELINE config 5).

Follows `StyleGAN2Projector_Flip.project_wplus` of /root/reference/exp/cips3d/models/projector_v10.py:915-1280:
three Adam optimisers (camera angles; NeRF W+ styles; decoder W+ styles + decoder parameters [+ noise buffers]), the
cosine ramp-down / linear ramp-up learning-rate multiplier (:174-186), a pose phase with the decoder frozen (lr 0),
an appearance phase that starts from the truncated NeRF style and flips the decoder styles of the (image, mirrored
image) pair every `flip_w_decoder_every` steps, and the noise regulariser (:1179-1192).  The batch is the image and its
horizontal flip rendered from mirrored azimuths.

What is NOT here: the VGG perceptual features (`get_perceptual_fea`, pretrained weights are not obtainable) -- the loss
is a callable; the default is the surrogate of SURVEY 8d config 5 (MSE on `rgb` + `thumb_weight` x MSE on `thumb_rgb`
against fixed targets).  Streamlit charts, videos and PSNR logging are out of scope.

The three Adam optimisers of the reference run as `optim.HipAdam` (csrc/optim.hip: torch.optim.Adam's update rule, one
bandwidth-bound launch per 48 tensors; CIPS3D_HIP_ADAM=0: torch.optim.Adam, fused where torch offers it).
"""
import copy
import math
import os

import torch
from torch import nn

from .camera import Camera


def _adam(groups):
    """Adam over the given parameter groups: the HIP kernel (optim.HipAdam: torch.optim.Adam's update rule, one launch per 48
    tensors) for CUDA parameters; CIPS3D_HIP_ADAM=0: torch.optim.Adam (fused=True where torch offers it, CIPS3D_FUSED_ADAM=0:
    its default form)."""
    on_gpu = all(p.is_cuda and p.dtype == torch.float32 for g in groups for p in g["params"])
    if os.environ.get("CIPS3D_HIP_ADAM", "1") != "0" and on_gpu:
        from .optim import HipAdam
        return HipAdam(groups)
    if os.environ.get("CIPS3D_FUSED_ADAM", "1") != "0" and all(p.is_cuda for g in groups for p in g["params"]):
        try:
            return torch.optim.Adam(groups, fused=True)
        except (RuntimeError, TypeError):
            pass
    return torch.optim.Adam(groups)


def _all_hip_adam(opts):
    from .optim import HipAdam
    return len(opts) > 0 and all(isinstance(o, HipAdam) for o in opts)


def cur_lr(step, num_steps, initial_learning_rate=1.0, lr_rampdown_length=0.25, lr_rampup_length=0.05):
    """projector_v10.py:174-186."""
    t = step / num_steps
    ramp = min(1.0, (1.0 - t) / lr_rampdown_length)
    ramp = 0.5 - 0.5 * math.cos(ramp * math.pi)
    ramp = ramp * min(1.0, t / lr_rampup_length)
    return initial_learning_rate * ramp


def _scale_lr(opt, mul):
    for g in opt.param_groups:          # tl2 mul_optimizer_lr: lr = initial_lr * mul
        g["lr"] = g["initial_lr"] * mul


def _set_lr(opt, lr):
    for g in opt.param_groups:
        g["lr"] = lr


def noise_regulariser(noise_bufs):
    """projector_v10.py:1179-1192 (StyleGAN2's multi-scale autocorrelation penalty)."""
    reg = 0
    for v in noise_bufs:
        noise = v
        while True:
            reg = reg + (noise * torch.roll(noise, shifts=1, dims=3)).mean() ** 2
            reg = reg + (noise * torch.roll(noise, shifts=1, dims=2)).mean() ** 2
            if noise.shape[2] <= 8:
                break
            noise = torch.nn.functional.avg_pool2d(noise, kernel_size=2)
    return reg


FUSED_LOSS = os.environ.get("CIPS3D_FUSED_LOSS", "1") != "0"      # 0: the torch expression (A/B knob)


def surrogate_loss(target_rgb, target_thumb, rgb_weight=1.0, thumb_weight=50.0):
    """rgb_weight mse(rgb, target) + thumb_weight mse(thumb, target_thumb): the structure of the reference's loss
    (projector_v10.py:1173-1178) on the images themselves.  On the GPU it is one autograd node (autograd.SqDiffPairFn)."""
    def loss(rgb, thumb):
        if (FUSED_LOSS and rgb.is_cuda and rgb.dtype == torch.float32 and thumb.dtype == torch.float32
                and rgb.shape == target_rgb.shape and thumb.shape == target_thumb.shape
                and not target_rgb.requires_grad and not target_thumb.requires_grad):
            from . import autograd as AG
            return AG.weighted_mse_pair(rgb, target_rgb.to(rgb.device, torch.float32), rgb_weight,
                                        thumb, target_thumb.to(thumb.device, torch.float32), thumb_weight)
        return rgb_weight * ((rgb - target_rgb) ** 2).mean() + thumb_weight * ((thumb - target_thumb) ** 2).mean()
    return loss


class FlipProjector:
    def __init__(self, G, device="cuda"):
        self.G, self.device = G, device

    # ---- optimisers (projector_v10.py:279-390)
    def _cam_optimizer(self, optim_cam, lr_cam, azim_init, bs):
        """-> (locations [bs, 2] = (azim, elev) per view, optimiser).  The reference keeps azim and elev as two [bs, 1] parameters of
        one Adam group and concatenates them every step (projector_v10.py:279-300, 240-241); Adam is element-wise, so ONE [bs, 2]
        parameter takes exactly the same steps -- without the cat launch of every forward and the two slice copies its backward is.
        `azim` / `elev` of the returned dict and of `on_step` are its two columns."""
        loc = torch.zeros(bs, 2, device=self.device)
        loc[:, 0] = torch.tensor(azim_init[:bs], dtype=torch.float32)
        groups = []
        if optim_cam:
            loc = nn.Parameter(loc)
            groups.append({"params": [loc], "lr": lr_cam, "initial_lr": lr_cam, "betas": (0.9, 0.999)})
        return loc, _adam(groups) if groups else None

    def _render_optimizer(self, G, mean_r, optim_render_w, lr_render_w, bs, optim_render_params=False):
        w = mean_r.detach().reshape(1, 1, -1).repeat(bs, G.N_layers_renderer + 1, 1).contiguous()
        groups = []
        if optim_render_w:
            w = nn.Parameter(w)
            groups.append({"params": [w], "lr": lr_render_w, "initial_lr": lr_render_w, "betas": (0.9, 0.999)})
        if optim_render_params:                          # projector_v10.py:866-872 (lr fixed at 1e-4 there)
            groups.append({"params": list(G.renderer.parameters()), "lr": 0.0001, "initial_lr": 0.0001, "betas": (0.9, 0.999)})
        return w, _adam(groups) if groups else None

    def _decoder_optimizer(self, G, mean_d, optim_decoder_w, optim_decoder_params, optim_noise_bufs, zero_noise_bufs,
                           lr_decoder_w, lr_decoder_params, lr_noise, bs, start_size):
        w = mean_d.detach().reshape(1, 1, -1).repeat(bs, G.decoder.n_latent, 1).contiguous()
        groups = []
        if optim_decoder_w:
            w = nn.Parameter(w)
            groups.append({"params": [w], "lr": lr_decoder_w, "initial_lr": lr_decoder_w, "betas": (0.9, 0.999)})
        if optim_decoder_params:
            groups.append({"params": list(G.decoder.parameters()), "lr": lr_decoder_params,
                           "initial_lr": lr_decoder_params, "betas": (0.9, 0.999)})
        noise_bufs = G.create_noise_bufs(start_size, self.device)
        if zero_noise_bufs:
            noise_bufs = [torch.zeros_like(b) for b in noise_bufs]
        if optim_noise_bufs:
            noise_bufs = [nn.Parameter(b) for b in noise_bufs]
            groups.append({"params": noise_bufs, "lr": lr_noise, "initial_lr": lr_noise, "betas": (0.9, 0.999)})
        return w, noise_bufs, _adam(groups) if groups else None

    # ---- one generator call of the loop (projector_v10.py:211-277)
    def g_forward(self, G, style_render, style_decoder, noise_bufs, cam_cfg, nerf_cfg, rot, trans=None, flip_w_decoder=False):
        """rot, trans: azimuth and elevation [B, 1] each -- or rot = the [B, 2] locations and trans = None."""
        cam_cfg = dict(cam_cfg)
        img_size = cam_cfg.pop("img_size")
        cam_cfg = {k: v for k, v in cam_cfg.items() if k in ("fov_ang", "dist_radius")}
        extr, focal, near, far, _ = Camera.generate_camera_params(img_size, self.device,
                                                                  locations=rot if trans is None else torch.cat([rot, trans], 1),
                                                                  **cam_cfg)
        if flip_w_decoder:
            style_decoder = style_decoder.detach().flip(dims=(0,))      # only the decoder parameters are updated
        r = G(zs=[None, None], style_render=style_render, style_decoder=style_decoder, cam_poses=extr, focals=focal,
              img_size=img_size, near=near, far=far, noise_bufs=noise_bufs, nerf_cfg=nerf_cfg, renderer_detach=False)
        return r["rgb"], r["thumb_rgb"], r["mask"]

    def project_wplus(self, cam_cfg, nerf_cfg, loss_fn, N_steps_pose=200, N_steps_app=0, optim_cam=True, optim_render_w=True,
                      optim_render_params=False, optim_decoder_w=True, optim_decoder_params=True, optim_noise_bufs=False, zero_noise_bufs=True,
                      bs_cam=2, bs_render=1, bs_decoder=2, lr_cam=0.02, lr_render_w=0.001, lr_decoder_w=0.01,
                      lr_decoder_params=0.005, lr_noise=0.001, truncation_psi=1.0, flip_w_decoder_every=10,
                      azim_init=(0.0, 0.0), w_avg_samples=10000, regularize_noise_weight=1e5, on_step=None):
        """Returns the dict `checkpoint.save_inversion` writes (azim, elev, W+ styles, state dicts, noise)."""
        G = copy.deepcopy(self.G).eval().requires_grad_(False).to(self.device)
        if optim_render_params:                              # projector_v10.py:967-968
            G.renderer.requires_grad_(True)
        G.decoder.requires_grad_(True)
        with torch.no_grad():
            mean_r, mean_d = G.get_mean_latent(w_avg_samples, self.device)
        loc, opt_cam = self._cam_optimizer(optim_cam, lr_cam, list(azim_init), bs_cam)
        azim, elev = loc.detach()[:, 0:1], loc.detach()[:, 1:2]          # (views: they follow the optimiser's in-place updates)
        one = torch.ones((), device=self.device)                         # d loss / d loss, allocated once (backward() would fill one per step)
        w_render, opt_render = self._render_optimizer(G, mean_r, optim_render_w, lr_render_w, bs_render, optim_render_params)
        w_decoder, noise_bufs, opt_dec = self._decoder_optimizer(
            G, mean_d, optim_decoder_w, optim_decoder_params, optim_noise_bufs, zero_noise_bufs, lr_decoder_w,
            lr_decoder_params, lr_noise, bs_decoder, cam_cfg["img_size"])
        opts = [o for o in (opt_cam, opt_render, opt_dec) if o is not None]
        N_steps = N_steps_pose + N_steps_app
        history = []
        for step in range(N_steps):
            if step < N_steps_pose:
                lr_mul = cur_lr(step, N_steps_pose)
            else:
                lr_mul = cur_lr(step - N_steps_pose, N_steps_app, lr_rampup_length=0.25)
            for o in opts:
                _scale_lr(o, lr_mul)
            flip_w_decoder = False
            if step < N_steps_pose:                      # camera + NeRF style; decoder frozen
                if opt_dec is not None:
                    _set_lr(opt_dec, 0)
            else:                                        # camera + decoder
                if step == N_steps_pose:
                    with torch.no_grad():
                        w_render.copy_(torch.lerp(mean_r.reshape(1, 1, -1).expand_as(w_render), w_render, truncation_psi))
                if (step + flip_w_decoder_every - 1) % flip_w_decoder_every == 0 and step != N_steps - 1:
                    flip_w_decoder = True
            # (one NeRF latent for both views, bs_render = 1: the generator broadcasts it inside the FiLM table's launch)
            rgb, thumb, _ = self.g_forward(
                G, w_render, w_decoder if w_decoder.shape[0] == 2 else w_decoder.repeat(2, 1, 1), noise_bufs, cam_cfg, nerf_cfg,
                rot=loc, flip_w_decoder=flip_w_decoder)
            loss = loss_fn(rgb, thumb)
            if optim_noise_bufs and regularize_noise_weight > 0:
                loss = loss + regularize_noise_weight * noise_regulariser(noise_bufs)
            for o in opts:
                o.zero_grad(set_to_none=True)
            loss.backward(one if loss.dim() == 0 and loss.dtype == one.dtype else None)
            if _all_hip_adam(opts):
                from .optim import step_many
                step_many(opts)                          # the three optimisers' tensors share launches
            else:
                for o in opts:
                    o.step()
            if on_step is not None:
                on_step(step, loss, azim, elev)
            else:
                history.append(loss.detach())
        return {"azim": azim.clone(), "elev": elev.clone(), "w_render_opt": w_render.detach(),
                "w_decoder_opt": w_decoder.detach(), "render_state_dict": G.renderer.state_dict(),
                "decoder_state_dict": G.decoder.state_dict(), "noise_bufs": [b.detach() for b in noise_bufs], "padding": 0,
                "loss_history": torch.stack(history).cpu() if history else None, "G": G}
