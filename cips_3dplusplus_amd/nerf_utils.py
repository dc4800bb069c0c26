"""`Render` / `Camera` of the reference's nerf_utils module on the HIP path
(/root/reference/exp/cips3d/nerf_utils.py:11-338,341-564): same static-method names, argument names, tensor layouts
and return tuples, so code written against `exp.cips3d.nerf_utils` keeps working.

`Generator.forward` does not call these — it runs the fused kernel (csrc/nerf.hip), which performs the same steps in
registers; they exist for callers that drive the renderer piecewise (the reference's own `VolumeFeatureRenderer.forward`
signature takes their outputs).  RNG sites stay where the reference has them: `get_z_vals(perturb=True)` draws one
uniform per ray with `torch.rand` on the device.  Inference only (no autograd through these entry points).
"""
import torch

from . import hip
from .camera import Camera  # noqa: F401  (nerf_utils.Camera of the reference)


def _flat_rays(t, last):
    """(b, h, w, ...) or (b, hw, ...) -> (b, R, ...) contiguous fp32 + the leading shape to restore."""
    lead = t.shape[:t.dim() - last]
    return t.float().reshape(lead[0], -1, *t.shape[t.dim() - last:]).contiguous(), lead


class Render(object):
    @staticmethod
    @torch.no_grad()
    def get_rays_in_world(focal, img_size, c2w, static_viewdirs=False):
        """nerf_utils.py:18-66 -> rays_o, rays_d, viewdirs, each (b, h, w, 3)."""
        return hip.rays_in_world(c2w, focal, img_size, static_viewdirs)

    @staticmethod
    @torch.no_grad()
    def get_z_vals(near, far, rays_d, N_samples, perturb=True, offset_sampling=True, perturb_u=None):
        """nerf_utils.py:69-121 -> (b, h, w, N_samples).  The offset-sampling branch is the generator path's (one uniform per
        ray, perturb_u [b,h,w,1]); the classic stratified branch (`mlp_init_pass`) draws one uniform per SAMPLE
        (perturb_u [b,h,w,N_samples])."""
        b, h, w, _ = rays_d.shape
        if not offset_sampling:
            t = None
            if perturb:
                t = perturb_u if perturb_u is not None else torch.rand(b, h, w, N_samples, device=rays_d.device)
            return hip.z_vals(near, far, b, h * w, N_samples, perturb_u=t, stratified=True).view(b, h, w, N_samples)
        u = None
        if perturb:
            u = perturb_u if perturb_u is not None else torch.rand(b, h, w, 1, device=rays_d.device)
        return hip.z_vals(near, far, b, h * w, N_samples, perturb_u=u).view(b, h, w, N_samples)

    @staticmethod
    @torch.no_grad()
    def get_points(rays_o, rays_d, z_vals):
        """nerf_utils.py:136-170 -> pts (b, h, w, N_samples, 3)."""
        o, lead = _flat_rays(rays_o, 1)
        d, _ = _flat_rays(rays_d, 1)
        z, _ = _flat_rays(z_vals, 1)
        pts, _ = hip.ray_points(o, d, z, want_pts=True, want_normalized=False)
        return pts.view(*lead, z.shape[-1], 3)

    @staticmethod
    @torch.no_grad()
    def normalize_points(pts, near, far):
        """nerf_utils.py:124-133: pts * 2 / (far - near), per batch element."""
        b = pts.shape[0]
        flat = pts.float().reshape(b, -1, 1, 3).contiguous()          # every point is its own "ray" with z = 1
        ones = torch.ones(b, flat.shape[1], 1, device=pts.device)
        zero = torch.zeros(b, flat.shape[1], 3, device=pts.device)
        _, ptsn = hip.ray_points(zero, flat.view(b, -1, 3), ones, near, far, want_pts=False, want_normalized=True)
        return ptsn.view(pts.shape)

    @staticmethod
    @torch.no_grad()
    def prepare_nerf_inputs(focal, img_size, cam_poses, near, far, N_samples, perturb, static_viewdirs=False,
                            perturb_u=None, **kwargs):
        """nerf_utils.py:173-218 -> pts (b h w N 3), rays_d (b h w 3), viewdirs (b h w 3), z_vals (b h w N)."""
        rays_o, rays_d, viewdirs = Render.get_rays_in_world(focal=focal, img_size=img_size, c2w=cam_poses,
                                                            static_viewdirs=static_viewdirs)
        z_vals = Render.get_z_vals(near=near, far=far, rays_d=rays_d, N_samples=N_samples, perturb=perturb,
                                   offset_sampling=True, perturb_u=perturb_u)
        pts = Render.get_points(rays_o=rays_o, rays_d=rays_d, z_vals=z_vals)
        return pts, rays_d, viewdirs, z_vals

    @staticmethod
    @torch.no_grad()
    def volume_integration(rgb, sdf, features, z_vals, rays_d, pts, with_sdf=True, sigmoid_beta=None, return_eikonal=False,
                           raw_noise_std=0., force_background=False):
        """nerf_utils.py:231-338 -> rgb_map, feature_map, xyz, mask (.., 2), eikonal_term (None).  with_sdf=False is the raw
        density branch (softplus of the network output, optionally + raw_noise_std * randn, :288-297), force_background the
        last-sample override (:309-310).  The eikonal term (a training regulariser built on autograd.grad, :270-275) is not
        computed."""
        if return_eikonal:
            raise NotImplementedError("the eikonal term needs double backward (training-only)")
        if with_sdf and sigmoid_beta is None:
            raise ValueError("with_sdf=True needs sigmoid_beta")
        z, lead = _flat_rays(z_vals, 1)
        n, N = z.shape[0] * z.shape[1], z.shape[2]
        r3 = rgb.float().reshape(n, N, 3).contiguous()
        s1 = sdf.float().reshape(n, N).contiguous()
        f = None if features is None else features.float().reshape(n, N, features.shape[-1]).contiguous()
        d = rays_d.float().reshape(n, 3).contiguous()
        p = pts.float().reshape(n, N, 3).contiguous()
        beta = None
        if with_sdf:
            beta = sigmoid_beta if torch.is_tensor(sigmoid_beta) else torch.tensor([float(sigmoid_beta)], device=z.device)
            beta = beta.detach().float().reshape(1)
        elif raw_noise_std > 0:
            s1 = s1 + torch.randn_like(s1) * raw_noise_std
        rgb_map, fmap, xyz, mask = hip.volume_integration(r3, s1, f, z.view(n, N), d, p, beta, raw_density=not with_sdf,
                                                          force_background=bool(force_background))
        return (rgb_map.view(*lead, 3), None if fmap is None else fmap.view(*lead, -1), xyz.view(*lead, 3),
                mask.view(*lead, 2), None)
