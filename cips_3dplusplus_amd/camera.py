"""Camera sampling and demo trajectories (host-side front end of the generator call).

`Camera.generate_camera_params` keeps the reference signature and return tuple
(/root/reference/exp/cips3d/nerf_utils.py:344-436; `_v1` with a custom up vector :466-564): the
random draws (azim/elev) are made with torch's device RNG exactly where the reference makes them,
the pose arithmetic runs in csrc/camera.hip.  Trajectories restate the three view modes of
`_sample_multi_view_web` (exp/cips3d/models/render_video_web_v10.py:1587-1649,1732-1783).
"""
import math

import torch

from . import hip


class Camera(object):
    @staticmethod
    def _angles(batch, device, locations, sweep, uniform, azim_range, elev_range):
        if locations is not None:
            return locations[:, 0].reshape(-1, 1), locations[:, 1].reshape(-1, 1), None
        if sweep:
            if isinstance(azim_range, list) and isinstance(elev_range, list):
                azim = (azim_range[0] + (azim_range[1] - azim_range[0]) / 7 * torch.arange(8, device=device)).view(-1, 1).repeat(batch, 1)
                elev = elev_range[0] + (elev_range[1] - elev_range[0]) * torch.rand(batch, 1, device=device).repeat(1, 8).view(-1, 1)
            else:
                azim = (-azim_range + (2 * azim_range / 7) * torch.arange(8, device=device)).view(-1, 1).repeat(batch, 1)
                elev = -elev_range + 2 * elev_range * torch.rand(batch, 1, device=device).repeat(1, 8).view(-1, 1)
            return azim, elev, 8
        if uniform:
            if isinstance(azim_range, list) and isinstance(elev_range, list):
                azim = azim_range[0] + (azim_range[1] - azim_range[0]) * torch.rand(batch, 1, device=device)
                elev = elev_range[0] + (elev_range[1] - elev_range[0]) * torch.rand(batch, 1, device=device)
            else:
                azim = -azim_range + 2 * azim_range * torch.rand(batch, 1, device=device)
                elev = -elev_range + 2 * elev_range * torch.rand(batch, 1, device=device)
        else:
            azim = azim_range * torch.randn(batch, 1, device=device)
            elev = elev_range * torch.randn(batch, 1, device=device)
        return azim, elev, None

    @staticmethod
    def generate_camera_params_v1(img_size, device, batch=1, locations=None, sweep=False, uniform=False,
                                  azim_range=0.3, elev_range=0.15, fov_ang=6, dist_radius=0.12, up=None):
        if locations is not None and locations.dim() == 2 and locations.shape[1] == 2:
            # (azim, elev) given: they ARE the viewpoint -- no column selects and re-concatenation (seven launches of autograd
            # plumbing per inversion step when `locations` carries gradients)
            viewpoint = locations.float()
        else:
            azim, elev, _ = Camera._angles(batch, device, locations, sweep, uniform, azim_range, elev_range)
            viewpoint = torch.cat([azim, elev], 1).float()
        n = viewpoint.shape[0]
        fov = fov_ang
        if torch.is_tensor(fov_ang) and fov_ang.numel() not in (1, n):
            raise RuntimeError("fov_ang tensor must have 1 or B elements")
        if torch.is_grad_enabled() and viewpoint.requires_grad:
            # inversion: the pose is optimised through `locations` (projector_v10.py:227-232)
            from .autograd import CameraFn
            extr, focal, near, far = CameraFn.apply(viewpoint.to(device), img_size, fov, dist_radius, up)
        else:
            extr, focal, near, far = hip.camera_params(viewpoint.to(device), img_size, fov, dist_radius, up=up)
        return extr, focal, near, far, viewpoint

    @staticmethod
    def generate_camera_params(img_size, device, batch=1, locations=None, sweep=False, uniform=False,
                               azim_range=0.3, elev_range=0.15, fov_ang=6, dist_radius=0.12):
        return Camera.generate_camera_params_v1(img_size, device, batch, locations, sweep, uniform, azim_range,
                                                elev_range, fov_ang, dist_radius, up=None)


# ---------------------------------------------------------------------------------------- trajectories
def yaw_trajectory(n_frames, azim_range=(-0.77, 0.77), elev=0.0, fov=6.0):
    """render_video_web_v10.py:1733-1739: azim = a0 + (a1-a0) sin(pi t); columns (azim, elev, fov)."""
    t = torch.linspace(0, 1, n_frames, dtype=torch.float64)
    traj = torch.zeros(n_frames, 3, dtype=torch.float32)
    traj[:, 0] = (azim_range[0] + (azim_range[1] - azim_range[0]) * torch.sin(t * math.pi)).float()
    traj[:, 1] = elev
    traj[:, 2] = fov
    return traj


def circle_trajectory(n_frames, azim_range=0.5, elev=0.0, fov_range=(6.0, 8.0)):
    """render_video_web_v10.py:1764-1775."""
    t = torch.linspace(0, 1, n_frames, dtype=torch.float64)
    traj = torch.zeros(n_frames, 3, dtype=torch.float32)
    traj[:, 0] = (azim_range * torch.sin(t * 2 * math.pi)).float()
    traj[:, 1] = elev
    traj[:, 2] = (fov_range[0] + (fov_range[1] - fov_range[0]) * torch.sin(t * math.pi)).float()
    return traj


def roll_up_vectors(n_frames):
    """render_video_web_v10.py:1625-1633: up = (cos a, sin a, 0), a = 2 pi t + pi/2 (in-plane roll)."""
    t = torch.linspace(0, 1, n_frames)
    a = t * 2 * math.pi + 0.5 * math.pi
    return torch.stack([torch.cos(a), torch.sin(a), torch.zeros(n_frames)], dim=1)


def cameras_from_trajectory(traj, img_size, device, dist_radius=0.12, up=None):
    """(azim, elev, fov) rows -> (extrinsics, focal, near, far) on `device`."""
    traj = traj.to(device)
    e, f, n, fa, _ = Camera.generate_camera_params_v1(img_size, device, locations=traj[:, :2].contiguous(),
                                                      fov_ang=traj[:, 2].contiguous(), dist_radius=dist_radius,
                                                      up=None if up is None else up.to(device))
    return e, f, n, fa


def translate_rotate_cameras(n_frames, trans_max, img_size, device, fov_ang=6.0, dist_radius=0.12):
    """render_video_web_v10.py:1587-1649: first a sideways translation of a frontal camera at z = 1
    (x = trans_max sin(2 pi t), identity rotation), then an in-plane roll of the frontal camera (custom up vectors).
    Returns (extrinsics [2n,3,4], trajectory [2n,3], focal, near, far)."""
    t = torch.linspace(0, 1, n_frames)
    ext_t = torch.zeros(n_frames, 3, 4, device=device)
    ext_t[:, :, :3] = torch.eye(3, device=device)
    ext_t[:, 0, 3] = (trans_max * torch.sin(t * 2 * math.pi)).to(device)
    ext_t[:, 2, 3] = 1
    traj = torch.zeros(n_frames, 3)
    traj[:, 2] = fov_ang
    _, f_t, n_t, fa_t = cameras_from_trajectory(traj, img_size, device, dist_radius)
    ext_r, f_r, n_r, fa_r = cameras_from_trajectory(traj, img_size, device, dist_radius, up=roll_up_vectors(n_frames))
    cat = lambda a, b: torch.cat([a, b], 0)
    return cat(ext_t, ext_r), cat(traj, traj).to(device), cat(f_t, f_r), cat(n_t, n_r), cat(fa_t, fa_r)
