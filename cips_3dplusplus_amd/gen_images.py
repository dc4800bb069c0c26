"""Rank-sharded bulk generation and depth-map surface extraction.

`gen_images` follows /root/reference/exp/cips3d/scripts/gen_images.py:33-92: every rank renders `batch_gpu`
random views per round (fresh z pair, camera drawn from `cam_cfg`), image `i` of round `b` is saved as
`{b * batch_gpu * world + i * world + rank:05d}.jpg`; there is no exchange step — the file system is the gather.
The images leave the GPU as uint8 (`cips3d_rgb_to_uint8`, the clamp[-1,1] -> [0,255] of `save_image(normalize=True,
value_range=(-1,1))`), 4x fewer bytes over PCIe than the fp32 tensor.

`xyz_to_mesh` follows exp/cips3d/utils.py:228-243 (`xyz2mesh`): vertices = the compositing-weighted surface points
of the `xyz` map, faces = a triangulation of the pixel grid with inverted normals.  The reference triangulates the
(degenerate, co-circular) regular grid with scipy's Delaunay, whose diagonal choice per cell is arbitrary; here every
cell is split along the same diagonal, which is one of the valid Delaunay triangulations of that grid.
"""
import os

import numpy as np
import torch


def mixing_noise(batch, latent_dim, device, n_noise=2, generator=None):
    """exp/cips3d/utils.py:84-105: `n_noise` independent z of shape (batch, latent_dim) from one randn call."""
    return list(torch.randn(n_noise, batch, latent_dim, device=device, generator=generator).unbind(0))


def image_index(round_idx, idx_in_batch, rank, world_size, batch_gpu):
    """gen_images.py:83: global index of image `idx_in_batch` rendered by `rank` in round `round_idx`."""
    return round_idx * batch_gpu * world_size + idx_in_batch * world_size + rank


def n_rounds(num_imgs, world_size, batch_gpu):
    return (num_imgs + batch_gpu * world_size - 1) // (batch_gpu * world_size)


def _save_uint8_chw(img_u8, path):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img_u8.permute(1, 2, 0).cpu().numpy())).save(path, quality=95)


def gen_images(rank, world_size, generator, G_kwargs, fake_dir, num_imgs, batch_gpu, truncation=1, to_uint8=None,
               save_fn=_save_uint8_chw, ext="jpg", barrier=None, seed=None, camera_fn=None):
    """Returns the list of files this rank wrote.  `G_kwargs` = {"cam_cfg": {...img_size...}, "nerf_cfg": {...}}
    (train_cips3d_ffhq_v10.yaml:129-140).  `to_uint8` defaults to the HIP image post-step."""
    if camera_fn is None:
        from .camera import Camera
        camera_fn = Camera.generate_camera_params
    if to_uint8 is None:
        from .hip import rgb_to_uint8 as to_uint8
    if rank == 0:
        os.makedirs(fake_dir, exist_ok=True)
    if barrier is not None:
        barrier()
    cam_cfg = dict(G_kwargs["cam_cfg"])
    nerf_cfg = dict(G_kwargs["nerf_cfg"])
    img_size = cam_cfg.pop("img_size")
    device = next(generator.parameters()).device
    gen = None
    if seed is not None:
        gen = torch.Generator(device=device).manual_seed(int(seed) + rank)
    written = []
    generator.eval()
    with torch.no_grad():
        for b in range(n_rounds(num_imgs, world_size, batch_gpu)):
            zs = mixing_noise(batch_gpu, generator.z_dim, device, generator=gen)
            cam, focal, near, far, _ = camera_fn(img_size, device, batch=batch_gpu, **cam_cfg)
            ret = generator(zs=zs, cam_poses=cam, focals=focal, img_size=img_size, near=near, far=far,
                            truncation=truncation, nerf_cfg=nerf_cfg)
            imgs = to_uint8(ret["rgb"])
            for i in range(imgs.shape[0]):
                idx = image_index(b, i, rank, world_size, batch_gpu)
                path = os.path.join(fake_dir, f"{idx:0>5}.{ext}")
                save_fn(imgs[i], path)
                written.append(path)
    if barrier is not None:
        barrier()
    return written


def grid_faces(h, w):
    """Two triangles per pixel cell, vertex index = row * w + col.  Winding as in the reference after its
    "invert normals" swap: scipy's simplices are counter-clockwise in (col, row), the world frame has y = -row, so
    after the swap the normals of a fronto-parallel depth map point at the camera (+z for the frontal view)."""
    r, c = np.meshgrid(np.arange(h - 1), np.arange(w - 1), indexing="ij")
    v00 = (r * w + c).reshape(-1)
    v01 = v00 + 1
    v10 = v00 + w
    v11 = v10 + 1
    return np.concatenate([np.stack([v00, v10, v01], 1), np.stack([v01, v10, v11], 1)], 0).astype(np.int64)


def xyz_to_mesh(xyz):
    """`xyz` (1,3,h,w) -> (vertices [h*w,3] float32, faces [2*(h-1)*(w-1),3] int64)."""
    if xyz.dim() != 4 or xyz.shape[0] != 1 or xyz.shape[1] != 3:
        raise ValueError("xyz must be (1,3,h,w)")
    _, _, h, w = xyz.shape
    verts = xyz[0].permute(1, 2, 0).reshape(h * w, 3).detach().float().cpu().numpy()
    return verts, grid_faces(h, w)


def vertex_normals(verts, faces):
    """Area-weighted vertex normals (what trimesh's `vertex_normals` provides to the mesh renderer,
    render_video_web_v10.py:1843-1850)."""
    tri = verts[faces]
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    vn = np.zeros_like(verts)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    n = np.linalg.norm(vn, axis=1, keepdims=True)
    return vn / np.maximum(n, 1e-20)


def write_obj(path, verts, faces):
    with open(path, "w") as f:
        for v in verts:
            f.write(f"v {v[0]:.7g} {v[1]:.7g} {v[2]:.7g}\n")
        for t in faces + 1:
            f.write(f"f {t[0]} {t[1]} {t[2]}\n")
    return path
