"""Checkpoint / inversion-file formats and rank-sharded bulk generation: CPU side.

The fixture `tests/golden/ckpt_tiny/` was written by `tests/golden/make_golden.py` from the imported reference:
`G_ema.pth` = `torch.save(reference_generator.state_dict())`, `w.pth` = the dict of projector_v10.py:1044-1055 built
from the reference generator's own sub-module state dicts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CKPT = os.path.join(ROOT, "tests", "golden", "ckpt_tiny")


def test_load_reference_written_checkpoint(golden):
    from cips_3dplusplus_amd import checkpoint
    G, cfg = checkpoint.load_generator(CKPT, device="cpu")
    assert cfg["G_cfg"]["name"] == "exp.cips3d.models.model_v3.Generator"
    assert cfg["G_kwargs"]["nerf_cfg"]["N_samples"] == 24
    fx = golden("tiny_generator")
    sd = G.state_dict()
    ref_sd = fx.sub("h32_d2.sd.")
    assert sorted(ref_sd) == sorted(sd.keys())
    for k, v in ref_sd.items():
        assert torch.equal(sd[k], v), k


def test_checkpoint_round_trip(tmp_path):
    from cips_3dplusplus_amd import checkpoint, configs, build_generator
    cfg = configs.tiny_G_cfg(32, 2, 1)
    G = build_generator(cfg, device="cpu", seed=3)
    d = checkpoint.save_generator(G, str(tmp_path / "ck"), cfg, G_kwargs={"cam_cfg": configs.FFHQ_CAM_CFG,
                                                                          "nerf_cfg": configs.TRAIN_NERF_CFG})
    G2, loaded = checkpoint.load_generator(d, device="cpu")
    assert checkpoint.generator_ctor_cfg(loaded["G_cfg"]) == cfg
    a, b = G.state_dict(), G2.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_inversion_file(tmp_path):
    from cips_3dplusplus_amd import checkpoint
    G, _ = checkpoint.load_generator(CKPT, device="cpu")
    G_n_latent, n_noise = G.decoder.n_latent, G.decoder.num_layers      # tiny decoder: 2 stages
    az, el, w_r, w_d, dec_sd, noise, ren_sd = checkpoint.load_inversion(os.path.join(CKPT, "w.pth"), w_idx=1)
    assert (round(az, 4), round(el, 4)) == (-0.25, 0.05)
    assert tuple(w_r.shape) == (1, 3, 32) and tuple(w_d.shape) == (1, G_n_latent, 32)
    assert float(w_d[0, 0, 0]) == 1.0          # row 1 of w_decoder_opt (the flip), row 0 of w_render_opt
    assert float(w_r[0, 0, 0]) == 0.0
    assert len(noise) == n_noise and not noise[0].requires_grad
    for p in G.decoder.parameters():
        p.data.zero_()
    checkpoint.apply_inversion(G, dec_sd, ren_sd)      # key names of the sub-modules are the contract
    assert any(float(p.detach().abs().sum()) > 0 for p in G.decoder.parameters())
    # write -> read
    path = checkpoint.save_inversion(str(tmp_path / "w2.pth"), torch.tensor([0.1, -0.1]), torch.tensor([0.0, 0.0]),
                                     w_r, torch.cat([w_d, w_d]), G, noise_bufs=noise, padding=0)
    again = checkpoint.load_inversion(path, 0)
    assert again[0] == pytest.approx(0.1) and set(again[4]) == set(dec_sd)
    import torch as _t
    assert set(_t.load(path, weights_only=True)) == set(checkpoint.INVERSION_KEYS)


def test_stage_blend():
    from cips_3dplusplus_amd import checkpoint
    G, _ = checkpoint.load_generator(CKPT, device="cpu")
    inv = {k: torch.zeros_like(v) for k, v in G.decoder.state_dict().items()}
    keys = checkpoint.stage_keys(inv, stages=(1,), conv_in=False)
    assert keys and all(k.startswith(("convs.2.", "convs.3.", "to_rgbs.1.")) for k in keys)
    checkpoint.blend_decoder_stages(inv, G.decoder, decay=0.25, stages=(1,))
    src = G.decoder.state_dict()
    k = "convs.2.conv.weight"
    assert torch.allclose(inv[k], 0.75 * src[k])
    assert float(inv["convs.0.conv.weight"].abs().sum()) == 0.0


def test_image_index_partition():
    from cips_3dplusplus_amd.gen_images import image_index, n_rounds
    for world, batch, num in ((1, 4, 10), (2, 2, 9), (8, 4, 100)):
        idx = sorted(image_index(b, i, r, world, batch) for b in range(n_rounds(num, world, batch))
                     for i in range(batch) for r in range(world))
        assert idx == list(range(len(idx))) and len(idx) >= num     # a permutation covering [0, num)


def test_mesh_from_xyz():
    from cips_3dplusplus_amd.gen_images import xyz_to_mesh, vertex_normals, write_obj
    h = w = 5
    r, c = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    # fronto-parallel surface in the world frame of the frontal camera: x = col, y = -row, z = const
    xyz = torch.tensor(np.stack([c, -r, np.full_like(c, 2)]), dtype=torch.float32)[None]
    v, f = xyz_to_mesh(xyz)
    assert v.shape == (25, 3) and f.shape == (2 * 4 * 4, 3)
    assert np.allclose(vertex_normals(v, f), [0, 0, 1])            # towards the camera
    assert len(set(map(tuple, np.sort(f, 1)))) == len(f)           # no duplicate faces
    # every cell covered once: total area = (h-1)*(w-1)
    tri = v[f]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1).sum()
    assert area == pytest.approx(16.0)


# ---- gen_images over gloo, world size 2 (stub generator: the HIP path needs a GPU) ---------------------------
class _StubG(torch.nn.Module):
    z_dim = 8

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.calls = 0

    def forward(self, zs, cam_poses, focals, img_size, near, far, truncation, nerf_cfg):
        assert len(zs) == 2 and zs[0].shape == (cam_poses.shape[0], 8) and cam_poses.shape[1:] == (3, 4)
        self.calls += 1
        return {"rgb": torch.zeros(cam_poses.shape[0], 3, 4, 4)}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cips_3dplusplus_amd.gen_images import gen_images
    from cips_3dplusplus_amd import configs

    def to_u8(x):
        return ((x.clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8)

    def cam_fn(img_size, device, batch, **cam_cfg):       # the HIP camera kernel needs a GPU
        assert img_size == 64 and cam_cfg["fov_ang"] == 6
        one = torch.ones(batch, 1, 1)
        return torch.zeros(batch, 3, 4), one, one, one, torch.zeros(batch, 2)

    G = _StubG()
    files = gen_images(rank, world, G, {"cam_cfg": configs.FFHQ_CAM_CFG, "nerf_cfg": configs.TRAIN_NERF_CFG}, out_dir,
                       num_imgs=10, batch_gpu=2, to_uint8=to_u8, barrier=dist.barrier, seed=1, camera_fn=cam_fn)
    q.put((rank, [os.path.basename(f) for f in files], G.calls))
    dist.barrier(); dist.destroy_process_group()


def test_gen_images_two_ranks(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port(); out = str(tmp_path / "fake")
    ps = [ctx.Process(target=_worker, args=(r, 2, port, out, q)) for r in range(2)]
    for p in ps: p.start()
    res = dict((r, (f, c)) for r, f, c in (q.get(timeout=90) for _ in range(2)))
    for p in ps:
        p.join(timeout=60); assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 3                     # ceil(10 / (2*2)) rounds each
    names = sorted(res[0][0] + res[1][0])
    assert names == [f"{i:05d}.jpg" for i in range(12)]
    assert all(int(n[:5]) % 2 == 0 for n in res[0][0]) and all(int(n[:5]) % 2 == 1 for n in res[1][0])
    assert sorted(os.listdir(out)) == names


# ---- GPU: the loaded formats drive the HIP path ----------------------------------------------------------------
@pytest.mark.gpu
def test_reference_checkpoint_runs_on_gpu(golden):
    """config_command.yaml + G_ema.pth written from the reference -> HIP forward == the reference's own outputs."""
    from cips_3dplusplus_amd import checkpoint
    from cips_3dplusplus_amd.camera import Camera
    fx, tag = golden("tiny_generator"), "h32_d2"
    G, cfg = checkpoint.load_generator(CKPT, device="cuda")
    cu = lambda t: t.cuda()
    zs = [cu(fx[f"{tag}.z0"]), cu(fx[f"{tag}.z1"])]
    e, f, n, fa, _ = Camera.generate_camera_params(8, "cuda", locations=cu(fx[f"{tag}.locs"]))
    nb = [cu(fx[f"{tag}.noise{i}"]) for i in range(G.decoder.num_layers)]
    r = G(zs=zs, cam_poses=e, focals=f, img_size=8, near=n, far=fa, noise_bufs=nb, truncation=1.0,
          nerf_cfg=dict(N_samples=6, perturb=False, static_viewdirs=False), return_xyz=True)
    for k in ("rgb", "thumb_rgb", "xyz", "mask", "depth"):
        assert float((r[k].cpu() - fx[f"{tag}.a.{k}"]).abs().max()) < 1e-4, k


@pytest.mark.gpu
def test_inversion_file_drives_forward(golden):
    """w.pth -> (azim, elev, W+ styles, decoder/renderer state, noise) -> HIP forward vs the oracle."""
    from cips_3dplusplus_amd import checkpoint, configs
    from cips_3dplusplus_amd.camera import Camera
    from oracle import path as O
    G, cfg = checkpoint.load_generator(CKPT, device="cuda")
    az, el, w_r, w_d, dec_sd, noise, ren_sd = checkpoint.load_inversion(os.path.join(CKPT, "w.pth"), w_idx=1)
    checkpoint.blend_decoder_stages(dec_sd, G.decoder, decay=0.7, stages=(1,))
    checkpoint.apply_inversion(G, dec_sd, ren_sd)
    g = torch.Generator().manual_seed(3)
    w_r = w_r + 0.3 * torch.randn(w_r.shape, generator=g)
    w_d = w_d + 0.3 * torch.randn(w_d.shape, generator=g)
    loc = torch.tensor([[az, el]])
    e, f, n, fa, _ = Camera.generate_camera_params(8, "cuda", locations=loc.cuda())
    ncfg = dict(N_samples=6, perturb=False, static_viewdirs=True)
    r = G(zs=[None, None], cam_poses=e, focals=f, img_size=8, near=n, far=fa, noise_bufs=[b.cuda() for b in noise],
          style_render=w_r.cuda(), style_decoder=w_d.cuda(), nerf_cfg=ncfg, return_xyz=True)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    ctor = checkpoint.generator_ctor_cfg(cfg["G_cfg"])
    cam = O.camera_params(loc, 8, 6, 0.12)
    ref = O.generator_forward(sd, ctor, [None, None], cam[0], cam[1], 8, cam[2], cam[3], ncfg, noise,
                              style_render=w_r, style_decoder=w_d, return_xyz=True)
    for k in ("rgb", "thumb_rgb", "xyz", "mask", "depth"):
        assert float((r[k].cpu() - ref[k]).abs().max()) < 1e-4, k


@pytest.mark.gpu
def test_gen_images_on_gpu(tmp_path):
    from PIL import Image
    from cips_3dplusplus_amd import checkpoint, configs, hip
    from cips_3dplusplus_amd.camera import Camera
    from cips_3dplusplus_amd.gen_images import gen_images, mixing_noise
    G, cfg = checkpoint.load_generator(CKPT, device="cuda")
    kw = {"cam_cfg": {**cfg["G_kwargs"]["cam_cfg"], "img_size": 8}, "nerf_cfg": {"N_samples": 6, "perturb": False,
                                                                                 "static_viewdirs": False}}
    nb = G.create_noise_bufs(8, "cuda")
    fwd = G.forward
    G.forward = lambda **k: fwd(noise_bufs=nb, **k)          # fixed decoder noise so the run can be repeated
    torch.manual_seed(4)
    files = gen_images(0, 1, G, kw, str(tmp_path / "fake"), num_imgs=3, batch_gpu=2, ext="png", seed=9)
    assert [os.path.basename(f) for f in files] == [f"{i:05d}.png" for i in range(4)]
    # replay round 0 by hand
    torch.manual_seed(4)
    gen = torch.Generator(device="cuda").manual_seed(9)
    zs = mixing_noise(2, G.z_dim, "cuda", generator=gen)
    cam_cfg = dict(kw["cam_cfg"]); cam_cfg.pop("img_size")
    e, f, n, fa, _ = Camera.generate_camera_params(8, "cuda", batch=2, **cam_cfg)
    r = fwd(zs=zs, cam_poses=e, focals=f, img_size=8, near=n, far=fa, truncation=1, nerf_cfg=kw["nerf_cfg"],
            noise_bufs=nb)
    u8 = hip.rgb_to_uint8(r["rgb"]).cpu()
    for i in range(2):
        img = torch.from_numpy(np.array(Image.open(files[i]))).permute(2, 0, 1)
        assert img.shape == u8[i].shape and torch.equal(img, u8[i])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,n", [("yaw", 5), ("circle", 4), ("translate_rotate", 3)])
def test_sample_multi_view_single_gpu(mode, n):
    """Frame loop of _sample_multi_view_web vs frame-by-frame oracle renders (same z, noise, truncation means)."""
    from cips_3dplusplus_amd import checkpoint, hip
    from cips_3dplusplus_amd.multiview import sample_multi_view
    from oracle import path as O
    G, cfg = checkpoint.load_generator(CKPT, device="cuda")
    ctor = checkpoint.generator_ctor_cfg(cfg["G_cfg"])
    g = torch.Generator().manual_seed(8)
    zs = [torch.randn(1, 32, generator=g), torch.randn(1, 32, generator=g)]
    mr, md = 0.2 * torch.randn(1, 32, generator=g), 0.2 * torch.randn(1, 32, generator=g)
    G.style_render_mean, G.style_decoder_mean = mr.cuda(), md.cuda()
    nb = [torch.randn(*b.shape, generator=g) for b in G.create_noise_bufs(8, "cpu")]
    cam_cfg = {"img_size": 8, "fov_ang": 6, "dist_radius": 0.12}
    out = sample_multi_view(G, cam_cfg, {"N_samples": 24, "static_viewdirs": True}, [z.cuda() for z in zs], view_mode=mode,
                            N_frames=n, truncation_ratio=0.5, N_samples=6, noise_bufs=[b.cuda() for b in nb], to_uint8=False)
    n_views = 2 * n if mode == "translate_rotate" else n
    assert out["rgb"].shape[0] == n_views and out["trajectory"].shape == (n_views, 3)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    traj = out["trajectory"].cpu()
    ups = None
    if mode == "translate_rotate":
        from cips_3dplusplus_amd.camera import roll_up_vectors
        ups = roll_up_vectors(n)
    for v in range(n_views):
        if mode == "translate_rotate" and v < n:
            import math
            ext = torch.zeros(1, 3, 4); ext[0, :, :3] = torch.eye(3)
            ext[0, 0, 3] = 0.04 * math.sin(2 * math.pi * v / (n - 1)); ext[0, 2, 3] = 1
            cam = (ext,) + tuple(O.camera_params(torch.zeros(1, 2), 8, 6, 0.12)[1:4])
        else:
            up = None if ups is None else ups[v - n: v - n + 1]
            cam = O.camera_params(traj[v:v + 1, :2], 8, float(traj[v, 2]), 0.12, up=up)
        ref = O.generator_forward(sd, ctor, zs, cam[0], cam[1], 8, cam[2], cam[3],
                                  dict(N_samples=6, perturb=False, static_viewdirs=True), nb, truncation=0.5,
                                  style_render_mean=mr, style_decoder_mean=md, return_xyz=True)
        for k in ("rgb", "thumb_rgb", "xyz"):
            assert float((out[k][v].cpu() - ref[k][0]).abs().max()) < 2e-4, (mode, v, k)
    u8 = sample_multi_view(G, cam_cfg, {"N_samples": 24, "static_viewdirs": True}, [z.cuda() for z in zs], view_mode=mode,
                           N_frames=n, truncation_ratio=0.5, N_samples=6, noise_bufs=[b.cuda() for b in nb])
    assert u8["rgb"].dtype == torch.uint8 and torch.equal(u8["rgb"], hip.rgb_to_uint8(out["rgb"]))


@pytest.mark.gpu
@pytest.mark.parametrize("res,n_samples,chunk", [(256, 12, 1), (256, 12, 3), (1024, 24, 1)])
def test_sample_multi_view_hoisted_tables_are_bit_identical(res, n_samples, chunk):
    """The sequence plan of config 4: frames after the first reuse the plan's style tables
    (`styles_resident`: no mapping network, style head or modulate-table launch) -- every output must carry the bits of the
    per-frame recomputation (`hoist=False`), at chunk 1 and with several frames per call, in the release recipe's split-fp16
    arithmetic whose range rows are zeroed per frame."""
    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs
    from cips_3dplusplus_amd.multiview import sample_multi_view
    G = pkg.build_generator(configs.ffhq_G_cfg(res, 2), "cuda", seed=3)
    g = torch.Generator(device="cuda").manual_seed(11)
    zs = [torch.randn(1, 256, device="cuda", generator=g), torch.randn(1, 256, device="cuda", generator=g)]
    nb = [torch.randn(b.shape, device="cuda", generator=g) for b in G.create_noise_bufs(64, "cuda")]
    G.style_render_mean = 0.1 * torch.randn(1, 256, device="cuda", generator=g)
    G.style_decoder_mean = 0.1 * torch.randn(1, 512, device="cuda", generator=g)
    cam_cfg = {"img_size": 64, "fov_ang": 12, "dist_radius": 0.12}
    kw = dict(view_mode="yaw", N_frames=5, truncation_ratio=0.7, N_samples=n_samples, noise_bufs=nb, to_uint8=False, chunk=chunk)
    a = sample_multi_view(G, cam_cfg, {"static_viewdirs": False}, zs, hoist=False, **kw)
    b = sample_multi_view(G, cam_cfg, {"static_viewdirs": False}, zs, hoist=True, **kw)
    for k in ("rgb", "thumb_rgb", "xyz"):
        assert a[k].shape[0] == 5 and torch.equal(a[k], b[k]), k
    assert not torch.equal(a["rgb"][0], a["rgb"][1])


@pytest.mark.gpu
def test_styles_resident_refuses_a_stale_plan():
    """`styles_resident=True` is a promise the plan checks: other latents, another truncation, a modified latent tensor, or any
    rewrite of the modules' style tables since the plan's last full forward (another plan of the same batch size, the per-op
    renderer) must raise instead of rendering with the wrong styles."""
    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs
    from cips_3dplusplus_amd.camera import Camera
    G = pkg.build_generator(configs.ffhq_G_cfg(256, 2), "cuda", seed=3)
    e, f, n, fa, _ = Camera.generate_camera_params(64, "cuda", locations=torch.zeros(1, 2, device="cuda"))
    zs = [torch.randn(1, 256, device="cuda"), torch.randn(1, 256, device="cuda")]
    nb = G.create_noise_bufs(64, "cuda")
    kw = dict(cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=nb, nerf_cfg=dict(N_samples=8, perturb=False))
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="styles_resident"):
            G(zs=zs, styles_resident=True, **kw)                  # no full forward yet
        G(zs=zs, **kw)                                           # a call outside any sequence (styles_resident=None) records nothing
        with pytest.raises(RuntimeError, match="styles_resident"):
            G(zs=zs, styles_resident=True, **kw)
        full = G(zs=zs, styles_resident=False, **kw)["rgb"]      # the first (full) frame of a sequence
        assert torch.equal(G(zs=zs, styles_resident=True, **kw)["rgb"], full)
        with pytest.raises(RuntimeError, match="styles_resident"):
            G(zs=[zs[0].clone(), zs[1]], styles_resident=True, **kw)     # other latents
        with pytest.raises(RuntimeError, match="styles_resident"):
            G(zs=zs, truncation=0.9, styles_resident=True, **kw)         # another truncation
        zs[0].mul_(1.0)                                           # an in-place write bumps the version
        with pytest.raises(RuntimeError, match="styles_resident"):
            G(zs=zs, styles_resident=True, **kw)
        full = G(zs=zs, styles_resident=False, **kw)["rgb"]
        kw2 = dict(kw, nerf_cfg=dict(N_samples=12, perturb=False))
        G(zs=[z.clone() for z in zs], **kw2)                      # another plan of the same batch size rewrote the shared tables
        with pytest.raises(RuntimeError, match="styles_resident"):
            G(zs=zs, styles_resident=True, **kw)
        assert torch.equal(G(zs=zs, styles_resident=False, **kw)["rgb"], full)
        assert torch.equal(G(zs=zs, styles_resident=True, **kw)["rgb"], full)
        G.decoder.conv1.conv.weight.mul_(1.0)                     # an optimiser step on the decoder: the modulated weights are stale
        with pytest.raises(RuntimeError, match="styles_resident"):
            G(zs=zs, styles_resident=True, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("res,B,flat", [(1024, 1, True), (1024, 2, True), (256, 3, True), (256, 3, False)])
def test_uint8_image_straight_from_the_last_stage(monkeypatch, res, B, flat):
    """`Generator.forward(rgb_out=<uint8 tensor>)` (cips3d_forward_io.rgb_is_u8 / CIPS3D_RGB_U8): the last fused stage stores
    the image as uint8 -- exactly hip.rgb_to_uint8 of the fp32 image the same forward would have written, every other output
    unchanged; a float32 `rgb_out` receives the fp32 image in place."""
    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs, hip, plan as planmod
    from cips_3dplusplus_amd.camera import Camera
    monkeypatch.setattr(planmod, "FLAT_STAGES", flat)
    G = pkg.build_generator(configs.ffhq_G_cfg(res, 2), "cuda", seed=5)
    g = torch.Generator(device="cuda").manual_seed(3)
    zs = [2.0 * torch.randn(B, 256, device="cuda", generator=g), 2.0 * torch.randn(B, 256, device="cuda", generator=g)]
    e, f, n, fa, _ = Camera.generate_camera_params(64, "cuda", locations=0.3 * torch.randn(B, 2, device="cuda", generator=g))
    nb = [torch.randn(b.shape, device="cuda", generator=g) for b in G.create_noise_bufs(64, "cuda")]
    kw = dict(zs=zs, cam_poses=e, focals=f, img_size=64, near=n, far=fa, noise_bufs=nb, nerf_cfg=dict(N_samples=12, perturb=False),
              return_xyz=True)
    # (the 1024^2 recipe ends in an up-sampling stage, the 256^2 one in the 512 / 1024 blocks at 256^2: flat stages of the same
    # kernel -- or, with CIPS3D_FLAT_STAGES=0, two plain StyledConv launches + a ToRGB launch: fp32 output only)
    capable = res == 1024 or flat
    assert G.can_emit_uint8(B, 64, 12) == capable
    with torch.no_grad():
        ref = G(**kw)
        f32 = torch.full((B, 3, res, res), float("nan"), device="cuda")
        out32 = G(rgb_out=f32, **kw)
        assert out32["rgb"] is f32 and torch.equal(f32, ref["rgb"])
        u8 = torch.full((B, 3, res, res), 77, dtype=torch.uint8, device="cuda")
        if not capable:
            with pytest.raises(RuntimeError, match="uint8"):
                G(rgb_out=u8, **kw)
            return
        out = G(rgb_out=u8, **kw)
    assert out["rgb"] is u8
    exp = hip.rgb_to_uint8(ref["rgb"])
    assert torch.equal(u8, exp)
    assert int(exp.min()) < 40 and int(exp.max()) > 215            # (the clamp and both ends of the range are exercised)
    for k in ("thumb_rgb", "xyz", "mask", "depth"):
        assert torch.equal(out[k], ref[k]), k
    with pytest.raises(RuntimeError, match="rgb_out"):
        G(rgb_out=torch.empty(B, 3, res, res + 1, dtype=torch.uint8, device="cuda"), **kw)


@pytest.mark.gpu
def test_lane_streams_are_served_by_different_hardware_queues():
    """HIP maps streams onto a few hardware queues; two lanes on one queue run one after the other (measured: slower than one
    stream).  pipeline.lane_streams hands out streams that demonstrably overtake each other, once per device, to every pipeline."""
    import itertools
    from cips_3dplusplus_amd import pipeline
    dev = torch.device("cuda:0")
    S = pipeline.lane_streams(dev, 3)
    assert len({s.cuda_stream for s in S}) == 3
    assert not pipeline._overtakes(S[0], S[0], dev)                     # the race itself: a stream does not overtake itself
    for a, b in itertools.permutations(S, 2):
        assert pipeline._overtakes(a, b, dev)
    lin = torch.nn.Linear(2, 2).to(dev)
    p1, p2 = pipeline.ViewPipeline(lin, 2), pipeline.ViewPipeline(lin, 2)
    assert [s.cuda_stream for s in p1.streams] == [s.cuda_stream for s in p2.streams] == [s.cuda_stream for s in S[:2]]


@pytest.mark.gpu
def test_views_in_flight_on_two_streams_equal_one_stream():
    """pipeline.ViewPipeline: independent views issued alternately on two streams (each stream = a lane with its own forward plans
    and style tables) come out bit-identical to the same calls on one stream -- with different latents and cameras per view, fixed
    noise, more views than lanes (every lane's workspaces are reused), and a resident-styles sequence per lane."""
    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs
    from cips_3dplusplus_amd.camera import Camera
    from cips_3dplusplus_amd.pipeline import ViewPipeline
    dev = "cuda"
    G = pkg.build_generator(configs.ffhq_G_cfg(256, 2), dev, seed=3)
    g = torch.Generator(device=dev).manual_seed(5)
    n = 7
    zs = [[torch.randn(1, 256, device=dev, generator=g), torch.randn(1, 256, device=dev, generator=g)] for _ in range(n)]
    locs = torch.randn(n, 2, device=dev, generator=g) * 0.3
    nb = [torch.randn(b.shape, device=dev, generator=g) for b in G.create_noise_bufs(64, dev)]
    cams = [Camera.generate_camera_params(64, dev, locations=locs[i:i + 1]) for i in range(n)]
    kws = [dict(zs=zs[i], cam_poses=cams[i][0], focals=cams[i][1], img_size=64, near=cams[i][2], far=cams[i][3], noise_bufs=nb,
                nerf_cfg=dict(N_samples=12, perturb=False, static_viewdirs=False), return_xyz=True) for i in range(n)]
    with torch.no_grad():
        ref = [{k: v.clone() for k, v in G(**kw).items() if torch.is_tensor(v)} for kw in kws]
        pipe = ViewPipeline(G, lanes=2)
        assert pipe.lanes == 2
        outs = [pipe.submit(**kw) for kw in kws]
        pipe.drain()
        torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ref, outs)):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k)
    assert len(G.__dict__["_stream_lanes"]) == 3 and set(G.__dict__["_lane_plans"]) == {1, 2}      # the test's stream + two lanes
    # a sequence of one latent on each lane: the lane's own first call computes its tables, later ones reuse them
    with torch.no_grad():
        seq = []
        for i in range(6):
            lane = pipe.next_lane()
            seq.append(pipe.submit(**{**kws[0], "cam_poses": cams[i][0], "styles_resident": i >= 2}))
        pipe.drain()
        want = [G(**{**kws[0], "cam_poses": cams[i][0]})["rgb"].clone() for i in range(6)]
    for i in range(6):
        assert torch.equal(seq[i]["rgb"], want[i]), i


@pytest.mark.gpu
@pytest.mark.parametrize("hoist", [False, True])
def test_sample_multi_view_lanes_are_bit_identical(hoist):
    import cips_3dplusplus_amd as pkg
    from cips_3dplusplus_amd import configs
    from cips_3dplusplus_amd.multiview import sample_multi_view
    dev = "cuda"
    G = pkg.build_generator(configs.ffhq_G_cfg(256, 2), dev, seed=4)
    g = torch.Generator(device=dev).manual_seed(9)
    zs = [torch.randn(1, 256, device=dev, generator=g), torch.randn(1, 256, device=dev, generator=g)]
    cam_cfg = {"img_size": 64, "fov_ang": 6, "dist_radius": 0.12}
    nb = G.create_noise_bufs(64, dev)
    kw = dict(view_mode="yaw", N_frames=7, truncation_ratio=0.5, N_samples=16, noise_bufs=nb, hoist=hoist)
    a = sample_multi_view(G, cam_cfg, {"static_viewdirs": False}, zs, lanes=1, **kw)
    b = sample_multi_view(G, cam_cfg, {"static_viewdirs": False}, zs, lanes=2, **kw)
    c = sample_multi_view(G, cam_cfg, {"static_viewdirs": False}, zs, lanes=2, **kw)      # (the lanes' plans and tables are reused)
    torch.cuda.synchronize()
    for k in ("rgb", "thumb_rgb", "xyz"):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
