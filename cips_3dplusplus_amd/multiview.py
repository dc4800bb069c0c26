"""View-sharded multi-GPU rendering: one process per GPU, views dealt to ranks, one gather to rank 0.

The generator forward shards over views with no coupling (weights, z, style means and noise buffers
are replicated read-only inputs), so ranks never talk during rendering; the only exchange is the
gather of finished images (reference analogue: rank-interleaved `gen_images`,
/root/reference/exp/cips3d/scripts/gen_images.py:49-84, which has no gather at all; the multi-view demo
itself is single-GPU, render_video_web_v10.py:1806-1824).  `torch.distributed` backend "nccl" is RCCL
on ROCm; the CPU tests drive the same code over gloo.
"""
import torch
import torch.distributed as dist


def view_slice(n_views, rank, world_size):
    """Contiguous block partition: rank r renders views [lo, hi).  Blocks differ by at most one view."""
    base, rem = divmod(n_views, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_views(local, n_views, dst=0, group=None):
    """Gather per-rank view blocks (dim 0) to `dst`.  Returns the full tensor on dst, None elsewhere.
    Ragged blocks are padded to the largest block so a single fixed-size gather is used."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [view_slice(n_views, r, ws) for r in range(ws)]
    cap = max(hi - lo for lo, hi in counts)
    buf = local
    if local.shape[0] < cap:
        pad = torch.zeros((cap - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf = torch.cat([local, pad], 0)
    buf = buf.contiguous()
    outs = [torch.empty_like(buf) for _ in range(ws)] if rank == dst else None
    dist.gather(buf, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, counts)], 0)


class PendingGather:
    """Handle of an asynchronous gather: the collective runs on the communicator's stream while the caller keeps
    rendering; `wait()` orders the current stream after it and (on dst) returns the assembled tensor."""

    def __init__(self, work, outs, counts, keep):
        self.work, self.outs, self.counts, self.keep = work, outs, counts, keep

    def wait(self, assemble=True):
        """assemble=False: only order the current stream after the collective (dst keeps the per-rank blocks in `outs`);
        a steady-state loop that does not look at every frame skips the concatenation copy."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.outs is None or not assemble:
            return None
        return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(self.outs, self.counts)], 0)


def gather_views_async(local, n_views, dst=0, group=None):
    """gather_views without blocking the compute stream (used to overlap the image gather of step i with the
    rendering of step i+1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return PendingGather(None, [local], [(0, local.shape[0])], None)
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [view_slice(n_views, r, ws) for r in range(ws)]
    cap = max(hi - lo for lo, hi in counts)
    buf = local
    if local.shape[0] < cap:
        pad = torch.zeros((cap - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf = torch.cat([local, pad], 0)
    buf = buf.contiguous()
    outs = [torch.empty_like(buf) for _ in range(ws)] if rank == dst else None
    work = dist.gather(buf, outs, dst=dst, group=group, async_op=True)
    return PendingGather(work, outs, counts if rank == dst else None, buf)


def _join(parts):
    """torch.cat(parts, 0) -- without the copy when the parts already are consecutive dim-0 slices of one buffer (a render
    function that writes its frames into a preallocated block)."""
    p0 = parts[0]
    if len(parts) == 1:
        return p0
    if all(p.is_contiguous() and p.dtype == p0.dtype and p.shape[1:] == p0.shape[1:] and
           p.untyped_storage().data_ptr() == p0.untyped_storage().data_ptr() for p in parts):
        off, ok = p0.storage_offset(), True
        for p in parts:
            ok = ok and p.storage_offset() == off
            off += p.numel()
        if ok:
            n = sum(p.shape[0] for p in parts)
            return p0.as_strided((n,) + tuple(p0.shape[1:]), p0.stride(), p0.storage_offset())
    return torch.cat(parts, 0)


def render_views_sharded(render_fn, n_views, keys=("rgb",), dst=0, group=None, chunk=1, finish=None):
    """render_fn(lo, hi) -> dict of tensors whose dim 0 is the view index for views [lo, hi).
    Every rank renders its block in `chunk`-view calls; the entries named in `keys` are gathered to dst.
    finish: called once behind the rank's last render call, before anything reads the results (a ViewPipeline's drain)."""
    ws = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if ws > 1 else 0
    if n_views < ws:
        # decided from (n_views, world size) alone, so EVERY rank raises here: no rank is left waiting in the gather
        raise ValueError(f"{n_views} view(s) for {ws} ranks: a rank would receive no views; use n_views >= world_size")
    lo, hi = view_slice(n_views, rank, ws)
    parts = {k: [] for k in keys}
    for a in range(lo, hi, chunk):
        out = render_fn(a, min(a + chunk, hi))
        for k in keys:
            parts[k].append(out[k])
    if finish is not None:
        finish()
    result = {}
    for k in keys:
        if parts[k]:
            local = _join(parts[k])
        else:
            local = None
        if ws > 1:
            result[k] = gather_views(local, n_views, dst=dst, group=group)
        else:
            result[k] = local
    return result


def sample_multi_view(G, cam_cfg, nerf_cfg, zs, view_mode="yaw", N_frames=8, truncation_ratio=0.5, N_samples=128,
                      zero_noise_bufs=False, noise_bufs=None, azim_range=(-0.77, 0.77), elev=0.0, circle=None,
                      trans_max=0.04, only_rotate=False, chunk=1, gather=("rgb", "thumb_rgb", "xyz"), to_uint8=True,
                      group=None, hoist=True, uint8_in_kernel=True, lanes=2):
    """The frame loop of `_sample_multi_view_web` (render_video_web_v10.py:1651-1899) without the web page: one z pair,
    one set of noise buffers, `perturb=False`, a camera trajectory (`yaw` / `circle` / `translate_rotate`), one
    `G(...)` call per `chunk` frames with `truncation=truncation_ratio, return_xyz=True`.  With torch.distributed
    initialised the frames are dealt to the ranks (contiguous blocks) and gathered to rank 0; `rgb` travels as uint8
    (the `img_tensor_to_pil` step, :1825-1826) unless `to_uint8=False`.  Video encoding / mesh shading stay with the
    caller (`gen_images.xyz_to_mesh` gives the surface of a frame's `xyz`).

    `hoist` (default): the loop renders every frame with the same sample_z, noise_bufs and truncation, so both mapping networks, the
    FiLM table and all modulated / demodulated decoder matrices are the same for the whole sequence: the first call of each call shape
    computes them, every later frame reuses them (`styles_resident=True`: no mapping / style-head / modulate-table launches between
    frames; the reference offers the same hoist through `style_render=` / `style_decoder=`, models/model_v3.py:875-914).
    Bit-identical to `hoist=False`, which recomputes them per frame.
    `uint8_in_kernel` (default): where the decoder ends in a fused up-sampling stage (the 1024^2 recipes) the uint8 frame is written
    by that kernel, straight into the rank's frame block (`rgb_out`); False: fp32 image + `hip.rgb_to_uint8` (same bits).
    `lanes` (default 2): the rank's calls alternate between that many streams (pipeline.ViewPipeline: frames are independent of each
    other, one frame's launch-bound phases run under another's large kernels); the results are ordered behind all of them before
    anything reads them.  With `hoist`, every lane computes the sequence's tables once (its first call).  1: one stream."""
    from . import hip
    from .camera import yaw_trajectory, circle_trajectory, cameras_from_trajectory, translate_rotate_cameras
    dev = next(G.parameters()).device
    img_size = cam_cfg["img_size"]
    fov, dist_radius = cam_cfg.get("fov_ang", 6), cam_cfg.get("dist_radius", 0.12)
    if view_mode == "yaw":
        traj = yaw_trajectory(N_frames, azim_range, elev, fov)
        ext, foc, near, far = cameras_from_trajectory(traj, img_size, dev, dist_radius)
    elif view_mode == "circle":
        c = circle or {}
        traj = circle_trajectory(N_frames, c.get("azim_range", 0.5), c.get("elev_range", 0.0), c.get("fov_range", (6.0, 8.0)))
        ext, foc, near, far = cameras_from_trajectory(traj, img_size, dev, dist_radius)
    elif view_mode == "translate_rotate":
        ext, traj, foc, near, far = translate_rotate_cameras(N_frames, trans_max, img_size, dev, fov, dist_radius)
        if only_rotate:
            ext, traj, foc, near, far = (t.chunk(2)[1] for t in (ext, traj, foc, near, far))
    else:
        raise ValueError(f"view_mode {view_mode!r}")
    n_views = ext.shape[0]
    ncfg = dict(nerf_cfg)
    ncfg["perturb"] = False
    ncfg["N_samples"] = N_samples
    if noise_bufs is None:
        noise_bufs = G.create_noise_bufs(img_size, dev)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            for b in noise_bufs:                   # every rank must draw the same buffers
                dist.broadcast(b, src=0, group=group)
    if zero_noise_bufs:
        noise_bufs = [torch.zeros_like(b) for b in noise_bufs]

    from .pipeline import pipeline_for
    pipe = pipeline_for(G, lanes=lanes if dev.type == "cuda" else 1, device=dev)
    full_done = [set() for _ in range(pipe.lanes)]     # per lane: call shapes (views per call) whose tables this sequence has computed
    ws = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    lo, hi = view_slice(n_views, dist.get_rank(group) if ws > 1 else 0, ws)
    frames_u8 = None                   # this rank's uint8 frames land in ONE block (no concatenation copy at the end)

    def render(a, b):
        nonlocal frames_u8
        direct = None
        static = bool(ncfg.get("static_viewdirs", False))
        if to_uint8 and uint8_in_kernel and G.can_emit_uint8(b - a, img_size, N_samples, static):
            # this rank's frames land in ONE uint8 block; where the decoder ends in a fused up-sampling stage the image leaves
            # that kernel as uint8 straight into its slice of the block (no fp32 image, no conversion launch)
            if frames_u8 is None:
                R = G._forward_plan(b - a, img_size, int(N_samples), static).out_res
                frames_u8 = torch.empty(hi - lo, 3, R, R, dtype=torch.uint8, device=dev)
            direct = frames_u8[a - lo:b - lo]
        lane = pipe.next_lane()
        done = full_done[lane]
        cams = (ext[a:b].contiguous(), foc[a:b].contiguous(), near[a:b].contiguous(), far[a:b].contiguous())

        def one_call():
            nonlocal frames_u8
            with torch.no_grad():
                r = G(zs=zs, cam_poses=cams[0], focals=cams[1], img_size=img_size, near=cams[2], far=cams[3], noise_bufs=noise_bufs,
                      truncation=truncation_ratio, nerf_cfg=ncfg, return_xyz=True,
                      styles_resident=hoist and done == {b - a}, rgb_out=direct)
            if to_uint8 and direct is None:
                r = dict(r)
                if frames_u8 is None:
                    frames_u8 = torch.empty((hi - lo,) + tuple(r["rgb"].shape[1:]), dtype=torch.uint8, device=dev)
                r["rgb"] = hip.rgb_to_uint8(r["rgb"], out=frames_u8[a - lo:b - lo])
            return r

        r = pipe.run(one_call)
        done.clear()                   # (a call of another shape rewrites the lane's style tables: only the last shape is resident)
        done.add(b - a)
        return r

    out = render_views_sharded(render, n_views, keys=gather, group=group, chunk=chunk, finish=pipe.drain)
    out["trajectory"] = traj
    return out
