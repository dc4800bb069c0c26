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

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.outs is None:
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


def render_views_sharded(render_fn, n_views, keys=("rgb",), dst=0, group=None, chunk=1):
    """render_fn(lo, hi) -> dict of tensors whose dim 0 is the view index for views [lo, hi).
    Every rank renders its block in `chunk`-view calls; the entries named in `keys` are gathered to dst."""
    ws = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if ws > 1 else 0
    lo, hi = view_slice(n_views, rank, ws)
    parts = {k: [] for k in keys}
    for a in range(lo, hi, chunk):
        out = render_fn(a, min(a + chunk, hi))
        for k in keys:
            parts[k].append(out[k])
    result = {}
    for k in keys:
        if parts[k]:
            local = torch.cat(parts[k], 0)
        else:
            local = None
        if ws > 1:
            shape_src = local
            if shape_src is None:
                raise RuntimeError("a rank received no views; use n_views >= world_size")
            result[k] = gather_views(local, n_views, dst=dst, group=group)
        else:
            result[k] = local
    return result
