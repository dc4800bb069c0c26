"""View sharding + gather over torch.distributed with the gloo backend (CPU, world_size 2 and 3).

The render function is a stand-in (the HIP path needs a GPU); what is under test is the N > 1 host logic
that bench.py and multi-view rendering use on RCCL: the partition, the ragged gather and rank-0 assembly."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_views, chunk, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cips_3dplusplus_amd.multiview import gather_views, gather_views_async, render_views_sharded, view_slice
    lo, hi = view_slice(n_views, rank, world)

    def render(a, b):
        # "image" of view v is filled with v; a second map checks multi-key gathering
        v = torch.arange(a, b, dtype=torch.float32).view(-1, 1, 1, 1)
        return {"rgb": v.expand(b - a, 3, 4, 4).contiguous(), "thumb_rgb": (v * 10).expand(b - a, 3, 2, 2).contiguous()}

    out = render_views_sharded(render, n_views, keys=("rgb", "thumb_rgb"), chunk=chunk)
    ok = True
    if rank == 0:
        ok &= out["rgb"].shape == (n_views, 3, 4, 4)
        ok &= bool((out["rgb"][:, 0, 0, 0] == torch.arange(n_views, dtype=torch.float32)).all())
        ok &= bool((out["thumb_rgb"][:, 0, 0, 0] == 10 * torch.arange(n_views, dtype=torch.float32)).all())
    else:
        ok &= out["rgb"] is None and out["thumb_rgb"] is None
    # plain gather of a per-rank block
    g = gather_views(torch.full((hi - lo, 2), float(rank)), n_views)
    if rank == 0:
        exp = torch.cat([torch.full((view_slice(n_views, r, world)[1] - view_slice(n_views, r, world)[0], 2), float(r))
                         for r in range(world)])
        ok &= bool((g == exp).all())
    pg = gather_views_async(torch.full((hi - lo, 3), float(rank + 1)), n_views)
    ga = pg.wait()
    if rank == 0:
        ok &= ga.shape[0] == n_views and float(ga[0, 0]) == 1.0 and float(ga[-1, 0]) == float(world)
    else:
        ok &= ga is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _worker_too_few(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cips_3dplusplus_amd.multiview import render_views_sharded
    try:
        render_views_sharded(lambda a, b: {"rgb": torch.zeros(b - a, 1)}, world - 1)
        q.put((rank, False))
    except ValueError:
        q.put((rank, True))
    dist.barrier()                     # every rank raised: nobody is stuck in a gather
    dist.destroy_process_group()


def test_fewer_views_than_ranks_raises_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_too_few, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


@pytest.mark.parametrize("world,n_views,chunk", [(2, 8, 1), (2, 7, 2), (3, 8, 3)])
def test_sharded_render_and_gather(world, n_views, chunk):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_view_slice_partition():
    sys.path.insert(0, ROOT)
    from cips_3dplusplus_amd.multiview import view_slice
    for n in (1, 7, 8, 111):
        for w in (1, 2, 3, 8):
            sl = [view_slice(n, r, w) for r in range(w)]
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            sizes = [hi - lo for lo, hi in sl]
            assert max(sizes) - min(sizes) <= 1


def test_join_reuses_a_block_of_consecutive_slices():
    """multiview._join: consecutive dim-0 slices of one buffer come back as a view of it (the frame block sample_multi_view
    fills), anything else is concatenated."""
    import torch
    from cips_3dplusplus_amd.multiview import _join
    buf = torch.arange(24.).reshape(6, 4)
    j = _join([buf[0:2], buf[2:3], buf[3:6]])
    assert j.data_ptr() == buf.data_ptr() and torch.equal(j, buf)
    j = _join([buf[1:3], buf[3:4]])
    assert j.data_ptr() == buf[1:].data_ptr() and torch.equal(j, buf[1:4])
    j = _join([buf[0:2], buf[3:6]])                      # a gap: copied
    assert j.shape == (5, 4) and j.data_ptr() != buf.data_ptr() and torch.equal(j, torch.cat([buf[0:2], buf[3:6]]))
    j = _join([torch.ones(2, 3), torch.zeros(1, 3)])
    assert j.shape == (3, 3)
    one = torch.ones(2, 3)
    assert _join([one]) is one
