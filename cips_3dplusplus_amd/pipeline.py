"""Independent views in flight together: consecutive `Generator.forward` calls alternate between a few HIP streams ("lanes").

A view at batch 1 is ~25 dependent launches; a fifth of its time goes to launches that leave most of the chip idle (the style
phase's dependent GEMVs, the modulate table, ramps and tails of the large kernels).  Views are independent of each other -- the
reference's loops render them one after the other on one stream (/root/reference/exp/tests/test_cips3dpp.py:721-738: 1000
`G_ema(...)` calls; models/render_video_web_v10.py:1806-1824: one call per frame) -- so the idle part of one view can run under the
large kernels of another.  `Generator.forward` keys its plans (workspaces, style tables) by the stream it is called on, which
makes calls on different streams independent of each other; this class supplies the streams and the ordering:

    pipe = ViewPipeline(G, lanes=2)
    outs = [pipe.submit(zs=..., cam_poses=..., ...) for ...]     # returns at once; the tensors are NOT ready on the caller's stream
    pipe.drain()                                                # the caller's stream now waits for every submitted view

`submit` orders the lane's stream behind everything the caller's stream has been given so far (the call's inputs), runs the
forward there, and marks the returned tensors as in use by the caller's stream (allocator safety).  Nothing makes the caller's
stream wait until `drain()` -- a wait per view would put every view behind the previous one again.
Measured (MI355X, FFHQ 1024^2, D = 2, N = 24, batch 1): 0.318 ms per view with two lanes against 0.363 with one (+14 % views/s);
a third lane adds nothing.  The lanes' streams are tested to run concurrently (`lane_streams`): two HIP streams may share a
hardware queue, and a pipeline on such a pair is slower than no pipeline.
"""
import os
import weakref

import torch

_HINT = os.environ.get("CIPS3D_HALF_CHIP", "1") != "0"        # A/B knob: 0 = no views-in-flight hint to the forward's launches


# Lane streams are made once per device and shared by every pipeline of the process.  HIP serves its streams from a few hardware
# queues and two streams of one queue run strictly one after the other -- measured on MI355X: of seven streams created in a row the
# pairs (0,5), (1,4), (2,3) share a queue, and a two-lane pipeline on such a pair is SLOWER than one stream (0.369 against 0.355 ms
# per view; 0.300 on any other pair).  The runtime does not say which queue a stream got, so candidates are tested: a stream joins
# the pool only if a small launch on it overtakes a long one on every stream already in the pool.
_LANE_STREAMS = {}
_REJECTED = []          # (kept alive: a released stream would be handed out again)


def _overtakes(a, b, dev):
    """True when work enqueued on stream b (after a long launch on stream a) finishes first: different hardware queues."""
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    x = torch.zeros(1 << 22, device=dev)
    y = torch.zeros(8, device=dev)
    with torch.cuda.stream(b):                        # (a stream's first launch creates its queue: not part of the race)
        y.add_(1.0)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(a):
        try:
            torch.cuda._sleep(6_000_000)              # ~3 ms of device spin
        except Exception:                             # noqa: BLE001  (no spin kernel in this build: a long chain of passes instead)
            for _ in range(400):
                x.add_(1.0)
        ea.record(a)
    with torch.cuda.stream(b):
        y.add_(1.0)
        eb.record(b)
    eb.synchronize()
    first = not ea.query()
    ea.synchronize()
    return first


def lane_streams(device, n):
    """`n` streams of `device` that run concurrently with each other (see above); fewer distinct queues than lanes: the rest are
    taken as they come."""
    dev = torch.device(device)
    pool = _LANE_STREAMS.setdefault(str(dev), [])
    tries = 0
    while len(pool) < n and tries < 24:
        tries += 1
        s = torch.cuda.Stream(device=dev)
        if all(_overtakes(p, s, dev) for p in pool):
            pool.append(s)
        else:
            _REJECTED.append(s)
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


class ViewPipeline:
    def __init__(self, G, lanes=2, device=None):
        if lanes < 1:
            raise ValueError("lanes >= 1")
        self._G = weakref.ref(G)               # (pipelines are cached per generator, weakly keyed: a strong reference here would pin it)
        dev = device if device is not None else next(G.parameters()).device
        self.device = torch.device(dev)
        self.streams = lane_streams(self.device, lanes) if self.device.type == "cuda" and lanes > 1 else []
        self._next = 0
        self._dirty = set()
        self._callers = {}                     # raw stream id -> torch.cuda.Stream of a calling stream (the object costs ~10 us to build)

    def _caller_stream(self):
        from ._lib import stream_ptr
        sid = stream_ptr()
        cur = self._callers.get(sid)
        if cur is None:
            if len(self._callers) > 16:
                self._callers.clear()
            cur = self._callers[sid] = torch.cuda.current_stream(self.device)
        return cur

    @property
    def G(self):
        return self._G()

    @property
    def lanes(self):
        return max(1, len(self.streams))

    def next_lane(self):
        return self._next % self.lanes

    def submit(self, lane=None, **kw):
        """One `G(**kw)` on the next lane (or `lane`).  Returns G's dict; see the module docstring for when it may be read."""
        return self.run(lambda: self.G(**kw), lane)

    def run(self, fn, lane=None, wait_inputs=True):
        """`fn()` -- a forward and whatever belongs to the same view (a uint8 conversion) -- on the next lane's stream (or `lane`'s).
        wait_inputs=False: the call's inputs are known to be complete (made before anything else was enqueued on the caller's
        stream and synchronised since): the lane is then not ordered behind the caller's stream -- needed where the caller's
        stream itself waits for lanes between calls (a per-view gather), which would otherwise chain the views again."""
        if not self.streams:
            return fn()
        i = self.next_lane() if lane is None else lane
        if lane is None:
            self._next += 1
        s = self.streams[i]
        cur = self._caller_stream()
        if wait_inputs:
            s.wait_stream(cur)                   # the call's inputs (and whatever else the caller enqueued before)
        torch.cuda.set_stream(s)                 # (the `with torch.cuda.stream(s)` context costs ~25 us of host time per use)
        G = self.G
        if G is not None:
            G._views_in_flight = self.lanes if _HINT else 1     # (a hint for the forward's launches: cips3d_forward_io.views_in_flight)
        try:
            out = fn()
        finally:
            torch.cuda.set_stream(cur)
            if G is not None:
                G._views_in_flight = 1
        vals = out.values() if isinstance(out, dict) else (out if isinstance(out, (tuple, list)) else (out,))
        for v in vals:                           # (allocated on the lane's stream, read on the caller's)
            if torch.is_tensor(v):
                v.record_stream(cur)
        self._dirty.add(i)
        self.last_lane = i
        return out

    def wait_lane(self, lane):
        """Order the caller's current stream behind ONE lane (what it has been given so far)."""
        if self.streams:
            self._caller_stream().wait_stream(self.streams[lane])

    def drain(self):
        """Order the caller's current stream behind every view submitted so far."""
        if not self.streams:
            return
        cur = self._caller_stream()
        for i in sorted(self._dirty):
            cur.wait_stream(self.streams[i])
        self._dirty.clear()


# one pipeline (= one set of streams, hence of lanes and forward plans) per generator and lane count, reused by every sequence
_PIPES = weakref.WeakKeyDictionary()


def pipeline_for(G, lanes=2, device=None):
    per = _PIPES.setdefault(G, {})
    dev = torch.device(device if device is not None else next(G.parameters()).device)
    key = (int(lanes), str(dev))
    if key not in per:
        per[key] = ViewPipeline(G, lanes=lanes, device=dev)
    return per[key]
