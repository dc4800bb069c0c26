"""Thin tensor-level wrappers over the C ABI (one function per entry point of include/cips3d_hip.h).

PyTorch is used for device memory and streams only: outputs are allocated with torch, pointers and
the current stream are handed to the HIP library.  Nothing here computes on the host.
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import dev_ptr, stream_ptr, check

MOD_DEMODULATE = 1
MOD_PACKED = 2
MOD_CHAINED = 4       # with MOD_PACKED: k-steps in the MFMA D-layout order (cips3d_fused_up_conv_next)
MOD_FLIP = 8          # with MOD_PACKED and ksq = 9: taps stored 180 degrees rotated (up-sampling branch of cips3d_modconv3x3)
MOD_SPLIT = 16        # with MOD_PACKED and ksq = 1: fp16 hi + lo fragments of 2^8 wm for the split-fp16 GEMM mode (GEMM_SPLIT)
MOD_SPLIT16 = 32      # with MOD_PACKED [| MOD_CHAINED]: split-fp16 fragments for the fused up-sampling stage (16-channel k-groups)
MOD_TRANSPOSE = 64    # with MOD_PACKED and ksq = 1 (fp32 fragments): the packed form of wm^T (data-gradient GEMM operand)
MOD_BF16 = 128        # with MOD_PACKED and ksq = 1: bf16 A fragments for modconv1x1_planes16 (bf16 decoder mode)

# bench.py sets this to a list to collect (start, end) event pairs around the dominant kernel's launch;
# events are recorded on the stream the kernel is launched on (torch's current stream).
KERNEL_EVENTS = {}
# An event record between two kernels drains the queue (~6 us bubble on MI355X), so a bench that times every launch slows
# the step it measures by 2-3 %: collect on every n-th call only, with event handles created ahead of the timed region.
KERNEL_EVENTS_STRIDE = 1
KERNEL_EVENTS_PHASE = 0      # the call within each stride that is timed (0 = the first; a bench picks a mid-region one)
_EVENT_POOL = []
_event_calls = {}


def prepare_event_pairs(n):
    """Create n (start, stop) event pairs with live hipEvent_t handles now (a handle only exists after a first record)."""
    for _ in range(n):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(); ev[1].record()
        _EVENT_POOL.append(ev)


def want_events(name):
    """The list to append this call's event pair to, or None (not collecting / not this call's turn)."""
    lst = KERNEL_EVENTS.get(name)
    if lst is None:
        return None
    k = _event_calls.get(name, 0)
    _event_calls[name] = k + 1
    st = max(1, KERNEL_EVENTS_STRIDE)
    return lst if k % st == KERNEL_EVENTS_PHASE % st else None


class DecoderMarks:
    """Timeline of the decoder launches of ONE cips3d_generator_forward call (cips3d_forward_io.ev_marks): hipEvent_t handles
    the call records before / after each decoder launch, and what each launch was.  Measurement only (bench.py); set
    `hip.DECODER_MARKS = DecoderMarks()` and the next Generator.forward fills it, then read `intervals()` after a
    synchronise."""
    KINDS = {0: "start", 1: "planes_gemm", 2: "gemm", 3: "lowres_gemm", 4: "fused_stage", 5: "torgb", 6: "other"}

    def __init__(self, n=40):
        self.events = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for e in self.events:
            e.record()                                # materialise the hipEvent_t handles
        self.handles = (C.c_void_p * n)(*[e.cuda_event for e in self.events])
        self.info = (C.c_int32 * (4 * n))()
        self.count = C.c_int32(0)
        self.n = n

    def io_fields(self):
        return (C.addressof(self.handles), C.addressof(self.info), C.addressof(self.count), self.n)

    def intervals(self):
        """[(kind, C_in, C_out, H_out, microseconds)] for launch k = mark k-1 -> mark k (after torch.cuda.synchronize()).
        fused_stage: C_in = the stage's width, C_out = the width of the next stage's low-resolution GEMM it also computes (0:
        none)."""
        out = []
        for k in range(1, self.count.value):
            out.append((self.KINDS.get(self.info[4 * k], "?"), self.info[4 * k + 1], self.info[4 * k + 2], self.info[4 * k + 3],
                        self.events[k - 1].elapsed_time(self.events[k]) * 1e3))
        return out


DECODER_MARKS = None      # a DecoderMarks: filled by the next one-call forward, then reset to None


def event_pair():
    if _EVENT_POOL:
        return _EVENT_POOL.pop()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record(); ev[1].record()              # materialise the hipEvent_t handles
    return ev


def _timed(name):
    lst = want_events(name)
    if lst is None:
        return None
    ev = event_pair()
    lst.append(ev)
    ev[0].record()
    return ev


def linear(x, W, bias=None, out=None, w_scale=1.0, b_scale=1.0, pixelnorm=False, lrelu=False, act_gain=1.0,
           out_scale=1.0, out_shift=0.0, trunc_mean=None, trunc_psi=1.0, out_repeat=1, out_repeat_stride=0):
    """x [B,in] -> [B,out]; see cips3d_linear."""
    lib = _lib.load()
    B, in_dim = x.shape
    out_dim = W.shape[0]
    if out is None:
        out = torch.empty(B, out_dim, device=x.device, dtype=torch.float32)
    check(lib.cips3d_linear(dev_ptr(x, "x"), x.stride(0), dev_ptr(W, "W"), dev_ptr(bias, "bias", True), dev_ptr(out, "out"),
                            out.stride(0), B, in_dim, out_dim, w_scale, b_scale, int(pixelnorm), int(lrelu), act_gain,
                            out_scale, out_shift, dev_ptr(trunc_mean, "trunc_mean", True), trunc_psi, out_repeat, out_repeat_stride,
                            stream_ptr()),
          "cips3d_linear")
    return out


# Bumped by everything that rewrites a module's style tables (FiLM table, modulation table: LinearTable.run, a full
# ForwardPlan.run).  A forward with styles_resident=True (plan.py) is only valid while nothing did since its plan's last full run.
# Bumped by everything that rewrites a module's style tables (FiLM table, decoder modulations): a forward plan's resident frame
# (plan.run, styles_resident) checks that nothing has since its last full run.  One counter per LANE: the tables of lane k > 0 are
# private to the forward plans of that lane (Generator.forward on another stream, pipeline.ViewPipeline).
import collections
STYLE_EPOCHS = collections.defaultdict(int)


class LinearTable:
    """A device-resident table of independent dense heads evaluated by one launch."""

    def __init__(self, device, lane=0):
        self.device = device
        self.lane = lane
        self._descs = []
        self._rows = 0
        self._dev = None
        self._keep = []

    def __deepcopy__(self, memo):
        # descriptors hold raw device pointers of the ORIGINAL module; the owners key their caches on parameter
        # addresses and rebuild the table for a copied module
        return LinearTable(self.device, self.lane)

    def add(self, W, bias, x, x_stride, out, out_stride, w_scale=1.0, b_scale=1.0, out_scale=1.0, out_shift=0.0,
            x_offset=0, out_offset=0):
        out_dim, in_dim = W.shape
        d = _lib.LinearDesc()
        d.W = dev_ptr(W, "W")
        d.bias = dev_ptr(bias, "bias", True)
        d.x = dev_ptr(x, "x") + 4 * x_offset
        d.out = dev_ptr(out, "out") + 4 * out_offset
        d.x_stride, d.out_stride = x_stride, out_stride
        d.in_dim, d.out_dim = in_dim, out_dim
        d.w_scale, d.b_scale, d.out_scale, d.out_shift = w_scale, b_scale, out_scale, out_shift
        d.row_begin = self._rows
        self._rows += out_dim
        self._descs.append(d)
        self._keep += [W, bias, x, out]
        self._dev = None

    def _upload(self):
        arr = (_lib.LinearDesc * len(self._descs))(*self._descs)
        raw = bytes(memoryview(arr))
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self._dev = host.to(self.device)

    def repointed(self, x_base, x_tensor, x_stride=None):
        """This table with every head reading its input from `x_tensor` (a contiguous fp32 tensor laid out like the staging
        buffer whose base address is `x_base`) instead of from that buffer -- the caller's own W+ styles, without the copy launch
        into the staging buffer.  x_stride = 0: ONE row broadcast to every sample of the batch (a [1, ...] latent for B views; the
        backward's atomics then sum the batch's gradients into that row).  The uploaded descriptors are cached per (address,
        stride): a parameter that an optimiser updates in place keeps its address from step to step, so the steady state uploads
        nothing.  Returns None the first time an address is seen (a tensor that is new on every call would cost an upload per
        call -- more than the copy it saves): the caller then stages as before."""
        key = (x_tensor.data_ptr(), x_stride)
        cache = self.__dict__.setdefault("_repointed", {})
        ent = cache.get(key)
        if ent is None:
            seen, self.__dict__["_repoint_seen"] = self.__dict__.get("_repoint_seen"), key
            if seen != key:
                return None
            if len(cache) >= 4:
                cache.clear()
            ent = LinearTable(self.device, self.lane)
            for d in self._descs:
                c = _lib.LinearDesc.from_buffer_copy(d)
                c.x = x_tensor.data_ptr() + (d.x - x_base)
                if x_stride is not None:
                    c.x_stride = x_stride
                ent._descs.append(c)
            ent._rows = self._rows
            ent._keep = list(self._keep)
            ent._upload()
            cache[key] = ent
        ent._keep_x = x_tensor
        return ent

    def run(self, B):
        if not self._descs:
            return
        if self._dev is None:
            self._upload()
        STYLE_EPOCHS[self.lane] += 1
        check(_lib.load().cips3d_linear_table(self._dev.data_ptr(), len(self._descs), self._rows, B, stream_ptr()),
              "cips3d_linear_table")

    def backward(self, B, out_base, dy_base, x_base, dx_base=None, need_dW=True, need_db=True):
        """Gradients of every head (cips3d_linear_table_bwd).  dy_base / dx_base mirror the forward's out / x buffers
        (dx_base is accumulated into: zero it first).  Returns (dW_flat, w_offsets (python list), db_flat)."""
        if self._dev is None:
            self._upload()
        in_dim = self._descs[0].in_dim
        offs, off = [], 0
        for d in self._descs:
            offs.append(off)
            off += d.out_dim * d.in_dim
        if getattr(self, "_woff_dev", None) is None or self._woff_dev.numel() != len(offs):
            self._woff_dev = torch.tensor(offs, dtype=torch.int64).to(self.device)
        dW = torch.empty(off, device=self.device) if need_dW else None
        db = torch.empty(self._rows, device=self.device) if need_db else None
        check(_lib.load().cips3d_linear_table_bwd(self._dev.data_ptr(), len(self._descs), self._rows, in_dim, B,
                                                  dev_ptr(out_base, "out"), dev_ptr(dy_base, "dy"), dev_ptr(x_base, "x"),
                                                  dev_ptr(dx_base, "dx", True), self._woff_dev.data_ptr(), dev_ptr(dW, "dW", True),
                                                  dev_ptr(db, "db", True), stream_ptr()), "cips3d_linear_table_bwd")
        return dW, offs, db


def camera_params(locations, img_size, fov_ang=6.0, dist_radius=0.12, up=None):
    """locations [B,2] -> extrinsics [B,3,4], focal/near/far [B,1,1] (reference return shapes)."""
    lib = _lib.load()
    B = locations.shape[0]
    dev = locations.device
    loc = locations.detach().float().contiguous()
    fov_t = None
    fov_s = 0.0
    if torch.is_tensor(fov_ang):
        fov_t = fov_ang.detach().float().reshape(-1).contiguous()
        if fov_t.numel() == 1 and B != 1:
            fov_t = fov_t.expand(B).contiguous()
    else:
        fov_s = float(fov_ang)
    up_t = up.detach().float().contiguous() if up is not None else None
    extr = torch.empty(B, 3, 4, device=dev)
    focal = torch.empty(B, 1, 1, device=dev)
    near = torch.empty(B, 1, 1, device=dev)
    far = torch.empty(B, 1, 1, device=dev)
    check(lib.cips3d_camera_params(dev_ptr(loc, "locations"), dev_ptr(fov_t, "fov", True), fov_s, dev_ptr(up_t, "up", True),
                                   float(dist_radius), int(img_size), B, dev_ptr(extr), dev_ptr(focal), dev_ptr(near),
                                   dev_ptr(far), stream_ptr()), "cips3d_camera_params")
    return extr, focal, near, far


def nerf_pack_weights(w_hidden, w_view, hidden, depth):
    lib = _lib.load()
    packed = torch.empty(int(lib.cips3d_nerf_packed_floats(hidden, depth)), device=w_view.device, dtype=torch.float32)
    check(lib.cips3d_nerf_pack_weights(dev_ptr(w_hidden, "w_hidden", True), dev_ptr(w_view, "w_view"), dev_ptr(packed),
                                       hidden, depth, stream_ptr()), "cips3d_nerf_pack_weights")
    return packed


def nerf_pack_weights32(w_hidden, w_view, hidden, depth):
    """The exact-fp32 weight stream (cips3d_nerf_params.packed32): fp32 A fragments of v_mfma_f32_16x16x4_f32."""
    lib = _lib.load()
    packed = torch.empty(int(lib.cips3d_nerf_packed_floats(hidden, depth)), device=w_view.device, dtype=torch.float32)
    check(lib.cips3d_nerf_pack_weights32(dev_ptr(w_hidden, "w_hidden", True), dev_ptr(w_view, "w_view"), dev_ptr(packed),
                                         hidden, depth, stream_ptr()), "cips3d_nerf_pack_weights32")
    return packed


def nerf_suggest_chunks(B, img_size, n_samples):
    return int(_lib.load().cips3d_nerf_suggest_chunks(B, img_size, n_samples))


def _nerf_params(kw):
    p = _lib.NerfParams()
    ptr_fields = ("near_", "far_", "w_first", "packed", "w_view", "film", "layer_bias",
                  "w_sigma", "w_rgb", "b_sigma", "b_rgb", "sigmoid_beta")
    for f in ptr_fields:
        setattr(p, f, dev_ptr(kw[f], f))
    p.packed32 = dev_ptr(kw.get("packed32"), "packed32", True)          # exact-fp32 arithmetic (csrc/nerf.hip, F32 instantiation)
    p.part = dev_ptr(kw.get("part"), "part", True)
    for f in ("cam_poses", "focals"):
        setattr(p, f, dev_ptr(kw.get(f), f, kw.get("x_pts") is not None))
    p.perturb_u = dev_ptr(kw.get("perturb_u"), "perturb_u", True)
    p.sdf = dev_ptr(kw.get("sdf"), "sdf", True)
    for f in ("stash", "bwd_sdf", "bwd_crgb"):                        # differentiable forward (cips3d_nerf_bwd_fused's inputs)
        setattr(p, f, dev_ptr(kw.get(f), f, True))
    for f in ("x_pts", "x_rays_d", "x_viewdirs", "x_z_vals"):          # explicit-geometry mode
        setattr(p, f, dev_ptr(kw.get(f), f, True))
    p.n_rays = int(kw.get("n_rays", 0))
    zw = kw.get("zero_words")                                          # float scratch the launch leaves zeroed (cips3d_nerf_params.zero_words)
    if zw is not None:
        p.zero_words, p.n_zero_words = dev_ptr(zw, "zero_words"), zw.numel()
    p.raw_density = int(bool(kw.get("raw_density", False)))            # with_sdf = False (backward: sigmoid_beta = None)
    for f in ("B", "img_size", "n_samples", "hidden", "depth", "static_viewdirs", "n_chunks"):
        setattr(p, f, int(kw[f]))
    return p


def nerf_render(**kw):
    """Fill cips3d_nerf_params from keyword tensors / ints and launch the fused renderer (chunk partials -> `part`)."""
    lib = _lib.load()
    if kw.get("part") is None:
        raise RuntimeError("part is required")
    p = _nerf_params(kw)
    ev = _timed("nerf_render")
    check(lib.cips3d_nerf_render(C.byref(p), stream_ptr()), "cips3d_nerf_render")
    if ev:
        ev[1].record()


def nerf_render_maps(**kw):
    """Renderer + ordered combination of the chunk partials -> (features, thumb, xyz, mask), [B,C,S,S] (or [B,C,n_rays,1] in
    explicit-geometry mode).  The render kernel combines the partials itself when cips3d_nerf_fuses_finish says so; else
    they go through `part` and cips3d_nerf_finish."""
    lib = _lib.load()
    p = _nerf_params(kw)
    B, H, S, n_rays = p.B, p.hidden, p.img_size, p.n_rays
    dev = kw["near_"].device
    shp = (n_rays, 1) if n_rays > 0 else (S, S)
    features = torch.empty(B, H, *shp, device=dev)
    thumb, xyz = torch.empty(B, 3, *shp, device=dev), torch.empty(B, 3, *shp, device=dev)
    mask = torch.empty(B, 2, *shp, device=dev)
    p.o_features, p.o_thumb, p.o_xyz, p.o_mask = dev_ptr(features), dev_ptr(thumb), dev_ptr(xyz), dev_ptr(mask)
    fused = bool(lib.cips3d_nerf_fuses_finish(C.byref(p)))
    planar = bool(kw.get("planar_mask"))          # mask as [2,B,...]: written that way by the fused finish, one transposing copy otherwise
    if planar and fused:
        mask = torch.empty(2, B, *shp, device=dev)
        p.o_mask, p.mask_planar = dev_ptr(mask), 1
    part = None
    if not fused:
        R = n_rays if n_rays > 0 else S * S
        part = torch.empty(p.n_chunks, B, H + 8, R, device=dev)
        p.part = dev_ptr(part)
    ev = _timed("nerf_render")
    check(lib.cips3d_nerf_render(C.byref(p), stream_ptr()), "cips3d_nerf_render")
    if ev:
        ev[1].record()
    if not fused:
        check(lib.cips3d_nerf_finish_rays(p.part, p.n_chunks, B, n_rays if n_rays > 0 else S * S, H, dev_ptr(features),
                                          dev_ptr(thumb), dev_ptr(xyz), dev_ptr(mask), stream_ptr()), "cips3d_nerf_finish_rays")
        if planar:
            mask = mask.transpose(0, 1).contiguous()
    return features, thumb, xyz, mask


def rays_in_world(cam_poses, focals, img_size, static_viewdirs=False):
    """-> rays_o, rays_d, viewdirs, each [B,S,S,3] (Render.get_rays_in_world)."""
    lib = _lib.load()
    B = cam_poses.shape[0]
    o, d, v = (torch.empty(B, img_size, img_size, 3, device=cam_poses.device) for _ in range(3))
    c2w, foc = cam_poses.float().contiguous(), focals.float().reshape(B).contiguous()     # converted copies live until the launch
    check(lib.cips3d_rays_in_world(dev_ptr(c2w, "c2w"), dev_ptr(foc, "focal"),
                                   img_size, int(bool(static_viewdirs)), B, dev_ptr(o), dev_ptr(d), dev_ptr(v), stream_ptr()),
          "cips3d_rays_in_world")
    return o, d, v


def z_vals(near, far, B, R, N, perturb_u=None, stratified=False):
    """offset sampling (perturb_u: one uniform per ray, [B,R]) or, stratified=True, the classic stratified branch (perturb_u:
    one uniform per sample, [B,R,N])."""
    lib = _lib.load()
    z = torch.empty(B, R, N, device=near.device)
    nr, fr = near.float().reshape(B).contiguous(), far.float().reshape(B).contiguous()   # converted copies live until the launch
    if stratified:
        u = None if perturb_u is None else perturb_u.float().reshape(B, R, N).contiguous()
        check(lib.cips3d_z_vals_stratified(dev_ptr(nr, "near"), dev_ptr(fr, "far"), dev_ptr(u, "u", True), B, R, N, dev_ptr(z),
                                           stream_ptr()), "cips3d_z_vals_stratified")
        return z
    u = None if perturb_u is None else perturb_u.float().reshape(B, R).contiguous()
    check(lib.cips3d_z_vals(dev_ptr(nr, "near"), dev_ptr(fr, "far"),
                            dev_ptr(u, "u", True), B, R, N, dev_ptr(z), stream_ptr()), "cips3d_z_vals")
    return z


def ray_points(rays_o, rays_d, z, near=None, far=None, want_pts=True, want_normalized=False):
    """rays [B,R,3], z [B,R,N] -> pts and/or normalised pts [B,R,N,3]."""
    lib = _lib.load()
    B, R, N = z.shape
    dev = z.device
    pts = torch.empty(B, R, N, 3, device=dev) if want_pts else None
    ptsn = torch.empty(B, R, N, 3, device=dev) if want_normalized else None
    nr = None if near is None else near.float().reshape(B).contiguous()
    fr = None if far is None else far.float().reshape(B).contiguous()
    check(lib.cips3d_ray_points(dev_ptr(rays_o, "rays_o"), dev_ptr(rays_d, "rays_d"), dev_ptr(z, "z_vals"), dev_ptr(nr, "near", True),
                                dev_ptr(fr, "far", True), B, R, N, dev_ptr(pts, "pts", True), dev_ptr(ptsn, "ptsn", True),
                                stream_ptr()), "cips3d_ray_points")
    return pts, ptsn


VI_RAW_DENSITY, VI_FORCE_BACKGROUND = 1, 2


def volume_integration(rgb, sdf, features, z, rays_d, pts, sigmoid_beta, raw_density=False, force_background=False):
    """[n,N,3], [n,N], [n,N,C]|None, [n,N], [n,3], [n,N,3] -> rgb_map [n,3], feature_map [n,C]|None, xyz [n,3], mask [n,2].
    raw_density: `sdf` is the raw density output (softplus branch, with_sdf=False); force_background: nerf_utils.py:309-310."""
    lib = _lib.load()
    n, N = z.shape
    dev = z.device
    Cc = features.shape[-1] if features is not None else 0
    rgb_map, xyz, mask = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 2, device=dev)
    fmap = torch.empty(n, Cc, device=dev) if features is not None else None
    check(lib.cips3d_volume_integration(dev_ptr(rgb, "rgb"), dev_ptr(sdf, "sdf"), dev_ptr(features, "features", True),
                                        dev_ptr(z, "z_vals"), dev_ptr(rays_d, "rays_d"), dev_ptr(pts, "pts"),
                                        dev_ptr(sigmoid_beta, "sigmoid_beta", raw_density), n, N, Cc,
                                        (VI_RAW_DENSITY if raw_density else 0) | (VI_FORCE_BACKGROUND if force_background else 0),
                                        dev_ptr(rgb_map),
                                        dev_ptr(fmap, "feature_map", True), dev_ptr(xyz), dev_ptr(mask), stream_ptr()),
          "cips3d_volume_integration")
    return rgb_map, fmap, xyz, mask


def points_linear(x, W, bias=None, film=None, out_scale=1.0, out_shift=0.0):
    """x (b, ..., in) point-major -> (b, ..., out): LinearLayer (film None) or FiLMSiren (film [B,2,out] = gamma, beta)."""
    lib = _lib.load()
    xin = x.float().contiguous()
    B, in_dim = xin.shape[0], xin.shape[-1]
    n = xin.numel() // in_dim
    out_dim = W.shape[0]
    y = torch.empty(*xin.shape[:-1], out_dim, device=x.device)
    check(lib.cips3d_points_linear(dev_ptr(xin, "x"), dev_ptr(W, "W"), dev_ptr(bias, "bias", True), dev_ptr(film, "film", True), n,
                                   max(1, n // B), in_dim, out_dim, 0 if film is None else 1, float(out_scale), float(out_shift),
                                   dev_ptr(y), stream_ptr()), "cips3d_points_linear")
    return y


def nerf_finish(part, n_chunks, B, img_size, hidden, n_rays=None):
    lib = _lib.load()
    dev = part.device
    if n_rays is not None:           # explicit-geometry mode: a [B, C, n_rays, 1] "image"
        features = torch.empty(B, hidden, n_rays, 1, device=dev)
        thumb, xyz = torch.empty(B, 3, n_rays, 1, device=dev), torch.empty(B, 3, n_rays, 1, device=dev)
        mask = torch.empty(B, 2, n_rays, 1, device=dev)
        check(lib.cips3d_nerf_finish_rays(dev_ptr(part), n_chunks, B, n_rays, hidden, dev_ptr(features), dev_ptr(thumb),
                                          dev_ptr(xyz), dev_ptr(mask), stream_ptr()), "cips3d_nerf_finish_rays")
        return features, thumb, xyz, mask
    R = img_size * img_size
    features = torch.empty(B, hidden, img_size, img_size, device=dev)
    thumb = torch.empty(B, 3, img_size, img_size, device=dev)
    xyz = torch.empty(B, 3, img_size, img_size, device=dev)
    mask = torch.empty(B, 2, img_size, img_size, device=dev)
    check(lib.cips3d_nerf_finish(dev_ptr(part), n_chunks, B, img_size, hidden, dev_ptr(features), dev_ptr(thumb),
                                 dev_ptr(xyz), dev_ptr(mask), stream_ptr()), "cips3d_nerf_finish")
    return features, thumb, xyz, mask


def modulate_weights(W, s, s_stride, B, Cout, Cin, ksq, scale, demodulate, packed, s_offset=0, out=None, flip=False,
                     split=False, bf16=False):
    lib = _lib.load()
    if out is None:
        # (the 3x3 kernel's split-fp16 tap-pair fragments hold ten taps per weight row: five pairs, the tenth tap zero)
        out = torch.empty(B * Cout * Cin * (10 if (split and packed and ksq == 9) else ksq), device=W.device, dtype=torch.float32)
    flags = ((MOD_DEMODULATE if demodulate else 0) | (MOD_PACKED if packed else 0) | (MOD_FLIP if flip else 0) |
             (MOD_SPLIT if split else 0) | (MOD_BF16 if bf16 else 0))
    check(lib.cips3d_modulate_weights(dev_ptr(W, "W"), dev_ptr(s, "s") + 4 * s_offset, s_stride, dev_ptr(out), B, Cout, Cin,
                                      ksq, float(scale), flags, stream_ptr()), "cips3d_modulate_weights")
    return out


def modconv1x1_supported(Cin, Cout, HW):
    return bool(_lib.load().cips3d_modconv1x1_supported(Cin, Cout, HW))


# ---- range tracking of the split-fp16 modes (cips3d_range, include/cips3d_hip.h): every split happens on x * 2^-e with a power
# of two per (tensor, sample) taken from a bound of max|x|, so that the decoder's default arithmetic keeps fp32's exponent
# range.  amax arrays are [B, AMAX_FLOATS] fp32 tensors; a kernel that produced a tensor leaves its amax attached to it
# (`tag_amax`), a split GEMM that reads an untagged tensor measures it first (one HBM pass).
def amax_floats():
    """floats per sample of an amax array (CIPS3D_AMAX_SLOTS * CIPS3D_AMAX_STRIDE of the loaded library)"""
    _lib.load()
    return _lib.AMAX_FLOATS


NOISE_BOUND_RNG = 6.0        # cips3d_rng_fill: |n| <= sqrt(50 ln 2) = 5.89
NOISE_BOUND_TORCH = 7.0      # torch.randn on the device: Box-Muller on 32-bit uniforms, |n| <= 6.8


def new_amax(B, device):
    return torch.zeros(B, amax_floats(), device=device, dtype=torch.float32)


def absmax(x, B=None):
    """amax array of x viewed as [B, -1] (B defaults to x.shape[0])."""
    B = x.shape[0] if B is None else B
    x = x.contiguous()
    amax = torch.empty(B, amax_floats(), device=x.device, dtype=torch.float32)
    check(_lib.load().cips3d_absmax(dev_ptr(x, "x"), B, x.numel() // B, dev_ptr(amax), stream_ptr()), "cips3d_absmax")
    return amax


def amax_value(amax):
    """[B] tensor of the maxima an amax array holds (tests, diagnostics)."""
    return amax.view(amax.shape[0], _lib.AMAX_SLOTS, _lib.AMAX_STRIDE)[:, :, 0].max(dim=1).values


def tag_amax(t, amax):
    """Remember the amax array of tensor t (valid while t is not modified in place)."""
    t._cips3d_amax = (amax, t._version)
    return t


def amax_of(t, measure=True):
    tag = getattr(t, "_cips3d_amax", None)
    if tag is not None and tag[1] == t._version and tag[0].shape[0] == t.shape[0]:
        return tag[0]
    return absmax(t) if measure else None


_const_amax = {}


def const_amax(B, bound, device):
    """amax array of a tensor whose bound is known a priori (sines: 1)."""
    key = (B, float(bound), str(device))
    t = _const_amax.get(key)
    if t is None:
        t = _const_amax[key] = torch.full((B, amax_floats()), float(bound), device=device, dtype=torch.float32)
    return t


def range_consts(B, bias, noise_w, w_gain, fir=None, noise=None, noise_bound=0.0):
    """lconst [B, 4] of one StyledConv (cips3d_range_consts).  The bound of |noise| is measured from `noise` (a device tensor)
    when given -- no host round trip -- and / or taken from `noise_bound`."""
    lib = _lib.load()
    dev = bias.device
    lconst = torch.empty(B, 4, device=dev, dtype=torch.float32)
    na = absmax(noise, B=1) if noise is not None else None
    bflat = bias.contiguous().view(-1)
    check(lib.cips3d_range_consts(dev_ptr(bflat, "bias"), bflat.numel(), dev_ptr(noise_w, "noise_w", True), float(noise_bound),
                                  dev_ptr(na, "noise_amax", True), float(w_gain), dev_ptr(fir, "fir", True), dev_ptr(lconst), B,
                                  stream_ptr()), "cips3d_range_consts")
    return lconst


def _range(**kw):
    rg = _lib.Range()
    keep = []
    for k, v in kw.items():
        if v is not None:
            keep.append(v)
            setattr(rg, k, v.data_ptr())
    return rg, keep


GEMM_BF16 = 0x100      # CIPS3D_GEMM_BF16: bf16 compute mode of the decoder GEMMs (BASELINE config 3)
Y_BF16 = 0x200         # CIPS3D_Y_BF16: the pre-FIR low-resolution GEMM result of an up-sampling stage is stored as bf16
GEMM_SPLIT = 0x400     # CIPS3D_GEMM_SPLIT: fp32-equivalent split-fp16 products (weights packed with MOD_SPLIT)
STAGE_FLAT = 0x1000    # CIPS3D_STAGE_FLAT: the fused stage kernel on a block that does not up-sample


def modconv1x1(x, wm_packed, Cout, epilogue=0, noise=None, noise_w=None, bias=None, out=None, bf16=False, out_bf16=False,
               split=False, x_amax=None, track=None):
    """out_bf16 (epilogue 0 only): the result is stored as a torch.bfloat16 tensor (CIPS3D_Y_BF16).
    split: fp32-equivalent split-fp16 products; wm_packed must come from modulate_weights(..., packed=True, split=True).
    The activations are split as x * 2^-e (cips3d_range): x_amax = the amax array of x, else the one attached to x by the kernel
    that made it, else measured here.  track (default: split): leave the output's amax attached to the result."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    if split and x_amax is None:
        x_amax = amax_of(x)
    if track is None:
        track = split
    out_amax = new_amax(B, x.device) if (track and not out_bf16) else None
    rg, _keep = _range(x_amax=x_amax if split else None, out_amax=out_amax)
    odt = torch.bfloat16 if out_bf16 else torch.float32
    if out is None:
        out = torch.empty(B, Cout, H, W, device=x.device, dtype=odt)
    nb = 0
    if noise is not None and noise.shape[0] == B and B > 1:
        nb = H * W
    if noise is not None and noise.shape[0] not in (1, B):
        raise RuntimeError("noise batch must be 1 or B")
    check(lib.cips3d_modconv1x1(dev_ptr(x, "x"), dev_ptr(wm_packed, "wm"), dev_ptr(out, "out", dtype=odt), B, Cin, Cout, H * W,
                                epilogue | (GEMM_BF16 if bf16 else 0) | (Y_BF16 if out_bf16 else 0) | (GEMM_SPLIT if split else 0),
                                dev_ptr(noise, "noise", True), nb, dev_ptr(noise_w, "noise_w", True), dev_ptr(bias, "bias", True),
                                C.byref(rg), stream_ptr()), "cips3d_modconv1x1")
    if out_amax is not None:
        tag_amax(out, out_amax)
    return out


def up2_fir_act(y_lo, fir, noise, noise_w, bias, out=None, track=False):
    """track: the kernel records the output's amax array (cips3d_range) and leaves it attached to the result."""
    lib = _lib.load()
    B, Cc, H, W = y_lo.shape
    if out is None:
        out = torch.empty(B, Cc, 2 * H, 2 * W, device=y_lo.device, dtype=torch.float32)
    nb = 4 * H * W if (noise is not None and noise.shape[0] == B and B > 1) else 0
    out_amax = new_amax(B, y_lo.device) if track else None
    check(lib.cips3d_up2_fir_act(dev_ptr(y_lo, "y_lo"), dev_ptr(fir, "fir"), dev_ptr(out), B, Cc, H, W,
                                 dev_ptr(noise, "noise", True), nb, dev_ptr(noise_w, "noise_w", True), dev_ptr(bias, "bias"),
                                 dev_ptr(out_amax, "out_amax", True), stream_ptr()), "cips3d_up2_fir_act")
    if out_amax is not None:
        tag_amax(out, out_amax)
    return out


def noise_bias_act(x, noise, noise_w, bias):
    lib = _lib.load()
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    nb = H * W if (noise is not None and noise.shape[0] == B and B > 1) else 0
    check(lib.cips3d_noise_bias_act(dev_ptr(x, "x"), dev_ptr(noise, "noise", True), nb, dev_ptr(noise_w, "noise_w", True), dev_ptr(bias, "bias"),
                                    dev_ptr(out), B, Cc, H * W, stream_ptr()), "cips3d_noise_bias_act")
    return out


def torgb(x, wm, bias, skip=None, skip_up=False, fir=None, out=None):
    lib = _lib.load()
    B, Cin, H, W = x.shape
    if out is None:
        out = torch.empty(B, 3, H, W, device=x.device, dtype=torch.float32)
    check(lib.cips3d_torgb(dev_ptr(x, "x"), dev_ptr(wm, "wm"), dev_ptr(bias, "bias"), dev_ptr(skip, "skip", True),
                           int(bool(skip_up)), dev_ptr(fir, "fir", True), dev_ptr(out), B, Cin, H, W, stream_ptr()),
          "cips3d_torgb")
    return out


def modconv_kxk(x, wm, Cout, k, transpose2=False):
    lib = _lib.load()
    B, Cin, H, W = x.shape
    OH = 2 * H - 1 + k - 1 if transpose2 else H
    OW = 2 * W - 1 + k - 1 if transpose2 else W
    out = torch.empty(B, Cout, OH, OW, device=x.device, dtype=torch.float32)
    check(lib.cips3d_modconv_kxk(dev_ptr(x, "x"), dev_ptr(wm, "wm"), dev_ptr(out), B, Cin, Cout, H, W, k,
                                 int(bool(transpose2)), stream_ptr()), "cips3d_modconv_kxk")
    return out


def planes_supported(Cin, Cout, HW):
    return bool(_lib.load().cips3d_planes_supported(Cin, Cout, HW))


def to_planes(x, ranged=True):
    """fp32 [B,C,H,W] -> split-fp16 planes (torch.float16 tensor [B, C/8, 2, H*W, 8]: hi plane, lo plane) of x * 2^-e, e one
    power of two per sample that puts max|x| into [2^14, 2^15) (cips3d_range).  The exponents ([B, blocks of 128 pixels] int32:
    the format allows one per pixel block, this conversion writes the sample's everywhere) travel as an attribute of the
    result (`.cips3d_exp`); ranged=False stores x itself (e = 0)."""
    lib = _lib.load()
    B, Cc, H, W = x.shape
    p = torch.empty(B, Cc // 8, 2, H * W, 8, device=x.device, dtype=torch.float16)
    amax = amax_of(x) if ranged else None
    nblk = (H * W + _lib.PLANES_EXP_BLOCK - 1) // _lib.PLANES_EXP_BLOCK
    exps = torch.zeros(B, nblk, device=x.device, dtype=torch.int32) if ranged else None
    pmax = torch.zeros(B, (H * W + 63) // 64, Cc // 16, device=x.device) if (ranged and Cc % 16 == 0) else None
    check(lib.cips3d_to_planes(dev_ptr(x, "x"), p.data_ptr(), B, Cc, H * W, dev_ptr(amax, "amax", True),
                               exps.data_ptr() if ranged else None, dev_ptr(pmax, "pmax", True), stream_ptr()), "cips3d_to_planes")
    p.cips3d_exp, p.cips3d_pmax = exps, pmax
    return p


def from_planes(p, H, W):
    lib = _lib.load()
    B, C8 = p.shape[0], p.shape[1]
    x = torch.empty(B, C8 * 8, H, W, device=p.device, dtype=torch.float32)
    exps = getattr(p, "cips3d_exp", None)
    check(lib.cips3d_from_planes(dev_ptr(p, "planes", dtype=torch.float16), dev_ptr(x), B, C8 * 8, H * W,
                                 exps.data_ptr() if exps is not None else None, stream_ptr()), "cips3d_from_planes")
    return x


def modconv1x1_planes(xp, wm_split, Cout, HW, out_format="planes", epilogue=0, noise=None, noise_w=None, bias=None,
                      rgb_w=None, rgb_part=None, demodulated=True, lconst=None, ride=None, half_chip=False):
    """1x1 modulated conv on split-fp16 planes (csrc/chain.hip).  xp from to_planes / a previous call (its exponents are read
    from its attribute; a planes output carries its own); wm_split from
    modulate_weights(..., packed=True, split=True).  out_format: "planes" | "fp32" | "bf16" ([B,Cout,HW]).
    demodulated: the weights were demodulated (unit row norm -> the sqrt(Cin) gain of the output bound); for others pass lconst
    from a modulate table that measured the row L1 norms.
    ride: a ToRGB fold carried by this launch (cips3d_reduce_job; range-tracked inputs only) -- dict(part [n_slots, Br, 3, HWr],
    biases [list of [3]], skip [Br, 3, HWr] or None, out [Br, 3, HWr]): out = skip + sum of the slots + sum of the biases, the
    arithmetic and order of torgb_reduce.
    half_chip: cips3d_range.half_chip (another view's launches are in flight: 128 x 128 tiles on half the CUs; same results)."""
    lib = _lib.load()
    B, Cin = xp.shape[0], xp.shape[1] * 8
    dev = xp.device
    fmt = {"fp32": 0, "planes": 1, "bf16": 2}[out_format]
    if fmt == 1:
        out = torch.empty(B, Cout // 8, 2, HW, 8, device=dev, dtype=torch.float16)
    else:
        out = torch.empty(B, Cout, HW, device=dev, dtype=torch.bfloat16 if fmt == 2 else torch.float32)
    nb = HW if (noise is not None and noise.shape[0] == B and B > 1) else 0
    x_exp, x_pmax = getattr(xp, "cips3d_exp", None), getattr(xp, "cips3d_pmax", None)
    ranged = x_exp is not None
    rg, _keep = None, None
    if ranged:
        if lconst is None and fmt == 1:
            if not demodulated:
                raise RuntimeError("a planes output of a non-demodulated conv needs the layer's lconst (row L1 norms)")
            if epilogue == 1:
                lconst = range_consts(B, bias, noise_w, Cin ** 0.5, noise=noise)
            else:
                lconst = range_consts(B, torch.zeros(1, device=dev), None, Cin ** 0.5)
                lconst[:, 1] /= 2.0 ** 0.5      # no activation: |out| <= sqrt(Cin) max|x|
        out_amax = new_amax(B, dev) if fmt != 1 else None
        nblk = (HW + _lib.PLANES_EXP_BLOCK - 1) // _lib.PLANES_EXP_BLOCK
        out_exp = torch.zeros(B, nblk, device=dev, dtype=torch.int32) if fmt == 1 else None
        out_pmax = torch.zeros(B, (HW + 63) // 64, Cout // 16, device=dev) if fmt == 1 else None
        # (x_pmax None: the kernel bounds max|in| by what the input's exponent encodes -- looser, still rigorous)
        rg, _keep = _range(x_exp=x_exp, x_pmax=x_pmax if fmt == 1 else None, lconst=lconst if fmt == 1 else None,
                           out_amax=out_amax, out_exp=out_exp, out_pmax=out_pmax)
    job = None
    if ride is not None:
        if rg is None:
            raise RuntimeError("a riding ToRGB fold needs range-tracked input planes (to_planes with x_amax)")
        part, r_out = ride["part"], ride["out"]
        n_slots, Br, _, HWr = part.shape
        job = _lib.ReduceJob()
        job.part, job.out = dev_ptr(part, "ride.part"), dev_ptr(r_out, "ride.out")
        job.skip = dev_ptr(ride.get("skip"), "ride.skip", True)
        for k, bk in enumerate(ride.get("biases", [])):
            job.bias[k] = dev_ptr(bk, "ride.bias")
        job.n4, job.HW4, job.slot_stride = Br * 3 * HWr // 4, HWr // 4, Br * 3 * HWr
        job.n_slots, job.n_bias = n_slots, len(ride.get("biases", []))
        rg.ride = C.addressof(job)
    if rg is not None:
        rg.half_chip = int(bool(half_chip))
    check(lib.cips3d_modconv1x1_planes(dev_ptr(xp, "x_planes", dtype=torch.float16), dev_ptr(wm_split, "wm"), out.data_ptr(), fmt,
                                       B, Cin, Cout, HW, epilogue, dev_ptr(noise, "noise", True), nb,
                                       dev_ptr(noise_w, "noise_w", True), dev_ptr(bias, "bias", True),
                                       dev_ptr(rgb_w, "rgb_w", True), dev_ptr(rgb_part, "rgb_part", True), None,
                                       C.byref(rg) if rg is not None else None, stream_ptr()),
          "cips3d_modconv1x1_planes")
    if ranged:
        if fmt == 1:
            out.cips3d_exp, out.cips3d_pmax = out_exp, out_pmax
        elif fmt == 0:
            tag_amax(out, out_amax)
    return out


def to_planes16(x):
    """fp32 [B,C,H,W] -> planes16 (torch.bfloat16 tensor [B, C/8, H*W, 8]; round to nearest even)."""
    lib = _lib.load()
    B, Cc, H, W = x.shape
    p = torch.empty(B, Cc // 8, H * W, 8, device=x.device, dtype=torch.bfloat16)
    check(lib.cips3d_to_planes16(dev_ptr(x, "x"), p.data_ptr(), B, Cc, H * W, stream_ptr()), "cips3d_to_planes16")
    return p


def from_planes16(p, H, W):
    lib = _lib.load()
    B, C8 = p.shape[0], p.shape[1]
    x = torch.empty(B, C8 * 8, H, W, device=p.device, dtype=torch.float32)
    check(lib.cips3d_from_planes16(dev_ptr(p, "planes16", dtype=torch.bfloat16), dev_ptr(x), B, C8 * 8, H * W, stream_ptr()),
          "cips3d_from_planes16")
    return x


def modconv1x1_planes16(xp, wm_bf16, Cout, HW, out_format="planes16", epilogue=0, noise=None, noise_w=None, bias=None,
                        rgb_w=None, rgb_part=None):
    """1x1 modulated conv of the bf16 decoder mode on planes16 (csrc/chain.hip, NP = 1).  xp from to_planes16 / a previous
    call; wm_bf16 from modulate_weights(..., packed=True, bf16=True).  out_format: "planes16" | "fp32" | "bf16" ([B,Cout,HW]).
    Returns (out, number of rgb_part row-block slots written)."""
    lib = _lib.load()
    B, Cin = xp.shape[0], xp.shape[1] * 8
    fmt = {"fp32": 0, "bf16": 2, "planes16": 3}[out_format]
    if fmt == 3:
        out = torch.empty(B, Cout // 8, HW, 8, device=xp.device, dtype=torch.bfloat16)
    else:
        out = torch.empty(B, Cout, HW, device=xp.device, dtype=torch.bfloat16 if fmt == 2 else torch.float32)
    nb = HW if (noise is not None and noise.shape[0] == B and B > 1) else 0
    nblk = C.c_int(0)
    check(lib.cips3d_modconv1x1_planes16(dev_ptr(xp, "x_planes16", dtype=torch.bfloat16), dev_ptr(wm_bf16, "wm"), out.data_ptr(),
                                         fmt, B, Cin, Cout, HW, epilogue, dev_ptr(noise, "noise", True), nb,
                                         dev_ptr(noise_w, "noise_w", True), dev_ptr(bias, "bias", True),
                                         dev_ptr(rgb_w, "rgb_w", True), dev_ptr(rgb_part, "rgb_part", True), C.byref(nblk),
                                         None, stream_ptr()), "cips3d_modconv1x1_planes16")
    return out, nblk.value


# Fresh noise (NoiseInjection maps, per-ray jitter) from the library's own generator (csrc/rng.hip) in one launch; 0 = torch's
# randn / rand, two launches (A/B knob).  Either way the stream is a function of torch's seed and generator state.
FAST_RNG = os.environ.get("CIPS3D_FAST_RNG", "1") != "0"


def rng_reserve(n_normal, n_uniform, device):
    """(seed, base) of a cips3d_rng_fill draw of these counts taken from torch's generator for `device`, whose Philox offset is
    advanced by what the draw consumes (torch keeps the offset a multiple of 4)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    seed, base = gen.initial_seed(), gen.get_offset()
    threads = int(_lib.load().cips3d_rng_fill_threads(n_normal, n_uniform))
    gen.set_offset(base + 4 * ((threads + 3) // 4))
    return int(seed) & 0xFFFFFFFFFFFFFFFF, int(base)


def rng_fill(n_normal, n_uniform, device, seed=None, base=None):
    """(normal [n_normal] ~ N(0,1), uniform [n_uniform] ~ U[0,1)) fp32 tensors from cips3d_rng_fill (None for a zero count).
    With seed / base omitted the state is torch's: key = the device generator's initial_seed(), counter base = its Philox
    offset, which is advanced by what the call consumes -- torch.manual_seed / get_rng_state / set_rng_state govern this
    stream like torch's own draws."""
    lib = _lib.load()
    device = torch.device(device)
    normal = torch.empty(n_normal, device=device, dtype=torch.float32) if n_normal else None
    uniform = torch.empty(n_uniform, device=device, dtype=torch.float32) if n_uniform else None
    if n_normal == 0 and n_uniform == 0:
        return normal, uniform
    if seed is None:
        seed, base = rng_reserve(n_normal, n_uniform, device)
    check(lib.cips3d_rng_fill(int(seed) & 0xFFFFFFFFFFFFFFFF, int(base), dev_ptr(normal, "normal", True), n_normal,
                              dev_ptr(uniform, "uniform", True), n_uniform, stream_ptr()), "cips3d_rng_fill")
    return normal, uniform


def rng_words(seed, base, n_threads, device):
    """The raw Philox words of threads 0 .. n_threads-1 as an int64 tensor [n_threads, 4] (tests)."""
    out = torch.empty(n_threads, 4, device=device, dtype=torch.int32)
    check(_lib.load().cips3d_rng_words(int(seed) & 0xFFFFFFFFFFFFFFFF, int(base), out.data_ptr(), n_threads, stream_ptr()),
          "cips3d_rng_words")
    return out.to(torch.int64) & 0xFFFFFFFF


def modconv3x3_supported(Cin, Cout, H, W, up):
    return bool(_lib.load().cips3d_modconv3x3_supported(Cin, Cout, H, W, int(bool(up))))


def modconv3x3(x, wm_packed, Cout, up=False, fir=None, epilogue=0, noise=None, noise_w=None, bias=None, split=False, x_amax=None):
    """3x3 modulated conv on the LDS-tiled MFMA kernel (csrc/conv3x3.hip).  x [B,Cin,H,W]; wm_packed from
    modulate_weights(..., ksq=9, packed=True, flip=up); up: conv_transpose2d(stride 2) + Blur(fir) fused -> [B,Cout,2H,2W].
    split: fp32-equivalent split-fp16 products (wm_packed from modulate_weights(..., split=True) as well); x is split as x * 2^-e
    from its measured maximum (x_amax: its amax array, else the one attached to x, else measured here)."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    OH, OW = (2 * H, 2 * W) if up else (H, W)
    out = torch.empty(B, Cout, OH, OW, device=x.device, dtype=torch.float32)
    nb = 0
    if noise is not None:
        if noise.shape[0] not in (1, B) or tuple(noise.shape[-2:]) != (OH, OW):
            raise RuntimeError(f"noise must be (1|{B},1,{OH},{OW}), got {tuple(noise.shape)}")
        nb = OH * OW if (noise.shape[0] == B and B > 1) else 0
    rg, _keep = None, None
    if split:
        rg, _keep = _range(x_amax=x_amax if x_amax is not None else amax_of(x))
    check(lib.cips3d_modconv3x3(dev_ptr(x, "x"), dev_ptr(wm_packed, "wm"), dev_ptr(out), B, Cin, Cout, H, W, int(bool(up)),
                                dev_ptr(fir, "fir", True), int(epilogue) | (GEMM_SPLIT if split else 0), dev_ptr(noise, "noise", True), nb,
                                dev_ptr(noise_w, "noise_w", True), dev_ptr(bias, "bias", True),
                                C.byref(rg) if rg is not None else None, stream_ptr()),
          "cips3d_modconv3x3")
    return out


def fused_up_conv_chains(C_):
    return bool(_lib.load().cips3d_fused_up_conv_chains(C_))


def fused_up_conv(y_lo, fir, noise1, noise_w1, bias1, wm2_packed, noise2, noise_w2, bias2, wm_rgb=None, bias_rgb=None,
                  skip=None, skip_up=True, want_out2=True, bf16=False, wm_next=None, split=False, ranged=True, flat=False):
    """FIR up-sampling + act -> 1x1 conv + act -> ToRGB for one up-sampling stage (see cips3d_fused_up_conv).
    wm_next (MOD_PACKED | MOD_CHAINED weights of the next stage's C -> C/2 up-conv): also returns its low-res GEMM y_next.
    A torch.bfloat16 `y_lo` selects the bf16-storage form (CIPS3D_Y_BF16, needs bf16=True): y_next is then bf16 as well.
    split: fp32-equivalent split-fp16 products; wm2_packed / wm_next must then be MOD_SPLIT16-packed.  The stage's operands are
    then split under power-of-two scales from rigorous bounds (cips3d_range): y_lo's amax array is taken from its tag or
    measured, the two layers' constants are made here (noise bounds measured on the device); out2 / y_next leave tagged.
    flat: a block that does not up-sample (CIPS3D_STAGE_FLAT): y_lo is conv1's GEMM result at the block's resolution, fir is
    not read, outputs and skip have y_lo's size (skip_up must be False)."""
    lib = _lib.load()
    B, Cc, H, W = y_lo.shape
    up = 1 if flat else 2
    if flat:
        skip_up = False          # (the skip image of a flat stage has the output's size and is added as it is)
    rg, _keep, next_amax = None, None, None
    if split and ranged:
        lc1 = range_consts(B, bias1, noise_w1, 0.0, fir=fir, noise=noise1)
        lc2 = range_consts(B, bias2, noise_w2, Cc ** 0.5, noise=noise2)
        next_amax = new_amax(B, y_lo.device) if wm_next is not None else None
        rg, _keep = _range(x_amax=amax_of(y_lo), lconst=lc1, lconst2=lc2, next_amax=next_amax)
    dev = y_lo.device
    ydt = y_lo.dtype
    if ydt == torch.bfloat16 and not bf16:
        raise RuntimeError("a bf16 y_lo needs the bf16 GEMM mode (bf16=True)")
    out2 = torch.empty(B, Cc, up * H, up * W, device=dev) if want_out2 else None
    rgb = torch.empty(B, 3, up * H, up * W, device=dev) if wm_rgb is not None else None
    y_next = torch.empty(B, Cc // 2, up * H, up * W, device=dev, dtype=ydt) if wm_next is not None else None

    def bs(nz):
        return up * up * H * W if (nz is not None and nz.shape[0] == B and B > 1) else 0

    check(lib.cips3d_fused_up_conv_next(dev_ptr(y_lo, "y_lo", dtype=ydt), dev_ptr(fir, "fir", flat), dev_ptr(noise1, "noise1", True), bs(noise1),
                                        dev_ptr(noise_w1, "noise_w1", True), dev_ptr(bias1, "bias1"), dev_ptr(wm2_packed, "wm2"),
                                        dev_ptr(noise2, "noise2", True), bs(noise2), dev_ptr(noise_w2, "noise_w2", True),
                                        dev_ptr(bias2, "bias2"), dev_ptr(out2, "out2", True), dev_ptr(wm_rgb, "wm_rgb", True),
                                        dev_ptr(bias_rgb, "bias_rgb", True), dev_ptr(skip, "skip", True),
                                        int(bool(skip_up)) | (GEMM_BF16 if bf16 else 0) |
                                        (Y_BF16 if ydt == torch.bfloat16 else 0) | (GEMM_SPLIT if split else 0) | (STAGE_FLAT if flat else 0),
                                        dev_ptr(rgb, "rgb", True), dev_ptr(wm_next, "wm_next", True),
                                        dev_ptr(y_next, "y_next", True, dtype=ydt), B, Cc, H, W,
                                        C.byref(rg) if rg is not None else None, stream_ptr()), "cips3d_fused_up_conv_next")
    if next_amax is not None:
        tag_amax(y_next, next_amax)
    if wm_next is not None:
        return out2, rgb, y_next
    return out2, rgb


def rgb_to_uint8(rgb, out=None):
    """[-1,1] float image -> uint8 (clamp, scale, round to nearest) on the device; `out`: a contiguous uint8 tensor of the
    same shape to write into (a slice of a sequence's frame buffer: multiview.sample_multi_view)."""
    lib = _lib.load()
    x = rgb.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    elif out.dtype != torch.uint8 or out.shape != x.shape or not out.is_contiguous() or out.device != x.device:
        raise RuntimeError("rgb_to_uint8: `out` must be a contiguous uint8 tensor of the input's shape on its device")
    check(lib.cips3d_rgb_to_uint8(dev_ptr(x, "rgb"), out.data_ptr(), x.numel(), stream_ptr()), "cips3d_rgb_to_uint8")
    return out


# ---------------------------------------------------------------------------------------------- backward entry points
def linear_bwd(x, W, dout, out=None, w_scale=1.0, b_scale=1.0, lrelu=False, act_gain=1.0, out_scale=1.0, need_dx=True,
               need_dW=True, need_db=True):
    lib = _lib.load()
    B, in_dim = x.shape
    out_dim = W.shape[0]
    dev = x.device
    dx = torch.empty(B, in_dim, device=dev) if need_dx else None
    dW = torch.empty(out_dim, in_dim, device=dev) if need_dW else None
    db = torch.empty(out_dim, device=dev) if need_db else None
    check(lib.cips3d_linear_bwd(dev_ptr(x, "x"), x.stride(0), dev_ptr(W, "W"), dev_ptr(out, "out", True),
                                out.stride(0) if out is not None else 0, dev_ptr(dout, "dout"), dout.stride(0), B, in_dim,
                                out_dim, w_scale, b_scale, int(lrelu), act_gain, out_scale, dev_ptr(dx, "dx", True), in_dim,
                                dev_ptr(dW, "dW", True), dev_ptr(db, "db", True), stream_ptr()), "cips3d_linear_bwd")
    return dx, dW, db


def modulate_bwd(dwm, W, s, Cout, Cin, ksq, scale, demodulate, need_dW=True):
    """dwm [B,Cout,Cin*ksq] is consumed (scratch).  Returns (dW shaped like W or None, ds [B,Cin])."""
    lib = _lib.load()
    B = s.shape[0]
    dW = torch.empty_like(W) if need_dW else None
    ds = torch.empty(B, Cin, device=W.device)
    check(lib.cips3d_modulate_bwd(dev_ptr(dwm, "dwm"), dev_ptr(W, "W"), dev_ptr(s, "s"), s.stride(0), B, Cout, Cin, ksq,
                                  float(scale), int(bool(demodulate)), dev_ptr(dW, "dW", True), dev_ptr(ds), Cin, stream_ptr()),
          "cips3d_modulate_bwd")
    return dW, ds


# The GEMMs of the differentiable path whose B operand is an ACTIVATION (decoder forward, the materialised NeRF recompute) run
# in the fp32-equivalent split-fp16 mode, as the inference forward's stand-alone GEMMs do; 0 = the fp32 MFMA (A/B knob).  The
# data-gradient GEMMs keep the fp32 MFMA: their B operand is a gradient tensor whose scale is arbitrary (1e-6 and below), and
# the fp16 halves of the split have an absolute floor of 2^-25 -- fine for O(1) activations, 5 bits for a 1e-6 gradient.
SPLIT_BACKWARD = os.environ.get("CIPS3D_SPLIT_BACKWARD", "1") != "0"


def pack_weights(wm, transpose=False, split=False):
    """wm [B,M,K] -> packed A fragments of wm (or of wm^T); split: fp16 hi + lo fragments for modconv1x1(split=True)."""
    lib = _lib.load()
    B, M, K = wm.shape
    out = torch.empty(B * M * K, device=wm.device)
    check(lib.cips3d_pack_weights(dev_ptr(wm, "wm"), dev_ptr(out), B, M, K, int(bool(transpose)) | (2 if split else 0),
                                  stream_ptr()), "cips3d_pack_weights")
    return out


def gemm_wgrad(dy, x):
    """dy [B,M,...], x [B,K,...] (same trailing pixels) -> dwm [B,M,K]."""
    lib = _lib.load()
    B, M = dy.shape[:2]
    K = x.shape[1]
    P = dy[0, 0].numel()
    dwm = torch.empty(B, M, K, device=dy.device)
    check(lib.cips3d_gemm_wgrad(dev_ptr(dy, "dy"), dev_ptr(x, "x"), dev_ptr(dwm), B, M, K, P, stream_ptr()), "cips3d_gemm_wgrad")
    return dwm


def modconv1x1_actbwd(g, wm_t_packed, y, noise=None, rgb_w=None, drgb=None, split=False, g_amax=None, d_bias=None,
                      d_noise_w=None, d_rgb_w=None, track=True):
    """The data-gradient GEMM with the previous layer's activation backward as its epilogue (cips3d_modconv1x1_actbwd):
    g [B,Cin,H,W], wm_t_packed = pack of wm^T [B,Cout,Cin], y [B,Cout,H,W] the previous layer's stored output -> dpre
    [B,Cout,H,W]; the given accumulators (zeroed by the caller) are added to.  Returns (dpre, amax array of dpre or None)."""
    lib = _lib.load()
    B, Cin, H, W = g.shape
    Cout = y.shape[1]
    out = torch.empty_like(y)
    out_amax = new_amax(B, g.device) if track else None
    rg, _keep = _range(x_amax=g_amax if split else None, out_amax=out_amax)
    ab = _lib.ActBwd()
    ab.y, ab.rgb_w, ab.drgb = dev_ptr(y, "y"), dev_ptr(rgb_w, "rgb_w", True), dev_ptr(drgb, "drgb", True)
    ab.d_bias, ab.d_noise_w, ab.d_rgb_w = dev_ptr(d_bias, "d_bias", True), dev_ptr(d_noise_w, "d_noise_w", True), dev_ptr(d_rgb_w, "d_rgb_w", True)
    nb = H * W if (noise is not None and noise.shape[0] == B and B > 1) else 0
    check(lib.cips3d_modconv1x1_actbwd(dev_ptr(g, "g"), dev_ptr(wm_t_packed, "wm_t"), dev_ptr(out), B, Cin, Cout, H * W,
                                       GEMM_SPLIT if split else 0, C.byref(ab), dev_ptr(noise, "noise", True), nb, C.byref(rg),
                                       stream_ptr()), "cips3d_modconv1x1_actbwd")
    return out, out_amax


def gemm_wgrad_split(dy, x, dy_amax=None, x_amax=None, out=None):
    """gemm_wgrad on split-fp16 products; dy_amax / x_amax: per-sample amax slot arrays (new_amax / absmax) or None.
    out: accumulate into this [B,M,K] tensor instead of returning a fresh one."""
    lib = _lib.load()
    B, M = dy.shape[:2]
    K = x.shape[1]
    P = dy[0, 0].numel()
    dwm = out if out is not None else torch.empty(B, M, K, device=dy.device)
    check(lib.cips3d_gemm_wgrad_split(dev_ptr(dy, "dy"), dev_ptr(x, "x"), dev_ptr(dwm), B, M, K, P, dev_ptr(dy_amax, "dy_amax", True),
                                      dev_ptr(x_amax, "x_amax", True), 1 if out is not None else 0, stream_ptr()),
          "cips3d_gemm_wgrad_split")
    return dwm


def noise_bias_act_bwd(dy, y, noise, noise_w, need_dnoise=False, need_dnw=True, need_db=True):
    lib = _lib.load()
    B, Cc = y.shape[:2]
    HW = y[0, 0].numel()
    dev = y.device
    dx = torch.empty_like(y)
    nb = HW if (noise is not None and noise.shape[0] == B and B > 1) else 0
    dnoise = torch.empty_like(noise) if (need_dnoise and noise is not None) else None
    dnw = torch.empty(1, device=dev) if (need_dnw and noise is not None) else None
    if need_db and dnw is not None:       # back to back: the library zeroes both with one memset
        both = torch.empty(2 * Cc, device=dev)
        db, scratch = both[:Cc], both[Cc:]
    else:
        db = torch.empty(Cc, device=dev) if need_db else None
        scratch = torch.empty(Cc, device=dev) if dnw is not None else None
    check(lib.cips3d_noise_bias_act_bwd(dev_ptr(dy, "dy"), dev_ptr(y, "y"), dev_ptr(noise, "noise", True), nb,
                                        dev_ptr(noise_w, "noise_w", True), dev_ptr(dx), dev_ptr(dnoise, "dnoise", True),
                                        dev_ptr(dnw, "dnw", True), dev_ptr(db, "db", True), dev_ptr(scratch, "scratch", True),
                                        B, Cc, HW, stream_ptr()),
          "cips3d_noise_bias_act_bwd")
    return dx, dnoise, dnw, db


def torgb_bwd(drgb, x, wm, need_db=True):
    lib = _lib.load()
    B, Cc = x.shape[:2]
    HW = x[0, 0].numel()
    dx = torch.empty_like(x)
    both = torch.empty(B * 3 * Cc + 3, device=x.device)      # dwm and dbias back to back: one memset in the library
    dwm = both[:B * 3 * Cc].view(B, 3, Cc)
    db = both[B * 3 * Cc:] if need_db else None
    check(lib.cips3d_torgb_bwd(dev_ptr(drgb, "drgb"), dev_ptr(x, "x"), dev_ptr(wm, "wm"), dev_ptr(dx), dev_ptr(dwm),
                               dev_ptr(db, "db", True), B, Cc, HW, stream_ptr()), "cips3d_torgb_bwd")
    return dx, dwm, db


def camera_params_bwd(locations, dextr, up=None):
    lib = _lib.load()
    B = locations.shape[0]
    loc = locations.detach().float().contiguous()
    up_t = up.detach().float().contiguous() if up is not None else None
    dloc = torch.empty(B, 2, device=loc.device)
    dextr = dextr.contiguous()
    check(lib.cips3d_camera_params_bwd(dev_ptr(loc, "locations"), dev_ptr(up_t, "up", True), dev_ptr(dextr, "dextr"),
                                       B, dev_ptr(dloc), stream_ptr()), "cips3d_camera_params_bwd")
    return dloc


def nerf_backward(net, sigmoid_beta, cam_poses, focals, near, far, perturb_u, film, layer_bias, img_size, n_samples,
                  static_viewdirs, d_features, d_thumb, need_params=False):
    """The materialised NeRF backward (see csrc/nerf_bwd.hip): returns (dfilm [B,L,2,H], dcam [B,3,4]) -- and, with
    need_params (`optim_render_params`, models/projector_v10.py:848-872), a dict of the gradients of the renderer's own
    weights: pts_linears.{l}.weight / .bias, views_linears.weight / .bias, rgb_linear.*, sigma_linear.*, sigmoid_beta
    (the gamma / beta heads get theirs from the FiLM table's backward).

    net = SirenGenerator (weights), film [B,L,2,H], layer_bias [L,H]; d_features [B,H,S,S], d_thumb [B,3,S,S]."""
    lib = _lib.load()
    st = stream_ptr()
    dev = cam_poses.device
    B = cam_poses.shape[0]
    H = net.W
    D = net.D
    L = D + 1
    R = img_size * img_size
    P = R * n_samples
    fb = L * 2 * H
    geom = _lib.NerfBwdGeom()
    keep = [cam_poses.float().contiguous(), focals.float().reshape(B).contiguous(), near.float().reshape(B).contiguous(),
            far.float().reshape(B).contiguous(),
            None if perturb_u is None else perturb_u.float().reshape(B, R).contiguous()]
    geom.cam_poses, geom.focals, geom.near_, geom.far_ = (dev_ptr(t) for t in keep[:4])
    geom.perturb_u = dev_ptr(keep[4], "perturb_u", True)
    geom.B, geom.img_size, geom.n_samples, geom.static_viewdirs = B, img_size, n_samples, int(bool(static_viewdirs))
    gp = C.byref(geom)
    new = lambda *shape: torch.empty(*shape, device=dev)
    film = film.contiguous()
    film_l = lambda l: film.data_ptr() + 4 * (l * 2 * H)

    # ---- forward recompute
    ptsn, viewdirs = new(B, 3, P), new(B, 3, R)
    pre = [new(B, H, P)] + [None] * (L - 1)
    hh = [new(B, H, P) for _ in range(L)]
    w_first = net.pts_linears[0].weight
    check(lib.cips3d_nerf_bwd_points(gp, dev_ptr(w_first), layer_bias.data_ptr(), film_l(0), fb, H, dev_ptr(ptsn),
                                     dev_ptr(pre[0]), dev_ptr(hh[0]), dev_ptr(viewdirs), st), "cips3d_nerf_bwd_points")
    w_view = net.views_linears.weight                      # [H, H+3]
    mats = [net.pts_linears[l].weight.detach() for l in range(1, D)] + [w_view.detach()[:, :H].contiguous()]
    split = SPLIT_BACKWARD and H % 32 == 0
    packed, packed_t = [], []
    for Wl in mats:                                        # shared over the batch: replicate the (small) matrix
        Wb = Wl.unsqueeze(0).expand(B, H, H).contiguous()
        packed.append(pack_weights(Wb, split=split))            # forward recompute: activations in [-1, 1]
        packed_t.append(pack_weights(Wb, transpose=True))       # data gradients: fp32 MFMA (see SPLIT_BACKWARD)

    sine_amax = const_amax(B, 1.0, dev)         # the recompute's GEMM inputs are FiLM sines

    def gemm(x, pk, sp=False):
        return modconv1x1(x.view(B, H, P, 1), pk, H, epilogue=0, split=sp, x_amax=sine_amax, track=False).view(B, H, P)

    for l in range(1, L):
        acc = gemm(hh[l - 1], packed[l - 1], split)
        is_view = l == D
        check(lib.cips3d_nerf_bwd_film(dev_ptr(acc), dev_ptr(hh[l]), layer_bias.data_ptr() + 4 * l * H, film_l(l), fb,
                                       (w_view.data_ptr() + 4 * H) if is_view else None, H + 3,
                                       dev_ptr(viewdirs) if is_view else None, B, H, R, P, st), "cips3d_nerf_bwd_film")
        pre[l] = acc
    h_last, f = hh[D - 1], hh[D]
    sdf, crgb, g = new(B, 1, P), new(B, 3, P), new(B, P)
    check(lib.cips3d_nerf_bwd_heads(dev_ptr(h_last), dev_ptr(net.sigma_linear.weight), H, 1, dev_ptr(net.sigma_linear.bias), 1,
                                    B, H, P, dev_ptr(sdf), st), "cips3d_nerf_bwd_heads")
    check(lib.cips3d_nerf_bwd_heads(dev_ptr(f), dev_ptr(net.rgb_linear.weight), H, 1, dev_ptr(net.rgb_linear.bias), 3, B, H, P,
                                    dev_ptr(crgb), st), "cips3d_nerf_bwd_heads")
    dF = d_features.contiguous().view(B, H, R)
    dth = d_thumb.contiguous().view(B, 3, R)
    check(lib.cips3d_nerf_bwd_dot(dev_ptr(dF), dev_ptr(f), B, H, R, P, dev_ptr(g), st), "cips3d_nerf_bwd_dot")

    # ---- compositing backward
    wts, Tb, dsdf, dcrgb, ddnorm = new(B, P), new(B, P), new(B, P), new(B, 3, P), new(B, R)
    dbeta_ray = new(B, R) if need_params else None
    check(lib.cips3d_nerf_bwd_composite(gp, dev_ptr(sdf), dev_ptr(crgb), dev_ptr(g), dev_ptr(dth),
                                        dev_ptr(sigmoid_beta, "sigmoid_beta", True),      # None: with_sdf = False
                                        dev_ptr(wts), dev_ptr(Tb), dev_ptr(dsdf), dev_ptr(dcrgb), dev_ptr(ddnorm),
                                        dev_ptr(dbeta_ray, "dbeta_ray", True), st), "cips3d_nerf_bwd_composite")

    # ---- gradients of the renderer's own weights (optional): the wide blocks W_l [H,H] on the split-fp16 weight-gradient
    # GEMM (contraction over the points), everything narrow -- biases, the first layer, the view-direction columns, the heads
    # -- as row dots
    pg = {}

    def row_dots(a, rows, Pa, x=None, nx=0, Px=0):
        out = torch.zeros(rows, 4, device=dev)
        check(lib.cips3d_nerf_bwd_row_dots(dev_ptr(a), dev_ptr(x, "x", True), nx, Px, dev_ptr(out), B, rows, Pa, st),
              "cips3d_nerf_bwd_row_dots")
        return out

    def wide(dp, h):                 # sum_b dp[b] h[b]^T
        if H % 32 or P % 32:
            return gemm_wgrad(dp, h).sum(0)
        acc = torch.zeros(1, H, H, device=dev)
        am = absmax(dp)
        for b in range(B):
            gemm_wgrad_split(dp[b:b + 1], h[b:b + 1], am[b:b + 1], sine_amax[:1], out=acc)
        return acc[0]

    if need_params:
        t = row_dots(f, H, P, dcrgb, 3, P)
        pg["rgb_linear.weight"] = t[:, :3].t().contiguous()
        pg["rgb_linear.bias"] = row_dots(dcrgb, 3, P)[:, 3].contiguous()
        pg["sigma_linear.weight"] = row_dots(h_last, H, P, dsdf, 1, P)[:, :1].t().contiguous()
        pg["sigma_linear.bias"] = row_dots(dsdf, 1, P)[:, 3].contiguous()
        pg["sigmoid_beta"] = row_dots(dbeta_ray, 1, R)[:, 3].contiguous() if sigmoid_beta is not None else torch.zeros(1, device=dev)

    # ---- MLP backward
    dfilm = torch.zeros(B, L, 2, H, device=dev)
    dfilm_l = lambda l: dfilm.data_ptr() + 4 * (l * 2 * H)
    check(lib.cips3d_nerf_bwd_film_grad(dev_ptr(f), dev_ptr(pre[D]), film_l(D), fb, 1, None, None, dev_ptr(wts), dev_ptr(dF),
                                        dev_ptr(net.rgb_linear.weight), dev_ptr(dcrgb), dfilm_l(D), B, H, R, P, st),
          "cips3d_nerf_bwd_film_grad")
    dpre = f
    dvd_pt = new(B, 3, P)
    check(lib.cips3d_nerf_bwd_heads(dev_ptr(dpre), w_view.data_ptr() + 4 * H, 1, H + 3, None, 3, B, H, P, dev_ptr(dvd_pt), st),
          "cips3d_nerf_bwd_heads")
    if need_params:                                        # the view layer: [h_{D-1} | view direction] -> H
        t = row_dots(dpre, H, P, viewdirs, 3, R)
        pg["views_linears.weight"] = torch.cat([wide(dpre, hh[D - 1]), t[:, :3]], 1)
        pg["views_linears.bias"] = t[:, 3].contiguous()
    for l in range(D, 0, -1):
        dh = gemm(dpre, packed_t[l - 1])                  # gradient w.r.t. h_{l-1}
        first = l == D                                     # h_{D-1} also feeds the sigma head
        check(lib.cips3d_nerf_bwd_film_grad(dev_ptr(dh), dev_ptr(pre[l - 1]), film_l(l - 1), fb, 0,
                                            dev_ptr(net.sigma_linear.weight) if first else None,
                                            dev_ptr(dsdf) if first else None, None, None, None, None, dfilm_l(l - 1), B, H, R,
                                            P, st), "cips3d_nerf_bwd_film_grad")
        dpre = dh                                          # now d(pre_{l-1})
        if need_params:
            if l - 1 >= 1:
                pg[f"pts_linears.{l - 1}.weight"] = wide(dpre, hh[l - 2])
                pg[f"pts_linears.{l - 1}.bias"] = row_dots(dpre, H, P)[:, 3].contiguous()
            else:
                t = row_dots(dpre, H, P, ptsn, 3, P)
                pg["pts_linears.0.weight"] = t[:, :3].contiguous()
                pg["pts_linears.0.bias"] = t[:, 3].contiguous()
    dptsn = new(B, 3, P)
    check(lib.cips3d_nerf_bwd_heads(dev_ptr(dpre), dev_ptr(w_first), 1, 3, None, 3, B, H, P, dev_ptr(dptsn), st),
          "cips3d_nerf_bwd_heads")
    dcam = new(B, 3, 4)
    check(lib.cips3d_nerf_bwd_camera(gp, dev_ptr(dptsn), dev_ptr(dvd_pt), dev_ptr(ddnorm), dev_ptr(dcam), st),
          "cips3d_nerf_bwd_camera")
    if need_params:
        return dfilm, dcam, pg
    return dfilm, dcam


def nerf_pack_weights_t(w_hidden, w_view, packed, hidden, depth):
    """The transposed weight stream of the fused NeRF backward (consumption order, forward scales; cips3d_nerf_pack_weights_t)."""
    lib = _lib.load()
    packed_t = torch.empty_like(packed)
    check(lib.cips3d_nerf_pack_weights_t(dev_ptr(w_hidden, "w_hidden", True), dev_ptr(w_view, "w_view"), dev_ptr(packed),
                                         dev_ptr(packed_t), hidden, depth, stream_ptr()), "cips3d_nerf_pack_weights_t")
    return packed_t


FUSED_NERF_BACKWARD = os.environ.get("CIPS3D_FUSED_NERF_BACKWARD", "1") != "0"   # A/B knob: 0 = the materialised sequence
STASH_IN_FORWARD = os.environ.get("CIPS3D_STASH_IN_FORWARD", "1") != "0"         # A/B knob: 0 = the backward recomputes the forward


def nerf_backward_fused_supported(hidden, depth, img_size, n_samples):
    return bool(_lib.load().cips3d_nerf_bwd_fused_supported(hidden, depth, img_size, n_samples))


def nerf_forward_stash(B, img_size, n_samples, hidden, depth, device, n_chunks=None):
    """Buffers a differentiable forward hands to cips3d_nerf_render (`stash`, `bwd_sdf`, `bwd_crgb`) and later, as `fwd`, to
    nerf_backward_fused: dict(stash, sdf, crgb, n_chunks)."""
    lib = _lib.load()
    if n_chunks is None:
        n_chunks = nerf_suggest_chunks(B, img_size, n_samples)
    P = img_size * img_size * n_samples
    return {"stash": torch.empty(int(lib.cips3d_nerf_bwd_fused_stash_floats(B, img_size, n_samples, hidden, depth, n_chunks)),
                                 device=device),
            "sdf": torch.empty(B, P, device=device), "crgb": torch.empty(B, 3, P, device=device), "n_chunks": n_chunks}


def nerf_backward_fused(net, sigmoid_beta, cam_poses, focals, near, far, perturb_u, film, layer_bias, packed, packed_t, img_size,
                        n_samples, static_viewdirs, d_features, d_thumb, fwd=None):
    """The fused NeRF backward (csrc/nerf_bwd_fused.hip): returns (dfilm [B,L,2,H], dcam [B,3,4]); same contract as
    nerf_backward.  `fwd` = the nerf_forward_stash buffers a differentiable forward filled: the forward is then not
    recomputed."""
    lib = _lib.load()
    dev = cam_poses.device
    B, H, D = cam_poses.shape[0], net.W, net.D
    L, R = D + 1, img_size * img_size
    p = _lib.NerfBwdFusedParams()
    keep = [cam_poses.float().contiguous(), focals.float().reshape(B).contiguous(), near.float().reshape(B).contiguous(),
            far.float().reshape(B).contiguous(),
            None if perturb_u is None else perturb_u.float().reshape(B, R).contiguous(),
            film.float().contiguous(), d_features.float().contiguous(), d_thumb.float().contiguous()]
    g = p.geom
    g.cam_poses, g.focals, g.near_, g.far_ = (dev_ptr(t) for t in keep[:4])
    g.perturb_u = dev_ptr(keep[4], "perturb_u", True)
    g.B, g.img_size, g.n_samples, g.static_viewdirs = B, img_size, n_samples, int(bool(static_viewdirs))
    if fwd is not None:
        n_chunks, stash = fwd["n_chunks"], fwd["stash"]
        p.fwd_sdf, p.fwd_crgb = dev_ptr(fwd["sdf"]), dev_ptr(fwd["crgb"])
    else:
        n_chunks = nerf_suggest_chunks(B, img_size, n_samples)
        stash = torch.empty(int(lib.cips3d_nerf_bwd_fused_stash_floats(B, img_size, n_samples, H, D, n_chunks)), device=dev)
    scratch = torch.empty(int(lib.cips3d_nerf_bwd_fused_scratch_floats(B, img_size, n_samples, H, D)), device=dev)
    dfilm, dcam = torch.empty(B, L, 2, H, device=dev), torch.empty(B, 3, 4, device=dev)
    p.w_first, p.packed, p.packed_t = dev_ptr(net.pts_linears[0].weight), dev_ptr(packed), dev_ptr(packed_t)
    p.w_view, p.film, p.layer_bias = dev_ptr(net.views_linears.weight), dev_ptr(keep[5]), dev_ptr(layer_bias)
    p.w_sigma, p.b_sigma = dev_ptr(net.sigma_linear.weight), dev_ptr(net.sigma_linear.bias)
    p.w_rgb, p.b_rgb = dev_ptr(net.rgb_linear.weight), dev_ptr(net.rgb_linear.bias)
    p.sigmoid_beta = dev_ptr(sigmoid_beta, "sigmoid_beta", True)      # None: with_sdf = False (raw density)
    p.d_features, p.d_thumb = dev_ptr(keep[6]), dev_ptr(keep[7])
    p.stash, p.scratch, p.dfilm, p.dcam = dev_ptr(stash), dev_ptr(scratch), dev_ptr(dfilm), dev_ptr(dcam)
    p.hidden, p.depth, p.n_chunks = H, D, n_chunks
    check(lib.cips3d_nerf_bwd_fused(C.byref(p), stream_ptr()), "cips3d_nerf_bwd_fused")
    return dfilm, dcam


def inversion_roofline(renderer, B, n_samples, img_size=64, iters=10):
    """Roofline entry of the flip-inversion step's dominant piece: the fused NeRF backward (csrc/nerf_bwd_fused.hip:
    nerf_stash_kernel + nerf_bwd_kernel, ~23 % of the step), timed stand-alone on the step's shape with HIP events around
    cips3d_nerf_bwd_fused (the three per-ray / preparation kernels inside the call are ~4 % of it).  Algorithmic flop = the
    forward point MLP once more + the data-gradient GEMMs W_l^T d(pre_l); executed as split-fp16 products, so the peak is the
    fp16 dense MFMA peak / 3 (as for the forward render kernel)."""
    from . import autograd as AG
    from .camera import Camera
    dev = "cuda"
    net = renderer.network
    H, D, S = net.W, net.D, img_size
    locs = torch.tensor([[0.25, 0.1], [-0.25, 0.1]] * ((B + 1) // 2))[:B].to(dev)
    cam, focal, near, far = Camera.generate_camera_params(locations=locs, img_size=S, device=dev, fov_ang=15,
                                                          dist_radius=0.3)[:4]
    from . import weights
    styles = (0.5 * weights.det_normal("roofline.styles", (B, D + 1, renderer.style_dim), 1.0, 1)).to(dev)
    film = AG.film_table(renderer, styles).detach()
    u = weights.det_unit_uniform("roofline.u", (B, S, S, 1), 2).to(dev)
    dF = (1e-5 * weights.det_normal("roofline.dF", (B, H, S, S), 1.0, 3)).to(dev)
    dT = (1e-4 * weights.det_normal("roofline.dT", (B, 3, S, S), 1.0, 4)).to(dev)
    packed, layer_bias = renderer._derived_buffers()
    packed_t = renderer._packed_transposed()
    run = lambda: nerf_backward_fused(net, renderer.sigmoid_beta.detach(), cam, focal, near, far, u, film, layer_bias, packed,
                                      packed_t, S, n_samples, False, dF, dT)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    P = B * S * S * n_samples
    fwd = 2.0 * 3 * H + (D - 1) * 2.0 * H * H + 2.0 * (H + 3) * H + 2.0 * H * 4
    bwd = D * 2.0 * H * H + 2.0 * 6 * H
    flop = P * (fwd + bwd)
    a = flop / (ms * 1e-3) / 1e12
    peak = 2500.0 / 3
    return {"kernel": "cips3d_nerf_bwd_fused = nerf_stash_kernel (forward recompute, accumulators stashed) + nerf_bwd_kernel "
                      "(register-resident MFMA backward); timed stand-alone on the step's shape",
            "bound": "mfma", "achieved": a, "peak": peak, "unit": "TFLOP/s", "frac": a / peak,
            "peak_definition": "fp16 dense MFMA peak 2500 TFLOP/s / 3 fp16 products per fp32 product",
            "avg_launch_ms": ms, "flop_per_launch": flop,
            "stash_bytes_per_launch": 2.0 * 4 * P * D * H,
            "note": "stash written once and read once: at this time it moves at %.2f TB/s" % (2.0 * 4 * P * D * H / (ms * 1e-3) / 1e12)}


def sqdiff_pair(a0, b0, c0, a1, b1, c1):
    """loss = c0 sum (a0 - b0)^2 + c1 sum (a1 - b1)^2 as a device scalar (cips3d_sqdiff_pair: the two squared-difference terms
    of the inversion loss, projector_v10.py:1173-1174), deterministic; a1 / b1 may be None."""
    lib = _lib.load()
    n0, n1 = a0.numel(), (a1.numel() if a1 is not None else 0)
    partial = torch.empty(2 * int(lib.cips3d_sqdiff_pair_partials(n0, n1)), device=a0.device)
    loss = torch.empty((), device=a0.device)
    check(lib.cips3d_sqdiff_pair(dev_ptr(a0, "a0"), dev_ptr(b0, "b0"), n0, float(c0), dev_ptr(a1, "a1", True), dev_ptr(b1, "b1", True),
                                 n1, float(c1), partial.data_ptr(), loss.data_ptr(), stream_ptr()), "cips3d_sqdiff_pair")
    return loss


def sqdiff_pair_bwd(a0, b0, c0, a1, b1, c1, gloss):
    """(d loss / d a0, d loss / d a1) of sqdiff_pair for the incoming gradient `gloss` (a device scalar), one launch."""
    lib = _lib.load()
    n0, n1 = a0.numel(), (a1.numel() if a1 is not None else 0)
    d0 = torch.empty_like(a0)
    d1 = torch.empty_like(a1) if a1 is not None else None
    check(lib.cips3d_sqdiff_pair_bwd(dev_ptr(a0, "a0"), dev_ptr(b0, "b0"), n0, float(c0), d0.data_ptr(), dev_ptr(a1, "a1", True),
                                     dev_ptr(b1, "b1", True), n1, float(c1), dev_ptr(d1, "d1", True), dev_ptr(gloss, "gloss"),
                                     stream_ptr()), "cips3d_sqdiff_pair_bwd")
    return d0, d1
