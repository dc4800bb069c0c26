"""Host side of the one-call decoder forward / backward (csrc/decoder_grad.hip): the whole Decoder as ONE autograd node.

`DecoderFn.apply(dec, features, styles, noise..., *parameters)` = Decoder.forward (models/model_v3.py:592-637) with the
gradients of the features, the W+ styles and every decoder parameter produced by two C calls
(cips3d_decoder_grad_forward / cips3d_decoder_grad_backward) instead of one torch.autograd.Function per op
(autograd.decoder_forward, which stays as the route for what this one does not cover: kernel_size 3, bf16 modes, noise
buffers that require gradients, batches above 4).  The plan (device pointers into the module's parameters + one workspace)
is built once per (batch, input size) and re-made when a parameter is re-allocated.
"""
import ctypes as C
import weakref

import torch
from torch.autograd import Function

from . import _lib, hip
from ._lib import dev_ptr

MAX_LAYERS = 48


class ModBwdDesc(C.Structure):
    _fields_ = [("d_wm", C.c_void_p), ("W", C.c_void_p), ("s", C.c_void_p), ("dW", C.c_void_p), ("ds", C.c_void_p),
                ("s_stride", C.c_int64), ("ds_stride", C.c_int64), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("scale", C.c_float), ("demodulate", C.c_int32), ("row_begin", C.c_int32), ("pad_", C.c_int32)]


class GradLayer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("noise_index", C.c_int32), ("flags", C.c_int32), ("pad_", C.c_int32),
                ("bias", C.c_void_p), ("noise_w", C.c_void_p), ("fir", C.c_void_p), ("wm", C.c_void_p), ("wm_t", C.c_void_p),
                ("y", C.c_void_p), ("y_amax", C.c_void_p), ("g_amax", C.c_void_p), ("glo_amax", C.c_void_p),
                ("d_wm", C.c_void_p), ("d_bias", C.c_void_p), ("d_nw_part", C.c_void_p),
                ("slots", C.c_int32), ("slot_stride", C.c_int32), ("rgb_slot_stride", C.c_int32), ("pad2_", C.c_int32)]


class SlotJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int32), ("slots", C.c_int32), ("stride", C.c_int32),
                ("row_begin", C.c_int32)]


class DecoderGradPlan(C.Structure):
    _fields_ = [("B", C.c_int32), ("n_layers", C.c_int32), ("style_dim", C.c_int32), ("pad_", C.c_int32),
                ("style_table", C.c_void_p), ("style_n", C.c_int32), ("style_rows", C.c_int32),
                ("mod_table", C.c_void_p), ("mod_n", C.c_int32), ("mod_rows", C.c_int32),
                ("modbwd_table", C.c_void_p), ("modbwd_n", C.c_int32), ("modbwd_blocks", C.c_int32),
                ("style_w_offsets", C.c_void_p), ("styles", C.c_void_p), ("s_all", C.c_void_p), ("ds_all", C.c_void_p),
                ("d_styles", C.c_void_p), ("d_style_W", C.c_void_p), ("d_style_b", C.c_void_p),
                ("amax_base", C.c_void_p), ("amax_bytes", C.c_int64), ("zero_base", C.c_void_p), ("zero_bytes", C.c_int64),
                ("feat_amax", C.c_void_p), ("y_lo", C.c_void_p), ("g", C.c_void_p * 2), ("g_lo", C.c_void_p),
                ("drgb_lo", C.c_void_p * 4), ("rgb", C.c_void_p * 2),
                ("nw_parts", C.c_void_p), ("nw_stride", C.c_int32), ("pad2_", C.c_int32),
                ("slot_table", C.c_void_p), ("slot_n", C.c_int32), ("slot_blocks", C.c_int32), ("d_noise_w", C.c_void_p),
                ("layers", GradLayer * MAX_LAYERS)]


class DecoderGradIO(C.Structure):
    _fields_ = [("features", C.c_void_p), ("noise", C.c_void_p * MAX_LAYERS), ("noise_bstride", C.c_int64 * MAX_LAYERS),
                ("rgb", C.c_void_p), ("d_rgb", C.c_void_p), ("d_features", C.c_void_p),
                ("style_table", C.c_void_p), ("styles", C.c_void_p)]


class Unsupported(RuntimeError):
    """This decoder / call shape is not covered by the one-call path (autograd.decoder_forward takes it)."""


def _upload(array, dev):
    return torch.frombuffer(bytearray(bytes(memoryview(array))), dtype=torch.uint8).to(dev)


def parameters_of(dec):
    """The decoder parameters the node differentiates, in the order DecoderFn takes them and returns their gradients."""
    from .decoder import StyledConv
    out = []
    for m, _ in dec._mod_layers():
        out += [m.conv.weight, m.conv.modulation.weight, m.conv.modulation.bias]
        out += [m.noise.weight, m.activate.bias] if isinstance(m, StyledConv) else [m.bias]
    return out


class GradPlan:
    def __init__(self, dec, B, H0, W0, device):
        from .decoder import StyledConv
        lib = _lib.load()
        if lib.cips3d_sizeof_grad_plan() != C.sizeof(DecoderGradPlan) or lib.cips3d_sizeof_grad_io() != C.sizeof(DecoderGradIO):
            raise RuntimeError("cips3d_decoder_grad_plan / _io layout mismatch between python and the library")
        seq = dec._mod_layers()
        if dec.kernel_size != 1 or len(seq) > MAX_LAYERS or B > 4 or dec.style_dim % 4:
            raise Unsupported("kernel_size != 1, too many layers, batch > 4 or style_dim % 4")
        # (no reference to `dec` itself: the plans live in a WeakKeyDictionary keyed by it, and a value that holds its key is never
        # collected -- one ~0.6 GB workspace leaked per copy.deepcopy(G) of the projector)
        self.B, self.dev, self.n_latent, self.style_dim = B, device, dec.n_latent, dec.style_dim
        self.key = self.weights_key(dec)
        AF = hip.amax_floats()
        sd = dec.style_dim

        # ---- geometry of every layer
        info, H, W, n_conv = [], H0, W0, 0
        for m, latent in seq:
            conv = m.conv
            if isinstance(m, StyledConv):
                if getattr(m, "bf16", False):
                    raise Unsupported("bf16 decoder modes take the per-op route")
                if not hip.modconv1x1_supported(conv.in_channel, conv.out_channel, H * W) or conv.in_channel > 512 or \
                        conv.in_channel % 32 or conv.out_channel % 32 or W % 4:
                    raise Unsupported(f"StyledConv {conv.in_channel}->{conv.out_channel} at {H}x{W}")
                up = bool(conv.upsample)
                info.append(dict(m=m, latent=latent, kind=1 if up else 0, Cin=conv.in_channel, Cout=conv.out_channel, H=H, W=W,
                                 Ho=H * (2 if up else 1), Wo=W * (2 if up else 1), split=bool(getattr(m, "split", True)),
                                 conv_i=n_conv))
                n_conv += 1
                H, W = info[-1]["Ho"], info[-1]["Wo"]
            else:
                up = bool(m.upsample) and len([i for i in info if i["kind"] >= 2]) > 0
                if conv.in_channel > 512:
                    raise Unsupported("ToRGB wider than 512")
                info.append(dict(m=m, latent=latent, kind=3 if up else 2, Cin=conv.in_channel, Cout=3, H=H, W=W, Ho=H, Wo=W))
        if info[-1]["kind"] < 2 or info[0]["kind"] >= 2:
            raise Unsupported("a decoder starts with a StyledConv and ends with a ToRGB")
        self.info, self.n_conv = info, n_conv
        self.out_hw = (H, W)
        # Slotted accumulators (cips3d_actbwd): the kernel that makes the gradient w.r.t. a StyledConv's pre-activation runs one
        # workgroup per 64 (128 output rows) or 128 pixels (the tail kernel: 1024) and sample; at most ~128 of them add into one copy
        pad32 = lambda n: (n + 31) // 32 * 32                       # noqa: E731
        convs = [i for i in info if i["kind"] < 2]
        for k, i in enumerate(convs):
            last = k == len(convs) - 1
            px = 1024 if last else (64 if i["Cout"] == 128 else 128)
            wgs = (i["Ho"] * i["Wo"] + px - 1) // px * B
            if last:          # the tail kernel's workgroups own ONE channel each: 32 of them add into the same 128-byte line
                wgs *= min(32, i["Cout"])
            S = 1
            while S < 16 and wgs > 128 * S:
                S *= 2
            i["S"], i["cpad"] = S, pad32(i["Cout"])
        for k, i in enumerate(info):
            if i["kind"] >= 2:
                i["S"], i["cpad"] = info[k - 1]["S"], pad32(B * 3 * i["Cin"])

        # ---- workspace layout (floats)
        off = [0]

        def take(n, align=64):
            o = (off[0] + align - 1) // align * align
            off[0] = o + n
            return o

        amax0 = take(0)
        self.o_feat_amax = take(B * AF)
        for i in info:
            if i["kind"] < 2:
                i["o_y_amax"] = take(B * AF)
        amax1 = take(0)
        zero0 = take(0)
        # the gradients' maxima are raised by the BACKWARD's kernels: their rows belong to the block the backward zeroes (in the
        # forward's block a second backward over a retained graph scaled its split-fp16 operands with the previous backward's
        # maxima -- lo halves underflow when the new d_rgb is orders of magnitude smaller)
        for i in info:
            if i["kind"] < 2:
                i["o_g_amax"] = take(B * AF)
                i["o_glo_amax"] = take(B * AF) if i["kind"] == 1 else None
        for i in info:
            i["o_d_wm"] = take(B * i["Cout"] * i["Cin"]) if i["kind"] < 2 else take(i["S"] * i["cpad"])
            i["o_d_bias_part"] = take(i["S"] * i["cpad"]) if (i["kind"] < 2 and i["S"] > 1) else None
        tot_cin = sum(i["Cin"] for i in info)
        o_ds_all = take(B * tot_cin)
        self.nw_stride = max(i["S"] * i["cpad"] for i in info if i["kind"] < 2)
        o_nw_parts = take(n_conv * self.nw_stride)
        out0 = take(0)                                     # ---- from here: what the backward returns (cloned once per step)
        for i in info:
            i["o_d_bias"] = take(i["Cout"], 4)
        o_d_styles = take(B * dec.n_latent * sd)
        zero1 = take(0)
        for i in info:
            i["o_dW"] = take(i["Cout"] * i["Cin"], 4)
            i["o_dmW"] = take(i["Cin"] * sd, 4)            # the style head's weight gradient (cips3d_linear_table_bwd layout)
        o_d_style_b = take(tot_cin, 4)
        o_d_noise_w = take(n_conv, 4)
        out1 = take(0)
        self.out_range = (out0, out1)
        o_styles = take(B * dec.n_latent * sd)
        o_s_all = take(B * tot_cin)
        for i in info:
            n = B * i["Cout"] * i["Cin"]
            i["o_wm"] = take(n)
            i["o_wm_t"] = take(n) if i["kind"] < 2 else None
            if i["kind"] < 2:
                i["o_y"] = take(B * i["Cout"] * i["Ho"] * i["Wo"])
        ups = [i for i in info if i["kind"] == 1]
        max_lo = max([B * i["Cout"] * i["H"] * i["W"] for i in ups], default=0)
        o_y_lo = take(max_lo) if ups else None
        o_g_lo = take(max_lo) if ups else None
        max_act = max(B * i["Cout"] * i["Ho"] * i["Wo"] for i in info if i["kind"] < 2)
        o_g = [take(max_act), take(max_act)]
        rgb_ups = [i for i in info if i["kind"] == 3]
        if len(rgb_ups) > 4:
            raise Unsupported("more than four resolutions below the output")
        o_drgb_lo = [take(B * 3 * (i["H"] // 2) * (i["W"] // 2)) for i in reversed(rgb_ups)]
        o_rgb = [take(B * 3 * H * W), take(B * 3 * H * W)]
        self.ws = torch.empty(take(0), device=device)
        base = self.ws.data_ptr()
        P = lambda o: None if o is None else base + 4 * o           # noqa: E731
        self._view = lambda o, n: self.ws[o:o + n]                  # noqa: E731

        # ---- style table (style -> s of every layer) and its backward offsets
        self.styles_buf = self._view(o_styles, B * dec.n_latent * sd).view(B, dec.n_latent, sd)
        tab = hip.LinearTable(device)
        so = 0
        woffs = []
        for i in info:
            mod = i["m"].conv.modulation
            i["o_s"] = o_s_all + so
            tab.add(mod.weight, mod.bias, self.styles_buf, dec.n_latent * sd, self.ws, i["Cin"], w_scale=mod.scale, b_scale=mod.lr_mul,
                    x_offset=i["latent"] * sd, out_offset=i["o_s"])
            woffs.append(i["o_dmW"] - info[0]["o_dmW"])
            so += B * i["Cin"]
        tab._upload()
        self._style_tab = tab
        self._woffs = torch.tensor(woffs, dtype=torch.int64).to(device)

        # ---- modulation tables
        mdescs, bdescs, rows, blocks = [], [], 0, 0
        for i in info:
            conv = i["m"].conv
            forms = ((i["o_wm"], hip.MOD_PACKED | (hip.MOD_SPLIT if i["split"] else 0)),
                     (i["o_wm_t"], hip.MOD_PACKED | hip.MOD_TRANSPOSE | (hip.MOD_SPLIT if i["split"] else 0))) if i["kind"] < 2 \
                else ((i["o_wm"], 0),)
            for o_out, fl in forms:
                d = _lib.ModulateDesc()
                d.W, d.s, d.out, d.s_stride = conv.weight.data_ptr(), P(i["o_s"]), P(o_out), i["Cin"]
                d.Cout, d.Cin, d.ksq = i["Cout"], i["Cin"], 1
                d.flags = fl | (hip.MOD_DEMODULATE if conv.demodulate else 0)
                d.scale, d.row_begin = conv.scale, rows
                rows += i["Cout"]
                mdescs.append(d)
            bd = ModBwdDesc()
            bd.d_wm, bd.W, bd.s, bd.dW = P(i["o_d_wm"]), conv.weight.data_ptr(), P(i["o_s"]), P(i["o_dW"])
            bd.ds = P(o_ds_all + (i["o_s"] - o_s_all))
            bd.s_stride = bd.ds_stride = i["Cin"]
            bd.Cout, bd.Cin, bd.scale, bd.demodulate, bd.row_begin = i["Cout"], i["Cin"], conv.scale, int(bool(conv.demodulate)), blocks
            blocks += (i["Cout"] + 31) // 32
            bdescs.append(bd)
        if len(mdescs) > 64 or len(bdescs) > 64:
            raise Unsupported("more than 64 table entries")
        self._mod_tab = _upload((_lib.ModulateDesc * len(mdescs))(*mdescs), device)
        jobs, jblocks = [], 0
        for i in info:
            if i["S"] > 1:
                j = SlotJob()
                if i["kind"] < 2:
                    j.src, j.dst, j.n = P(i["o_d_bias_part"]), P(i["o_d_bias"]), i["Cout"]
                else:
                    j.src, j.dst, j.n = P(i["o_d_wm"]), P(i["o_d_wm"]), B * 3 * i["Cin"]
                j.slots, j.stride, j.row_begin = i["S"], i["cpad"], jblocks
                jblocks += (j.n + 255) // 256
                jobs.append(j)
        self._slot_tab = _upload((SlotJob * len(jobs))(*jobs), device) if jobs else None
        self._modbwd_tab = _upload((ModBwdDesc * len(bdescs))(*bdescs), device)

        # ---- the plan struct
        p = DecoderGradPlan()
        p.B, p.n_layers, p.style_dim = B, len(info), sd
        p.style_table, p.style_n, p.style_rows = tab._dev.data_ptr(), len(tab._descs), tab._rows
        p.mod_table, p.mod_n, p.mod_rows = self._mod_tab.data_ptr(), len(mdescs), rows
        p.modbwd_table, p.modbwd_n, p.modbwd_blocks = self._modbwd_tab.data_ptr(), len(bdescs), blocks
        p.style_w_offsets = self._woffs.data_ptr()
        p.styles, p.s_all, p.ds_all = P(o_styles), P(o_s_all), P(o_ds_all)       # (the table backward mirrors out -> dy by offset)
        p.d_styles, p.d_style_W, p.d_style_b = P(o_d_styles), P(info[0]["o_dmW"]), P(o_d_style_b)
        p.amax_base, p.amax_bytes = P(amax0), 4 * (amax1 - amax0)
        p.zero_base, p.zero_bytes = P(zero0), 4 * (zero1 - zero0)
        p.feat_amax, p.y_lo, p.g_lo = P(self.o_feat_amax), P(o_y_lo), P(o_g_lo)
        p.g[0], p.g[1] = P(o_g[0]), P(o_g[1])
        for k, o in enumerate(o_drgb_lo):
            p.drgb_lo[k] = P(o)
        p.rgb[0], p.rgb[1] = P(o_rgb[0]), P(o_rgb[1])
        p.nw_parts, p.nw_stride, p.d_noise_w = P(o_nw_parts), self.nw_stride, P(o_d_noise_w)
        if self._slot_tab is not None:
            p.slot_table, p.slot_n, p.slot_blocks = self._slot_tab.data_ptr(), len(jobs), jblocks
        for k, i in enumerate(info):
            L, m = p.layers[k], i["m"]
            L.kind, L.Cin, L.Cout, L.H, L.W = i["kind"], i["Cin"], i["Cout"], i["H"], i["W"]
            L.wm, L.d_wm, L.d_bias = P(i["o_wm"]), P(i["o_d_wm"]), P(i["o_d_bias"])
            L.slots = i["S"]
            if i["kind"] < 2:
                L.slot_stride = i["cpad"]
                if i["S"] > 1:
                    L.d_bias = P(i["o_d_bias_part"])            # the copies; cips3d_slot_reduce writes their sum to o_d_bias
            else:
                L.rgb_slot_stride = i["cpad"]
            if i["kind"] < 2:
                L.noise_index, L.flags = i["conv_i"], 1 if i["split"] else 0
                L.bias, L.noise_w = m.activate.bias.data_ptr(), m.noise.weight.data_ptr()
                L.fir = m.conv.blur.kernel.data_ptr() if i["kind"] == 1 else None
                L.wm_t, L.y = P(i["o_wm_t"]), P(i["o_y"])
                L.y_amax, L.g_amax, L.glo_amax = P(i["o_y_amax"]), P(i["o_g_amax"]), P(i["o_glo_amax"])
                L.d_nw_part = P(o_nw_parts + i["conv_i"] * self.nw_stride)
            else:
                L.noise_index = -1
                L.bias = m.bias.data_ptr()
                L.fir = m.upsample.kernel.data_ptr() if i["kind"] == 3 else None
        self.plan = p
        self.o_d_styles, self.o_d_style_b, self.o_d_noise_w = o_d_styles, o_d_style_b, o_d_noise_w

    @staticmethod
    def weights_key(dec):
        return tuple(p.data_ptr() for p in parameters_of(dec)) + \
            tuple((bool(getattr(m, "split", True)), bool(getattr(m, "bf16", False))) for m, _ in dec._mod_layers())

    def _io(self, features, noise):
        io = DecoderGradIO()
        io.features = dev_ptr(features, "features")
        convs = [i for i in self.info if i["kind"] < 2]
        if len(noise) != len(convs):
            raise RuntimeError(f"{len(convs)} noise maps expected, got {len(noise)}")
        for k, (i, nz) in enumerate(zip(convs, noise)):
            if tuple(nz.shape[-2:]) != (i["Ho"], i["Wo"]) or nz.shape[0] not in (1, self.B):
                raise RuntimeError(f"noise {k}: shape {tuple(nz.shape)} for a {i['Ho']}x{i['Wo']} layer")
            io.noise[k] = dev_ptr(nz, "noise")
            io.noise_bstride[k] = i["Ho"] * i["Wo"] if (nz.shape[0] == self.B and self.B > 1) else 0
        if getattr(self, "_direct", None) is not None:
            io.style_table, io.styles = self._direct[0]._dev.data_ptr(), self._direct[1].data_ptr()
        return io

    def forward(self, features, styles, noise):
        lib = _lib.load()
        # the style heads read the caller's W+ tensor where it is when its address is a steady one (a parameter the optimiser
        # updates in place: hip.LinearTable.repointed), else from the plan's staging copy
        self._direct = None
        if styles.is_cuda and styles.dtype == torch.float32 and styles.is_contiguous() and tuple(styles.shape) == tuple(self.styles_buf.shape):
            tab = self._style_tab.repointed(self.styles_buf.data_ptr(), styles)
            if tab is not None:
                self._direct = (tab, styles)
        if self._direct is None:
            self.styles_buf.copy_(styles)
        H, W = self.out_hw
        rgb = torch.empty(self.B, 3, H, W, device=self.dev)
        io = self._io(features, noise)
        io.rgb = rgb.data_ptr()
        ev = hip.want_events("decoder_grad_forward")          # (bench.py: the node's launches between two events)
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if ev is not None else None
        if pair:
            pair[0].record()
        _lib.check(lib.cips3d_decoder_grad_forward(C.byref(self.plan), C.byref(io), _lib.stream_ptr()), "cips3d_decoder_grad_forward")
        if pair:
            pair[1].record()
            ev.append(pair)
        return rgb

    def backward(self, features, noise, d_rgb, need_features=True):
        """-> (d_features or None, d_styles [B, n_latent, style_dim], parameter gradients in parameters_of(dec) order).
        Everything returned is a view of ONE fresh tensor (the plan's output block, cloned): nothing aliases the plan."""
        lib = _lib.load()
        io = self._io(features, noise)
        io.d_rgb = dev_ptr(d_rgb, "d_rgb")
        d_features = torch.empty_like(features) if need_features else None
        io.d_features = dev_ptr(d_features, "d_features", True)
        ev = hip.want_events("decoder_grad_backward")
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if ev is not None else None
        if pair:
            pair[0].record()
        _lib.check(lib.cips3d_decoder_grad_backward(C.byref(self.plan), C.byref(io), _lib.stream_ptr()), "cips3d_decoder_grad_backward")
        if pair:
            pair[1].record()
            ev.append(pair)
        o0, o1 = self.out_range
        out = self.ws[o0:o1].clone()
        v = lambda o, *shape: out[o - o0:o - o0 + int(torch.Size(shape).numel())].view(*shape)     # noqa: E731
        sd = self.style_dim
        grads = []
        for i in self.info:
            m = i["m"]
            grads += [v(i["o_dW"], 1, i["Cout"], i["Cin"], 1, 1), v(i["o_dmW"], i["Cin"], sd)]
            row = i["o_s"] - self.info[0]["o_s"]          # rows of the style table before this head = sum of B * Cin ... / B
            grads.append(v(self.o_d_style_b + row // self.B, i["Cin"]))
            if i["kind"] < 2:
                grads += [v(self.o_d_noise_w + i["conv_i"], 1), v(i["o_d_bias"], i["Cout"])]
            else:
                grads.append(v(i["o_d_bias"], 1, 3, 1, 1))
        return d_features, v(self.o_d_styles, self.B, self.n_latent, sd), grads


# plans live beside the modules, not inside them: copy.deepcopy(G) (the projector's first step) must neither copy raw device
# pointers of the original's parameters nor trip over ctypes structures
_PLANS = weakref.WeakKeyDictionary()


def plan_for(dec, B, H0, W0, device):
    plans = _PLANS.setdefault(dec, {})
    key = (B, H0, W0, str(device))
    ent = plans.get(key)
    if ent is None or (ent is not False and ent.key != GradPlan.weights_key(dec)):
        try:
            ent = GradPlan(dec, B, H0, W0, device)
        except Unsupported:
            ent = False
        plans[key] = ent
    return ent or None


class DecoderFn(Function):
    """Decoder.forward as one node: apply(dec, plan, features, styles, n_noise, *noise, *parameters_of(dec))."""

    @staticmethod
    def forward(ctx, dec, plan, features, styles, n_noise, *rest):
        noise = [t.contiguous().float() for t in rest[:n_noise]]
        features = features.contiguous().float()
        rgb = plan.forward(features, styles.contiguous().float(), noise)
        ctx.plan, ctx.n_noise, ctx.n_params = plan, n_noise, len(rest) - n_noise
        plan.generation = ctx.generation = getattr(plan, "generation", 0) + 1      # the kept activations belong to THIS forward
        ctx.save_for_backward(features, *noise)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        features, *noise = ctx.saved_tensors
        if ctx.generation != ctx.plan.generation:
            raise RuntimeError(
                "DecoderFn.backward: the decoder ran forward again (same batch and size) before this graph's backward -- the "
                "one-call plan keeps ONE forward's activations.  Run backward before the next forward, or set "
                "CIPS3D_ONE_CALL_DECODER=0 (one autograd node per op, every graph keeps its own tensors)")
        d_features, d_styles, grads = ctx.plan.backward(features, noise, d_rgb.contiguous().float(),
                                                        need_features=ctx.needs_input_grad[2])
        need = ctx.needs_input_grad[5 + ctx.n_noise:]
        return (None, None, d_features, d_styles if ctx.needs_input_grad[3] else None, None) + (None,) * ctx.n_noise + \
            tuple(g if n else None for g, n in zip(grads, need))


def decoder_forward(dec, features, styles, noise):
    """The one-call route when it covers this call, else None (the caller falls back to the per-op walk)."""
    if noise is None or any(n is None for n in noise) or any(n.requires_grad for n in noise):
        return None
    B, _, H0, W0 = features.shape
    plan = plan_for(dec, B, H0, W0, features.device)
    if plan is None:
        return None
    return DecoderFn.apply(dec, plan, features, styles, len(noise), *noise, *parameters_of(dec))
