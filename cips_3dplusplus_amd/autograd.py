"""Differentiable form of the generator path: one torch.autograd.Function per HIP forward op, each with an explicit
HIP backward (csrc/backward.hip, csrc/nerf_bwd.hip).  torch's autograd engine is only the tape that orders the calls
and sums fan-out gradients; the reference relies on the same engine (`loss.backward()` in
/root/reference/exp/cips3d/models/projector_v10.py:1203-1209).

Used by `Generator.forward` when gradients are enabled and something that reaches the output requires them
(flip inversion: camera angles, W+ styles, decoder parameters, noise buffers — projector_v10.py:985-1009).
"""
import os

import torch
from torch.autograd import Function

from . import hip, op


def _c(t):
    return t.contiguous() if t is not None else None


# the decoder's 26 style modulations go through one table launch (and a two-launch table backward); 0 = one op per layer
STYLE_TABLE = os.environ.get("CIPS3D_STYLE_TABLE", "1") != "0"
# Decoder.forward under autograd as ONE node with a one-call backward (decoder_grad.py); 0 = one node per op (A/B knob, and the
# route for what the one-call plan does not cover)
ONE_CALL_DECODER = os.environ.get("CIPS3D_ONE_CALL_DECODER", "1") != "0"


class LinearFn(Function):
    """hip.linear without PixelNorm / truncation: EqualLinear, MappingLinear, LinearLayer."""

    @staticmethod
    def forward(ctx, x, W, bias, w_scale, b_scale, lrelu, act_gain, out_scale, out_shift):
        x = _c(x.float())
        y = hip.linear(x, W, bias, w_scale=w_scale, b_scale=b_scale, lrelu=lrelu, act_gain=act_gain, out_scale=out_scale,
                       out_shift=out_shift)
        ctx.save_for_backward(x, W, y if lrelu else None)
        ctx.cfg = (w_scale, b_scale, lrelu, act_gain, out_scale, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        w_scale, b_scale, lrelu, act_gain, out_scale, has_bias = ctx.cfg
        nx, nW, nb = ctx.needs_input_grad[:3]
        dx, dW, db = hip.linear_bwd(x, W, _c(dy), out=y, w_scale=w_scale, b_scale=b_scale, lrelu=lrelu, act_gain=act_gain,
                                    out_scale=out_scale, need_dx=nx, need_dW=nW, need_db=nb and has_bias)
        return dx, dW, db, None, None, None, None, None, None


def linear(x, W, bias=None, w_scale=1.0, b_scale=1.0, lrelu=False, act_gain=1.0, out_scale=1.0, out_shift=0.0):
    return LinearFn.apply(x, W, bias, float(w_scale), float(b_scale), bool(lrelu), float(act_gain), float(out_scale),
                          float(out_shift))


class ModulateFn(Function):
    """ModulatedConv2d weight modulation: W [1,Cout,Cin,k,k], s [B,Cin] -> wm [B,Cout,Cin] (k = 1) or [B,Cout,Cin,k,k]."""

    @staticmethod
    def forward(ctx, W, s, scale, demodulate, pre=None):
        """pre: this layer's wm when every layer was modulated by one table launch (modulate_all); the node then only
        records the dependency."""
        _, Cout, Cin, kh, kw = W.shape
        s = _c(s)
        if pre is None:
            pre = hip.modulate_weights(W, s, s.shape[1], s.shape[0], Cout, Cin, kh * kw, scale, demodulate, packed=False)
        ctx.save_for_backward(W, s)
        ctx.cfg = (Cout, Cin, kh * kw, scale, demodulate)
        return pre.view(s.shape[0], Cout, Cin) if kh * kw == 1 else pre.view(s.shape[0], Cout, Cin, kh, kw)

    @staticmethod
    def backward(ctx, dwm):
        W, s = ctx.saved_tensors
        Cout, Cin, ksq, scale, demodulate = ctx.cfg
        dW, ds = hip.modulate_bwd(dwm.contiguous().clone(), W, s, Cout, Cin, ksq, scale, demodulate,
                                  need_dW=ctx.needs_input_grad[0])
        return dW, ds, None, None, None


def _split_ok(k, m=None):
    """Split-fp16 GEMM mode for a forward GEMM (activation operand) whose contraction length is k (hip.SPLIT_BACKWARD: knob).
    m: the StyledConv the GEMM belongs to -- the differentiable route follows the module's precision like the inference route
    (Decoder.set_precision: "fp32" = split products, "fp32_exact" / "bf16*" = the fp32 matrix instruction here)."""
    if m is not None and (not getattr(m, "split", True) or getattr(m, "bf16", False)):
        return False
    return hip.SPLIT_BACKWARD and k % 32 == 0


class Conv1x1Fn(Function):
    """Per-sample GEMM y[b] = wm[b] x[b] (x [B,Cin,H,W]); backward = the same GEMM on wm^T + the pixel-contraction GEMM."""

    @staticmethod
    def forward(ctx, x, wm, packed=None, packed_t=None, split=None):
        x = _c(x)
        wm = _c(wm)
        Cout = wm.shape[1]
        sp = _split_ok(wm.shape[2]) if split is None else bool(split)   # `packed` (from modulate_all) is split-packed alike
        y = hip.modconv1x1(x, packed if packed is not None else hip.pack_weights(wm, split=sp), Cout, epilogue=0, split=sp)
        ctx.save_for_backward(x, wm)
        ctx.packed_t = packed_t              # fragments of wm^T from the same table launch (else packed in the backward)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wm = ctx.saved_tensors
        dy = _c(dy)
        dx = dwm = None
        if ctx.needs_input_grad[0]:          # gradient operand: fp32 MFMA (hip.SPLIT_BACKWARD explains)
            pt = ctx.packed_t if ctx.packed_t is not None else hip.pack_weights(wm, transpose=True)
            dx = hip.modconv1x1(dy, pt, wm.shape[2], epilogue=0)
        if ctx.needs_input_grad[1]:
            dwm = hip.gemm_wgrad(dy, x)
        return dx, dwm, None, None, None


class ConvKxKFn(Function):
    """k x k modulated convolution (models/model_v3.py:280-312): cross-correlation with padding k // 2, or -- `up` -- the
    stride-2 transposed convolution of the up-sampling branch (the blur that follows is op.upfirdn2d).  wm [B,Cout,Cin,k,k].
    Forward = cips3d_modconv_kxk.  Backward = ONE data-gradient GEMM and k^2 weight-gradient GEMMs on the 1x1 kernels: the k^2
    shifted (strided, for `up`) views of the output gradient are stacked along the channels, so that
        dx = [W_tap^T]_taps . [dy_tap]_taps                (cips3d_modconv1x1, contraction over k^2 Cout)
        dwm[..., tap] = dy_tap x^T  (up) / dy x_tap^T       (cips3d_gemm_wgrad, contraction over the pixels)
    -- the shifts are copies (pad / slice), every multiply-add runs in a HIP kernel."""

    @staticmethod
    def forward(ctx, x, wm, up):
        x, wm = _c(x), _c(wm)
        B, Cout, Cin, k, _ = wm.shape
        if Cout % 32 or Cin % 32 or (x.shape[2] * x.shape[3]) % 4:
            raise NotImplementedError("k x k backward needs channel counts that are multiples of 32 and H * W % 4 == 0")
        ctx.save_for_backward(x, wm)
        ctx.up = bool(up)
        return hip.modconv_kxk(x, wm, Cout, k, transpose2=bool(up))

    @staticmethod
    def backward(ctx, dy):
        x, wm = ctx.saved_tensors
        dy = _c(dy)
        B, Cout, Cin, k, _ = wm.shape
        H, W = x.shape[2:]
        taps = [(ky, kx) for ky in range(k) for kx in range(k)]
        pad = k // 2
        if ctx.up:        # y[o][2y + ky][2x + kx] += x[i][y][x] w[o][i][ky][kx]
            dys = [dy[:, :, ky:ky + 2 * H:2, kx:kx + 2 * W:2].contiguous() for ky, kx in taps]
        else:             # y[o][Y][X] = sum x[i][Y + ky - pad][X + kx - pad] w[o][i][ky][kx]
            dyp = torch.nn.functional.pad(dy, (pad, pad, pad, pad))
            dys = [dyp[:, :, 2 * pad - ky:2 * pad - ky + H, 2 * pad - kx:2 * pad - kx + W].contiguous() for ky, kx in taps]
        dx = dwm = None
        if ctx.needs_input_grad[0]:
            stacked = torch.cat(dys, 1)                                                   # [B, k^2 Cout, H, W]
            w_all = wm.permute(0, 2, 3, 4, 1).reshape(B, Cin, k * k * Cout).contiguous()    # [B, Cin, (tap, o)]
            dx = hip.modconv1x1(stacked, hip.pack_weights(w_all), Cin, epilogue=0)
        if ctx.needs_input_grad[1]:
            if ctx.up:
                parts = [hip.gemm_wgrad(d, x) for d in dys]
            else:
                xp = torch.nn.functional.pad(x, (pad, pad, pad, pad))
                parts = [hip.gemm_wgrad(dy, xp[:, :, ky:ky + H, kx:kx + W].contiguous()) for ky, kx in taps]
            dwm = torch.stack(parts, -1).view(B, Cout, Cin, k, k)
        return dx, dwm, None


class Conv1x1ActFn(Function):
    """StyledConv without up-sampling in one forward launch: GEMM with the noise + bias + leaky-ReLU epilogue
    (cips3d_modconv1x1, epilogue 1); backward = NoiseBiasActFn's followed by Conv1x1Fn's."""

    @staticmethod
    def forward(ctx, x, wm, packed, noise, noise_w, bias, packed_t=None, split=None):
        x, wm, noise = _c(x), _c(wm), _c(noise)
        ctx.packed_t = packed_t
        sp = _split_ok(wm.shape[2]) if split is None else bool(split)
        y = hip.modconv1x1(x, packed if packed is not None else hip.pack_weights(wm, split=sp), wm.shape[1], epilogue=1,
                           noise=noise, noise_w=noise_w, bias=bias, split=sp)
        ctx.save_for_backward(x, wm, y, noise, noise_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wm, y, noise, noise_w = ctx.saved_tensors
        n_x, n_wm, _, n_noise, n_nw, n_b = ctx.needs_input_grad[:6]
        dpre, dnoise, dnw, db = hip.noise_bias_act_bwd(_c(dy), y, noise, noise_w, need_dnoise=n_noise, need_dnw=n_nw, need_db=n_b)
        pt = ctx.packed_t if ctx.packed_t is not None else (hip.pack_weights(wm, transpose=True) if n_x else None)
        dx = hip.modconv1x1(dpre, pt, wm.shape[2], epilogue=0) if n_x else None
        dwm = hip.gemm_wgrad(dpre, x) if n_wm else None
        return dx, dwm, None, dnoise, dnw, db, None, None


class NoiseBiasActFn(Function):
    """NoiseInjection + FusedLeakyReLU: y = lrelu(x + nw*noise + bias_c)*sqrt2."""

    @staticmethod
    def forward(ctx, x, noise, noise_w, bias):
        x = _c(x)
        noise = _c(noise)
        y = hip.noise_bias_act(x, noise, noise_w, bias)
        ctx.save_for_backward(y, noise, noise_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, noise, noise_w = ctx.saved_tensors
        _, n_noise, n_nw, n_b = ctx.needs_input_grad
        dx, dnoise, dnw, db = hip.noise_bias_act_bwd(_c(dy), y, noise, noise_w, need_dnoise=n_noise, need_dnw=n_nw, need_db=n_b)
        return dx, dnoise, dnw, db


class ToRGBFn(Function):
    """rgb = wm x + bias (+ skip, already at the output resolution)."""

    @staticmethod
    def forward(ctx, x, wm, bias, skip):
        x = _c(x)
        wm = _c(wm)
        rgb = hip.torgb(x, wm, bias, skip=_c(skip), skip_up=False)
        ctx.save_for_backward(x, wm)
        ctx.has_skip = skip is not None
        return rgb

    @staticmethod
    def backward(ctx, drgb):
        x, wm = ctx.saved_tensors
        drgb = _c(drgb)
        dx, dwm, db = hip.torgb_bwd(drgb, x, wm, need_db=ctx.needs_input_grad[2])
        return dx, dwm, db.view(1, 3, 1, 1) if db is not None else None, (drgb if ctx.has_skip else None)


class StyleTableFn(Function):
    """Every ModulatedConv2d.modulation of the decoder (26 EqualLinears style -> Cin, models/model_v3.py:254,268) in one
    launch forward (cips3d_linear_table) and two backward (cips3d_linear_table_bwd) instead of 26 + 78 launches of ~5-14 us.
    Outputs: one contiguous [B, Cin_l] tensor per layer (blocks of one buffer)."""

    @staticmethod
    def _table(dec, B, device):
        key = (B, dec.conv1.conv.modulation.weight.data_ptr())
        ent = getattr(dec, "_grad_style_table", None)
        if ent is None or ent[0] != key:
            seq = dec._mod_layers()
            sizes = [m.conv.in_channel for m, _ in seq]
            styles_buf = torch.empty(B, dec.n_latent, dec.style_dim, device=device)
            s_buf = torch.empty(B * sum(sizes), device=device)
            tab = hip.LinearTable(device)
            offs, off = [], 0
            for (m, li), cin in zip(seq, sizes):
                mod = m.conv.modulation
                tab.add(mod.weight, mod.bias, styles_buf, dec.n_latent * dec.style_dim, s_buf, cin, w_scale=mod.scale,
                        b_scale=mod.lr_mul, x_offset=li * dec.style_dim, out_offset=off)
                offs.append(off)
                off += B * cin
            ent = (key, styles_buf, s_buf, tab, offs, sizes)
            dec._grad_style_table = ent
        return ent[1:]

    @staticmethod
    def forward(ctx, dec, styles, *params):
        B = styles.shape[0]
        styles_buf, s_buf, tab, offs, sizes = StyleTableFn._table(dec, B, styles.device)
        styles_buf.copy_(styles)
        tab.run(B)
        out = s_buf.clone()
        ctx.dec, ctx.B = dec, B
        ctx.save_for_backward(styles.detach().float().contiguous())
        return tuple(out[o:o + B * c].view(B, c) for o, c in zip(offs, sizes))

    @staticmethod
    def backward(ctx, *grads):
        dec, B = ctx.dec, ctx.B
        (styles,) = ctx.saved_tensors
        styles_buf, s_buf, tab, offs, sizes = StyleTableFn._table(dec, B, styles.device)
        styles_buf.copy_(styles)                    # the staging buffer may have been reused since the forward
        dy = torch.cat([(g if g is not None else torch.zeros(B, c, device=styles.device)).reshape(-1).float()
                        for g, c in zip(grads, sizes)])
        need_p = any(ctx.needs_input_grad[2:])
        dstyles = torch.zeros_like(styles_buf) if ctx.needs_input_grad[1] else None
        dW, woffs, db = tab.backward(B, s_buf, dy, styles_buf, dstyles, need_dW=need_p, need_db=need_p)
        outs = [None, dstyles]
        row = 0
        for i, (d, wo) in enumerate(zip(tab._descs, woffs)):
            nW, nb = ctx.needs_input_grad[2 + 2 * i], ctx.needs_input_grad[3 + 2 * i]
            outs.append(dW[wo:wo + d.out_dim * d.in_dim].view(d.out_dim, d.in_dim) if (nW and dW is not None) else None)
            outs.append(db[row:row + d.out_dim] if (nb and db is not None) else None)
            row += d.out_dim
        return tuple(outs)


@torch.no_grad()
def modulate_all(dec, s_list):
    """Modulated weights of every decoder layer from its s (values only; the autograd dependency is recorded per layer by
    ModulateFn(pre=...)): ONE cips3d_modulate_table launch writes the plain [B,Cout,Cin] form of all 26 layers and the
    MFMA-packed form of the 17 StyledConvs, instead of 26 modulate + 17 pack launches.
    Returns {id(module): (wm_plain, wm_packed or None)}."""
    from . import _lib
    from .decoder import StyledConv
    seq = dec._mod_layers()
    B, dev = s_list[0].shape[0], s_list[0].device
    key = (B, dec.conv1.conv.modulation.weight.data_ptr(), bool(getattr(dec, "split", True)), bool(getattr(dec, "bf16", False)),
           hip.SPLIT_BACKWARD)
    ent = getattr(dec, "_grad_mod_table", None)
    if ent is None or ent[0] != key:
        sizes = [m.conv.in_channel for m, _ in seq]
        s_stage = torch.empty(B * sum(sizes), device=dev)
        plain_n = [B * m.conv.out_channel * m.conv.in_channel for m, _ in seq]
        packs = [isinstance(m, StyledConv) and hip.modconv1x1_supported(m.conv.in_channel, m.conv.out_channel, 4096) for m, _ in seq]
        plain = torch.empty(sum(plain_n), device=dev)
        packed = torch.empty(sum(n for n, p in zip(plain_n, packs) if p), device=dev)
        packed_t = torch.empty_like(packed)      # fragments of wm^T: the data-gradient GEMMs' A operands, same launch
        descs, rows, so, po, ko = [], 0, 0, 0, 0
        layout = []
        for (m, _), cin, n, pk in zip(seq, sizes, plain_n, packs):
            conv = m.conv
            for want_packed in ((0, 1, 2) if pk else (0,)):            # 0 plain, 1 packed (forward), 2 packed transpose
                d = _lib.ModulateDesc()
                d.W = conv.weight.data_ptr()
                d.s = s_stage.data_ptr() + 4 * so
                d.out = (plain.data_ptr() + 4 * po, packed.data_ptr() + 4 * ko, packed_t.data_ptr() + 4 * ko)[want_packed]
                d.s_stride = cin
                d.Cout, d.Cin, d.ksq = conv.out_channel, conv.in_channel, 1
                d.flags = ((hip.MOD_DEMODULATE if conv.demodulate else 0) | (hip.MOD_PACKED if want_packed else 0) |
                           (hip.MOD_SPLIT if (want_packed == 1 and _split_ok(conv.in_channel, m)) else 0) |
                           (hip.MOD_TRANSPOSE if want_packed == 2 else 0))
                d.scale = conv.scale
                d.row_begin = rows
                rows += conv.out_channel
                descs.append(d)
            layout.append((po, n, ko if pk else None))
            so += B * cin
            po += n
            ko += n if pk else 0
        arr = (_lib.ModulateDesc * len(descs))(*descs)
        tab_dev = torch.frombuffer(bytearray(bytes(memoryview(arr))), dtype=torch.uint8).to(dev)
        ent = (key, s_stage, plain, packed, tab_dev, len(descs), rows, layout, packed_t)
        dec._grad_mod_table = ent
    _, s_stage, plain, packed, tab_dev, n_desc, rows, layout, packed_t = ent
    torch.cat([t.reshape(-1) for t in s_list], out=s_stage)
    _lib.check(_lib.load().cips3d_modulate_table(tab_dev.data_ptr(), n_desc, rows, B, 0.0, _lib.stream_ptr()), "cips3d_modulate_table")
    plain_now = plain.clone()            # saved by Conv1x1Fn / ToRGBFn for their backward: must outlive the next forward
    packed_t_now = packed_t.clone()      # read by the backward of this forward
    out = {}
    for (m, _), (po, n, ko) in zip(seq, layout):
        out[id(m)] = (plain_now[po:po + n], packed[ko:ko + n] if ko is not None else None,
                      packed_t_now[ko:ko + n] if ko is not None else None)
    return out


def decoder_styles(dec, styles):
    """s_l of every decoder layer, in `dec._mod_layers()` order."""
    params = []
    for m, _ in dec._mod_layers():
        params += [m.conv.modulation.weight, m.conv.modulation.bias]
    return StyleTableFn.apply(dec, styles, *params)


# ------------------------------------------------------------------------------------------ decoder, differentiable walk
def styled_conv(sc, x, style, noise, s=None, pre=None):
    """StyledConv (models/model_v3.py:444-454) as a chain of differentiable HIP ops.  `s`: the layer's modulation when it
    was computed by the table (decoder_styles); `pre`: (wm, packed wm) from modulate_all."""
    conv = sc.conv
    mod = conv.modulation
    if s is None:
        s = linear(style, mod.weight, mod.bias, w_scale=mod.scale, b_scale=mod.lr_mul)
    if conv.kernel_size != 1:              # k x k: modulate -> ConvKxKFn [-> blur] -> noise + bias + activation
        wm = ModulateFn.apply(conv.weight, s, conv.scale, conv.demodulate, None)
        y = ConvKxKFn.apply(x, wm, bool(conv.upsample))
        if conv.upsample:
            y = op.upfirdn2d(y, conv.blur.kernel, pad=conv.blur.pad)
        if noise is None:
            noise = torch.randn(y.shape[0], 1, y.shape[2], y.shape[3], device=y.device)
        return NoiseBiasActFn.apply(y, noise, sc.noise.weight, sc.activate.bias)
    wm = ModulateFn.apply(conv.weight, s, conv.scale, conv.demodulate, pre[0] if pre else None)
    packed = pre[1] if (pre and x.shape[2] * x.shape[3] % 4 == 0) else None
    packed_t = pre[2] if (pre and packed is not None and len(pre) > 2) else None
    if not conv.upsample:
        if noise is None:
            noise = torch.randn(x.shape[0], 1, x.shape[2], x.shape[3], device=x.device)
        return Conv1x1ActFn.apply(x, wm, packed, noise, sc.noise.weight, sc.activate.bias, packed_t, _split_ok(conv.in_channel, sc))
    y = Conv1x1Fn.apply(x, wm, packed, packed_t, _split_ok(conv.in_channel, sc))
    y = op.upfirdn2d(y, conv.blur.kernel, up=2, pad=(2, 1))
    if noise is None:
        noise = torch.randn(y.shape[0], 1, y.shape[2], y.shape[3], device=y.device)
    return NoiseBiasActFn.apply(y, noise, sc.noise.weight, sc.activate.bias)


def to_rgb(tr, x, style, skip, s=None, pre=None):
    conv = tr.conv
    mod = conv.modulation
    if s is None:
        s = linear(style, mod.weight, mod.bias, w_scale=mod.scale, b_scale=mod.lr_mul)
    wm = ModulateFn.apply(conv.weight, s, conv.scale, False, pre[0] if pre else None)
    if skip is not None and tr.upsample:
        skip = tr.upsample(skip)            # op.upfirdn2d, differentiable
    return ToRGBFn.apply(x, wm, tr.bias, skip)


def decoder_forward(dec, features, styles, noise=None):
    """Decoder.forward (models/model_v3.py:592-637) with gradients."""
    if ONE_CALL_DECODER and noise is not None and dec.kernel_size == 1:
        # the whole decoder as one autograd node (csrc/decoder_grad.hip); None when the plan does not cover this call
        from . import decoder_grad
        rgb = decoder_grad.decoder_forward(dec, features, styles, list(noise))
        if rgb is not None:
            return rgb
    if noise is None:
        noise = [None] * dec.num_layers
    # all 26 style modulations from one table launch (forward and backward); S maps module -> its s
    seq = dec._mod_layers()
    S, M = {}, {}
    if STYLE_TABLE and dec.style_dim % 4 == 0:
        s_list = decoder_styles(dec, styles)
        S = {id(m): sl for (m, _), sl in zip(seq, s_list)}
        if dec.kernel_size == 1 and all(m.conv.in_channel % 32 == 0 and m.conv.out_channel % 32 == 0
                                        for m, _ in seq if hasattr(m, "noise")):
            M = modulate_all(dec, s_list)
    out = styled_conv(dec.conv1, features, styles[:, 0], noise[0], S.get(id(dec.conv1)), M.get(id(dec.conv1)))
    skip = to_rgb(dec.to_rgb1, out, styles[:, 1], None, S.get(id(dec.to_rgb1)), M.get(id(dec.to_rgb1)))
    i = 1
    for st in range(len(dec.to_rgbs)):
        c0, c1, tr = dec.convs[2 * st], dec.convs[2 * st + 1], dec.to_rgbs[st]
        out = styled_conv(c0, out, styles[:, i], noise[2 * st + 1], S.get(id(c0)), M.get(id(c0)))
        out = styled_conv(c1, out, styles[:, i + 1], noise[2 * st + 2], S.get(id(c1)), M.get(id(c1)))
        skip = to_rgb(tr, out, styles[:, i + 2], skip, S.get(id(tr)), M.get(id(tr)))
        i += 2
    return skip


# ------------------------------------------------------------------------------------------ camera + NeRF
class SqDiffPairFn(Function):
    """c0 sum (a0 - b0)^2 + c1 sum (a1 - b1)^2 (the inversion loss' two squared-difference terms, projector_v10.py:1173-1174;
    c = weight / numel is F.mse_loss): two launches forward, one backward (the torch expression with its graph: ~23).
    Differentiable with respect to a0 and a1; b0 / b1 are targets."""

    @staticmethod
    def forward(ctx, a0, b0, c0, a1, b1, c1):
        a0, b0, a1, b1 = _c(a0), _c(b0), _c(a1), _c(b1)
        ctx.save_for_backward(a0, b0, a1, b1)
        ctx.c = (float(c0), float(c1))
        return hip.sqdiff_pair(a0, b0, c0, a1, b1, c1)

    @staticmethod
    def backward(ctx, g):
        a0, b0, a1, b1 = ctx.saved_tensors
        d0, d1 = hip.sqdiff_pair_bwd(a0, b0, ctx.c[0], a1, b1, ctx.c[1], g.contiguous().float())
        return d0, None, None, d1, None, None


def weighted_mse_pair(a0, b0, w0, a1, b1, w1):
    """w0 mse(a0, b0) + w1 mse(a1, b1) on the device in three launches per step (forward + backward)."""
    return SqDiffPairFn.apply(a0, b0, w0 / a0.numel(), a1, b1, w1 / a1.numel())


class CameraFn(Function):
    """Camera.generate_camera_params for given `locations` (azim, elev), differentiable w.r.t. them."""

    @staticmethod
    def forward(ctx, locations, img_size, fov_ang, dist_radius, up):
        extr, focal, near, far = hip.camera_params(locations, img_size, fov_ang, dist_radius, up=up)
        ctx.save_for_backward(locations.detach(), up)
        ctx.mark_non_differentiable(focal, near, far)
        ctx.set_materialize_grads(False)        # (else the engine fills a zero gradient for each of focal / near / far)
        return extr, focal, near, far

    @staticmethod
    def backward(ctx, dextr, dfocal, dnear, dfar):
        locations, up = ctx.saved_tensors
        if dextr is None:
            return None, None, None, None, None
        return hip.camera_params_bwd(locations, dextr, up).to(locations.dtype), None, None, None, None


class FilmTableFn(Function):
    """gamma / beta of every FiLM layer in one launch (the renderer's own table, cips3d_linear_table) with the table backward."""

    @staticmethod
    def _direct(styles, B):
        """Can the heads read `styles` where it is?  (contiguous fp32 on the device; [B, ...], or [1, ...] broadcast to B views)"""
        return (styles.is_cuda and styles.dtype == torch.float32 and styles.is_contiguous() and styles.shape[0] in (1, B))

    @staticmethod
    def forward(ctx, renderer, styles, B, *params):
        styles_buf, film, tab = renderer._film_table(B, styles.device)
        direct = None
        if FilmTableFn._direct(styles, B):
            # the heads read the caller's tensor: no staging copy, and a [1, ...] latent serves all B views with row stride 0
            direct = tab.repointed(styles_buf.data_ptr(), styles.detach(), 0 if styles.shape[0] == 1 and B > 1 else None)
        ctx.direct = direct is not None
        if direct is not None:
            tab = direct
        else:
            styles_buf.copy_(styles)                # (broadcasts a [1, ...] latent)
        tab.run(B)
        ctx.renderer, ctx.B = renderer, B
        ctx.save_for_backward(styles.detach())
        return film.clone()

    @staticmethod
    def backward(ctx, dfilm):
        renderer, B = ctx.renderer, ctx.B
        (styles,) = ctx.saved_tensors
        styles_buf, film, tab = renderer._film_table(B, styles.device)
        x_base = styles_buf
        direct = tab.repointed(styles_buf.data_ptr(), styles, 0 if styles.shape[0] == 1 and B > 1 else None) if ctx.direct else None
        if direct is not None:
            tab, x_base = direct, styles
        else:
            styles_buf.copy_(styles)                # the staging buffer may have been reused since the forward
        need_p = any(ctx.needs_input_grad[3:])
        # (direct + broadcast: the table backward's atomics add every view's gradient into the one row -- the sum `expand` would take)
        dstyles = torch.zeros_like(x_base) if ctx.needs_input_grad[1] else None
        dW, woffs, db = tab.backward(B, film, _c(dfilm.float()), x_base, dstyles, need_dW=need_p, need_db=need_p)
        if dstyles is not None and dstyles.shape[0] != styles.shape[0]:
            dstyles = dstyles.sum(0, keepdim=True)  # (staged broadcast: the direct form's atomics take this sum in the launch)
        outs = [None, dstyles, None]
        row = 0
        for i, (d, wo) in enumerate(zip(tab._descs, woffs)):
            nW, nb = ctx.needs_input_grad[3 + 2 * i], ctx.needs_input_grad[4 + 2 * i]
            outs.append(dW[wo:wo + d.out_dim * d.in_dim].view(d.out_dim, d.in_dim) if (nW and dW is not None) else None)
            outs.append(db[row:row + d.out_dim] if (nb and db is not None) else None)
            row += d.out_dim
        return tuple(outs)


def film_table(renderer, styles, batch=None):
    """gamma / beta of every FiLM layer from the W+ styles (B, D+1, style_dim) -> [B, D+1, 2, H]
    (cips3d/volume_renderer.py:66-67 through LinearLayer :15-35).  batch: B views from ONE latent -- styles (1, D+1, style_dim)
    is then broadcast inside the table launch (the inversion loop's `w_render.repeat(2, 1, 1)`, projector_v10.py:1131, without the
    repeat and without the sum its backward is)."""
    net = renderer.network
    B = styles.shape[0] if batch is None else int(batch)
    if styles.shape[0] not in (1, B):
        raise ValueError(f"{styles.shape[0]} latents for a batch of {B} views")
    if STYLE_TABLE and renderer.style_dim % 4 == 0:
        params = []
        for layer in list(net.pts_linears) + [net.views_linears]:
            for head in (layer.gamma, layer.beta):
                params += [head.weight, head.bias]
        return FilmTableFn.apply(renderer, styles, B, *params)
    if styles.shape[0] != B:
        styles = styles.expand(B, -1, -1)
    rows = []
    for l, layer in enumerate(list(net.pts_linears) + [net.views_linears]):
        st = styles[:, l]
        gb = [linear(st, head.weight, head.bias, out_scale=float(head.std_init), out_shift=float(head.bias_init))
              for head in (layer.gamma, layer.beta)]
        rows.append(torch.stack(gb, 1))
    return torch.stack(rows, 1)


def nerf_named_parameters(renderer):
    """(name, parameter) of the renderer's own weights NerfRenderFn differentiates, in the order it takes them: layer weights
    and biases, the two heads, sigmoid_beta -- everything of VolumeFeatureRenderer except the gamma / beta heads (film_table)."""
    net = renderer.network
    out = []
    for l, layer in enumerate(net.pts_linears):
        out += [(f"pts_linears.{l}.weight", layer.weight), (f"pts_linears.{l}.bias", layer.bias)]
    out += [("views_linears.weight", net.views_linears.weight), ("views_linears.bias", net.views_linears.bias),
            ("rgb_linear.weight", net.rgb_linear.weight), ("rgb_linear.bias", net.rgb_linear.bias),
            ("sigma_linear.weight", net.sigma_linear.weight), ("sigma_linear.bias", net.sigma_linear.bias),
            ("sigmoid_beta", renderer.sigmoid_beta)]
    return out


class NerfRenderFn(Function):
    """VolumeFeatureRenderer.render with gradients w.r.t. the camera pose and the FiLM table.  Forward = the fused kernel;
    backward = the fused recompute + backward kernels of csrc/nerf_bwd_fused.hip (the materialised sequence of
    csrc/nerf_bwd.hip for shapes they do not cover, or with CIPS3D_FUSED_NERF_BACKWARD=0).  The renderer's own weights are treated as constants
    (`optim_render_params: false` in the released inversion recipes, train_cips3d_compcars_v10.yaml:585)."""

    @staticmethod
    def forward(ctx, renderer, cam_poses, focals, near, far, film, perturb_u, img_size, n_samples, static_viewdirs, *params):
        """params: nerf_parameters(renderer) when the renderer's own weights are optimised (`optim_render_params`,
        models/projector_v10.py:848-872; the gamma / beta heads go through film_table) -- the backward then takes the
        materialised route, which keeps every layer's activations and can contract them with the gradients."""
        # With the fused backward available the forward keeps what it needs (accumulator stash, per-point sdf / rgb logits):
        # the backward then does not run the forward again (hip.STASH_IN_FORWARD = 0: it does, and nothing is held meanwhile).
        ctx.n_params = len(params)
        ctx.want_params = any(p.requires_grad for p in params)
        fwd = None
        if not ctx.want_params and hip.FUSED_NERF_BACKWARD and hip.STASH_IN_FORWARD and hip.nerf_backward_fused_supported(
                renderer.hidden_dim, renderer.N_layers_renderer, img_size, n_samples):
            fwd = hip.nerf_forward_stash(cam_poses.shape[0], img_size, n_samples, renderer.hidden_dim,
                                         renderer.N_layers_renderer, cam_poses.device)
        thumb, features, _, mask, xyz = renderer.render(cam_poses.detach(), focals, near, far, None, img_size, n_samples,
                                                        perturb_u=perturb_u, static_viewdirs=static_viewdirs, film=film,
                                                        stash=fwd, planar_mask=True)      # mask: [2,B,S,S]
        ctx.renderer = renderer
        ctx.fwd = fwd
        ctx.cfg = (img_size, n_samples, static_viewdirs)
        ctx.save_for_backward(cam_poses.detach(), focals, near, far, film.detach(), perturb_u)
        ctx.mark_non_differentiable(mask, xyz)
        ctx.set_materialize_grads(False)        # (else the engine fills zero gradients for xyz and mask; None is handled below)
        return features, thumb, xyz, mask

    @staticmethod
    def backward(ctx, dfeat, dthumb, dxyz, dmask):
        cam_poses, focals, near, far, film, perturb_u = ctx.saved_tensors
        img_size, n_samples, static = ctx.cfg
        r = ctx.renderer
        _, layer_bias = r._derived_buffers()
        B, H = cam_poses.shape[0], r.hidden_dim
        if dfeat is None:
            dfeat = torch.zeros(B, H, img_size, img_size, device=cam_poses.device)
        if dthumb is None:
            dthumb = torch.zeros(B, 3, img_size, img_size, device=cam_poses.device)
        if ctx.want_params:
            dfilm, dcam, pg = hip.nerf_backward(r.network, r.sigmoid_beta.detach() if r.with_sdf else None, cam_poses, focals, near, far, perturb_u, film,
                                                layer_bias, img_size, n_samples, static, dfeat.float(), dthumb.float(),
                                                need_params=True)
            names = [n for n, _ in nerf_named_parameters(r)]
            grads = tuple(pg[n].reshape(p.shape) if ctx.needs_input_grad[10 + i] else None
                          for i, (n, (_, p)) in enumerate(zip(names, nerf_named_parameters(r))))
            return (None, dcam, None, None, None, dfilm, None, None, None, None) + grads
        if hip.FUSED_NERF_BACKWARD and hip.nerf_backward_fused_supported(H, r.N_layers_renderer, img_size, n_samples):
            packed, _ = r._derived_buffers()
            dfilm, dcam = hip.nerf_backward_fused(r.network, r.sigmoid_beta.detach() if r.with_sdf else None, cam_poses, focals, near, far, perturb_u,
                                                  film, layer_bias, packed, r._packed_transposed(), img_size, n_samples,
                                                  static, dfeat, dthumb, fwd=ctx.fwd)
            ctx.fwd = None
        else:
            dfilm, dcam = hip.nerf_backward(r.network, r.sigmoid_beta.detach() if r.with_sdf else None, cam_poses, focals, near, far, perturb_u, film,
                                            layer_bias, img_size, n_samples, static, dfeat.float(), dthumb.float())
        return (None, dcam, None, None, None, dfilm, None, None, None, None) + (None,) * ctx.n_params
