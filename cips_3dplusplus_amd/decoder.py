"""StyleGAN2-style 2-D decoder modules (reference exp/cips3d/models/model_v3.py:32-757).

Same class names, constructor arguments, parameter / buffer names and shapes as the reference so
`decoder.*`, `style.*`, `style_decoder.*` keys of a reference checkpoint load unchanged
(SURVEY.md Appendix B).  Forward passes launch the HIP kernels of csrc/decoder.hip:

    ModulatedConv2d (k=1)   modulation GEMV -> modulate_weights (MFMA-packed) -> modconv1x1 GEMM
        up-sampling variant modconv1x1 at low resolution, then the 2x polyphase FIR
                            (== conv_transpose2d(stride 2) + Blur)
    StyledConv              the same with noise + bias + leaky-ReLU fused in the GEMM / FIR epilogue
    ToRGB                   3-channel modulated GEMV + bias + (FIR-upsampled) skip in one kernel
    ModulatedConv2d (k=3)   LDS-tiled implicit GEMM on MFMA (csrc/conv3x3.hip); the up-sampling variant applies the FIR to the
                            input tile in LDS; noise + bias + leaky-ReLU in its epilogue for StyledConv
    odd channel counts      direct k x k kernel + standalone FIR / epilogue kernels (generality path)
"""
import math

import torch
from torch import nn

from . import hip, op


class PixelNorm(nn.Module):
    """model_v3.py:32-37.  On the mapping path it is folded into the first EqualLinear launch."""

    @torch.no_grad()
    def forward(self, input):
        x = input.float().contiguous()
        out = torch.empty_like(x)
        lib = hip._lib.load()
        hip.check(lib.cips3d_pixel_norm(hip.dev_ptr(x, "input"), hip.dev_ptr(out), x.shape[0], x.shape[1], hip.stream_ptr()),
                  "cips3d_pixel_norm")
        return out


class MappingLinear(nn.Module):
    """model_v3.py:40-65: linear (no bias) -> lrelu(x + b) * 1."""

    def __init__(self, in_dim, out_dim, bias=True, activation=None, is_last=False):
        super().__init__()
        std = 0.25 if is_last else 1
        self.weight = nn.Parameter(std * nn.init.kaiming_normal_(torch.empty(out_dim, in_dim), a=0.2, mode="fan_in",
                                                                 nonlinearity="leaky_relu"))
        b = 1 / math.sqrt(in_dim)
        self.bias = nn.Parameter(torch.empty(out_dim).uniform_(-b, b)) if bias else None
        self.activation = activation

    def forward(self, input, out=None, trunc_mean=None, trunc_psi=1.0):
        return hip.linear(input, self.weight, self.bias, out=out, lrelu=self.activation is not None, act_gain=1.0,
                          trunc_mean=trunc_mean, trunc_psi=trunc_psi)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


class EqualLinear(nn.Module):
    """model_v3.py:183-210: weight stored / lr_mul, run-time scale lr_mul/sqrt(in), bias * lr_mul."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input, out=None, pixelnorm=False, trunc_mean=None, trunc_psi=1.0):
        return hip.linear(input, self.weight, self.bias, out=out, w_scale=self.scale, b_scale=self.lr_mul,
                          pixelnorm=pixelnorm, lrelu=bool(self.activation), act_gain=2 ** 0.5,
                          trunc_mean=trunc_mean, trunc_psi=trunc_psi)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


def make_kernel(k):
    """model_v3.py:73-81."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class Upsample(nn.Module):
    """model_v3.py:84-102: upfirdn2d(up=factor) with kernel * factor^2."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return op.upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Blur(nn.Module):
    """model_v3.py:126-142."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input):
        return op.upfirdn2d(input, self.kernel, pad=self.pad)


class ModulatedConv2d(nn.Module):
    """model_v3.py:218-314 (plain and up-sampling branches; the down-sampling branch is discriminator-only)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if downsample:
            raise NotImplementedError("down-sampling ModulatedConv2d is not on the generator path")
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample = upsample, downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")

    # -- pieces shared with StyledConv / ToRGB ------------------------------------------------------
    def fast(self, HW):
        return self.kernel_size == 1 and hip.modconv1x1_supported(self.in_channel, self.out_channel, HW)

    def tiled3x3(self, H, W):
        """The 3x3 MFMA kernel tiles this layer at input size H x W."""
        return self.kernel_size == 3 and hip.modconv3x3_supported(self.in_channel, self.out_channel, H, W, self.upsample)

    def modulated_weight(self, style, packed, flip=False, split=False):
        """style (B, style_dim) -> wm (packed MFMA order [split: fp16 hi + lo fragments] or plain [B,Cout,Cin,k,k])."""
        s = self.modulation(style.contiguous())
        return hip.modulate_weights(self.weight, s, s.shape[1], s.shape[0], self.out_channel, self.in_channel,
                                    self.kernel_size ** 2, self.scale, self.demodulate, packed, flip=flip, split=split)

    def forward(self, input, style):
        B, Cin, H, W = input.shape
        x = input.contiguous()
        if self.fast(H * W):
            wm = self.modulated_weight(style, packed=True)
            y = hip.modconv1x1(x, wm, self.out_channel, epilogue=0)
            if self.upsample:
                y = op.upfirdn2d(y, self.blur.kernel, up=2, pad=(2, 1))
            return y
        if self.tiled3x3(H, W):
            wm = self.modulated_weight(style, packed=True, flip=self.upsample)
            return hip.modconv3x3(x, wm, self.out_channel, up=self.upsample, fir=self.blur.kernel if self.upsample else None)
        wm = self.modulated_weight(style, packed=False)
        y = hip.modconv_kxk(x, wm, self.out_channel, self.kernel_size, transpose2=self.upsample)
        if self.upsample:
            y = self.blur(y)
        return y


class NoiseInjection(nn.Module):
    """model_v3.py:317-341 (project_noise is dead code for every released config)."""

    def __init__(self, project=False):
        super().__init__()
        if project:
            raise NotImplementedError("project_noise=True needs pytorch3d mesh rendering; unused by released configs")
        self.project = project
        self.weight = nn.Parameter(torch.zeros(1))


class StyledConv(nn.Module):
    """model_v3.py:418-454: ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU, fused."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 project_noise=False):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel)
        self.noise = NoiseInjection(project=project_noise)
        self.bias = nn.Parameter(torch.zeros(1, out_channel, 1, 1))     # present in checkpoints, unused in forward
        self.activate = op.FusedLeakyReLU(out_channel)
        self.bf16 = False        # bf16 compute mode of the GEMM (Decoder.set_precision)
        self.split = True        # fp32-equivalent split-fp16 products in the stand-alone GEMM (the default "fp32" precision)

    def forward(self, input, style, noise=None, transform=None, mesh_path=None, wm=None):
        B, Cin, H, W = input.shape
        conv = self.conv
        x = input.contiguous()
        Ho, Wo = (2 * H, 2 * W) if conv.upsample else (H, W)
        if noise is None:
            noise = torch.randn(B, 1, Ho, Wo, device=x.device)
        noise = noise.contiguous()
        nw = self.noise.weight          # device scalar, read by the kernels
        if conv.fast(H * W):
            split = self.split and not self.bf16
            if wm is None:
                wm = conv.modulated_weight(style, packed=True, split=split)
            if conv.upsample:
                y_lo = hip.modconv1x1(x, wm, conv.out_channel, epilogue=0, bf16=self.bf16, split=split)
                return hip.up2_fir_act(y_lo, conv.blur.kernel, noise, nw, self.activate.bias, track=split)
            return hip.modconv1x1(x, wm, conv.out_channel, epilogue=1, noise=noise, noise_w=nw, bias=self.activate.bias,
                                  bf16=self.bf16, split=split)
        if conv.tiled3x3(H, W):
            # (the default precision runs every decoder GEMM on split-fp16 products, this one included; "fp32_exact" and the
            # bf16 modes keep the fp32 matrix instruction here)
            split = self.split and not self.bf16
            if wm is None:
                wm = conv.modulated_weight(style, packed=True, flip=conv.upsample, split=split)
            return hip.modconv3x3(x, wm, conv.out_channel, up=conv.upsample, fir=conv.blur.kernel if conv.upsample else None,
                                  epilogue=1, noise=noise, noise_w=nw, bias=self.activate.bias, split=split)
        y = conv(x, style)
        return hip.noise_bias_act(y, noise, nw, self.activate.bias)


class ToRGB(nn.Module):
    """model_v3.py:457-482."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.upsample = upsample
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None, wm=None):
        x = input.contiguous()
        if wm is None:
            wm = self.conv.modulated_weight(style, packed=False)
        if skip is not None:
            skip = skip.contiguous()
        up = bool(self.upsample) and skip is not None
        return hip.torgb(x, wm, self.bias, skip=skip, skip_up=up, fir=self.upsample.kernel if up else None)


class Decoder(nn.Module):
    """model_v3.py:522-757."""

    def __init__(self, size_start, size_end, style_dim, in_channel, channel_multiplier, project_noise,
                 upsample_list=[], kernel_size=1, blur_kernel=[1, 3, 3, 1], **kwargs):
        super().__init__()
        self.module_name_list = []
        self.size_start, self.size_end, self.style_dim, self.in_channel = size_start, size_end, style_dim, in_channel
        self.channel_multiplier, self.project_noise = channel_multiplier, project_noise
        self.upsample_list, self.kernel_size, self.blur_kernel = upsample_list, kernel_size, blur_kernel
        m = channel_multiplier
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * m, 128: 128 * m, 256: 64 * m, 512: 32 * m,
                         1024: 16 * m}
        self.create_synthesis()
        self._tables = {}
        self.bf16 = False
        self.bf16_storage = False
        self.split = True         # "fp32" precision = split-fp16 stand-alone GEMMs (set_precision)

    def create_synthesis(self):
        self.log_in_size = int(math.log(self.size_start, 2))
        self.log_size = int(math.log(self.size_end, 2))
        cin, cout = self.in_channel, self.channels[self.size_start]
        self.conv1 = StyledConv(cin, cout, self.kernel_size, self.style_dim, blur_kernel=self.blur_kernel,
                                project_noise=self.project_noise)
        self.to_rgb1 = ToRGB(cout, self.style_dim, upsample=False)
        self.module_name_list.extend(["conv1", "to_rgb1"])
        self.convs = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        self.module_name_list.extend(["convs", "to_rgbs", "noises"])
        for i in range(self.log_in_size + 1, self.log_size + 1):
            cin, cout = cout, self.channels[2 ** i]
            up = (2 ** i) in self.upsample_list
            self.convs.append(StyledConv(cin, cout, self.kernel_size, self.style_dim, upsample=up,
                                         blur_kernel=self.blur_kernel, project_noise=self.project_noise))
            self.convs.append(StyledConv(cout, cout, self.kernel_size, self.style_dim, blur_kernel=self.blur_kernel,
                                         project_noise=self.project_noise))
            self.to_rgbs.append(ToRGB(cout, self.style_dim, upsample=up))
        self.num_layers = (self.log_size - self.log_in_size) * 2 + 1
        self.n_latent = (self.log_size - self.log_in_size) * 2 + 2

    def set_precision(self, precision):
        """"fp32" (default): fp32-equivalent results; the stand-alone GEMMs run split-fp16 products (x = hi + lo in fp16,
        three exact products accumulated in fp32: ~1e-6 relative to the fp32 MFMA, several times faster), the fused
        up-sampling stages the fp32 MFMA; "fp32_exact": the fp32 MFMA everywhere (bit-exact k-ordered fmaf chains);
        "bf16": every StyledConv GEMM rounds its operands to bf16 and accumulates in fp32
        (BASELINE config 3), storage / ToRGB / FIR / epilogues stay fp32; "bf16_storage": additionally the low-resolution
        GEMM result of every fused up-sampling stage (the only activation those stages move through HBM) is stored as bf16
        (one-call forward only; the per-op path keeps it in fp32)."""
        if precision not in ("fp32", "fp32_exact", "bf16", "bf16_storage"):
            raise ValueError(precision)
        self.bf16 = precision in ("bf16", "bf16_storage")
        self.bf16_storage = precision == "bf16_storage"
        self.split = precision == "fp32"
        for m in [self.conv1] + list(self.convs):
            m.bf16 = self.bf16
            m.split = self.split
        return self

    def create_noise_bufs(self, start_size, device):
        """model_v3.py:639-666."""
        bufs = [torch.randn(1, 1, start_size, start_size, device=device)]
        cur = start_size
        for i in range(self.log_in_size + 1, self.log_size + 1):
            if 2 ** i in self.upsample_list:
                cur *= 2
            bufs.append(torch.randn(1, 1, cur, cur, device=device))
            bufs.append(torch.randn(1, 1, cur, cur, device=device))
        return bufs

    # ---- all style modulations of the decoder in one launch ---------------------------------------
    def _mod_layers(self):
        """(module holding .conv, latent index) in evaluation order (model_v3.py:602-632)."""
        seq = [(self.conv1, 0), (self.to_rgb1, 1)]
        i = 1
        for s in range(len(self.to_rgbs)):
            seq += [(self.convs[2 * s], i), (self.convs[2 * s + 1], i + 1), (self.to_rgbs[s], i + 2)]
            i += 2
        return seq

    def _style_table(self, B, device, lane=0):
        key = (B, self.conv1.conv.modulation.weight.data_ptr())
        slot = B if lane == 0 else (B, lane)
        ent = self._tables.get(slot)
        if ent is None or ent[0] != key:
            seq = self._mod_layers()
            total = sum(m.conv.in_channel for m, _ in seq)
            styles_buf = torch.empty(B, self.n_latent, self.style_dim, device=device)
            s_buf = torch.empty(B, total, device=device)
            tab = hip.LinearTable(device, lane)
            offs, off = [], 0
            for m, li in seq:
                mod = m.conv.modulation
                tab.add(mod.weight, mod.bias, styles_buf, self.n_latent * self.style_dim, s_buf, total,
                        w_scale=mod.scale, b_scale=mod.lr_mul, x_offset=li * self.style_dim, out_offset=off)
                offs.append(off)
                off += m.conv.in_channel
            ent = (key, styles_buf, s_buf, tab, offs, total)
            self._tables[slot] = ent
        return ent[1:]

    def forward(self, features, styles, rgbd_in=None, transform=None, noise=None, mesh_path=None):
        B = features.shape[0]
        dev = features.device
        styles_buf, s_buf, tab, offs, total = self._style_table(B, dev)
        styles_buf.copy_(styles)
        tab.run(B)
        seq = self._mod_layers()
        if noise is None:
            noise = [None] * self.num_layers

        def wm_of(idx, H, W):
            m = seq[idx][0]
            conv = m.conv
            packed = isinstance(m, StyledConv) and (conv.fast(H * W) or conv.tiled3x3(H, W))
            if isinstance(m, StyledConv) and not packed:
                return None
            return hip.modulate_weights(conv.weight, s_buf, total, B, conv.out_channel, conv.in_channel,
                                        conv.kernel_size ** 2, conv.scale, conv.demodulate, packed, s_offset=offs[idx],
                                        flip=packed and conv.kernel_size == 3 and conv.upsample,
                                        split=packed and isinstance(m, StyledConv) and m.split and not m.bf16)

        H, W = features.shape[2], features.shape[3]
        out = self.conv1(features, styles[:, 0], noise=noise[0], wm=wm_of(0, H, W))
        skip = self.to_rgb1(out, styles[:, 1], skip=rgbd_in, wm=wm_of(1, H, W))
        i, q = 1, 2
        for s in range(len(self.to_rgbs)):
            c1, c2, tr = self.convs[2 * s], self.convs[2 * s + 1], self.to_rgbs[s]
            out = c1(out, styles[:, i], noise=noise[2 * s + 1], wm=wm_of(q, H, W))
            H, W = out.shape[2], out.shape[3]
            out = c2(out, styles[:, i + 1], noise=noise[2 * s + 2], wm=wm_of(q + 1, H, W))
            skip = tr(out, styles[:, i + 2], skip=skip, wm=wm_of(q + 2, H, W))
            i += 2
            q += 3
        return skip
