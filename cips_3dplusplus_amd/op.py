"""Op-level Python API of the reference's `op` package on top of the HIP C ABI.

Same names, arguments and error behaviour as /root/reference/exp/op/__init__.py:1-2:
    fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2**0.5)     op/fused_act.py:104-119
    FusedLeakyReLU(channel, bias=True, negative_slope=0.2, scale=2**0.5)     op/fused_act.py:87-101
    upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))                       op/upfirdn2d.py:146-157
The autograd wiring follows op/fused_act.py:20-84 and op/upfirdn2d.py:20-143: the backward of
bias+leaky-ReLU is the same kernel in mode (act 3, grad 1) keyed on the saved OUTPUT, the backward
of upfirdn2d is upfirdn2d with the flipped FIR and swapped up/down factors.

Device dispatch as in the reference (op/fused_act.py:105-116, op/upfirdn2d.py:147-150): a CPU tensor takes a pure-torch
evaluation of the same formula (differentiable through torch's own autograd), a GPU tensor the HIP kernels.  This is the
op-level API's own behaviour, not a fallback: a GPU tensor never takes the torch branch, a missing or stale HIP library still
raises (see _lib), and nothing else in the package accepts CPU tensors (hip.py / Generator refuse them).
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn import functional as F

from . import _lib


# ------------------------------------------------------------------------------------------ CPU tensors (torch ops)
def _fused_leaky_relu_cpu(x, bias, negative_slope, scale):
    """scale * leaky_relu(x + bias): the bias runs along dim 1."""
    if bias is not None:
        x = x + bias.reshape(1, -1, *([1] * (x.ndim - 2)))
    return F.leaky_relu(x, negative_slope) * scale


def _upfirdn2d_cpu(x, kernel, up, down, pad):
    """zero-stuff by `up` (the sample first, up - 1 zeros after it), pad (negative = crop), true convolution
    with `kernel`, keep every `down`-th sample.  x [N, C, H, W]; pad = (x0, x1, y0, y1)."""
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = up, down, pad
    n, c, h, w = x.shape
    z = x.reshape(n * c, 1, h, 1, w, 1)
    z = F.pad(z, (0, ux - 1, 0, 0, 0, uy - 1)).reshape(n * c, 1, h * uy, w * ux)
    z = F.pad(z, (max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)))
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    k = torch.flip(kernel, [0, 1]).to(z.dtype).reshape(1, 1, *kernel.shape)     # conv2d correlates: flip for a convolution
    z = F.conv2d(z, k)[:, :, ::dy, ::dx]
    return z.reshape(n, c, z.shape[2], z.shape[3])


# ------------------------------------------------------------------------------------------ raw calls
def bias_act_raw(x, bias, ref, act, grad, alpha, scale):
    lib = _lib.load()
    x = x.contiguous()
    out = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    size_b = bias.numel() if bias is not None and bias.numel() else 0
    bias_c = bias.contiguous() if size_b else None                      # contiguous copies live until the launch
    ref_c = ref.contiguous() if ref is not None and ref.numel() else None
    b_ptr = _lib.dev_ptr(bias_c, "bias", allow_none=True)
    r_ptr = _lib.dev_ptr(ref_c, "refer", allow_none=True)
    _lib.check(lib.cips3d_fused_bias_act(_lib.dev_ptr(x, "input"), b_ptr, r_ptr, _lib.dev_ptr(out), x.numel(), step_b,
                                         max(size_b, 1), act, grad, float(alpha), float(scale), _lib.stream_ptr()),
               "cips3d_fused_bias_act")
    return out


def upfirdn2d_raw(x4, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x4: [major, in_h, in_w, minor] -> [major, out_h, out_w, minor] (the native binding's layout)."""
    lib = _lib.load()
    x4 = x4.contiguous()
    kernel = kernel.contiguous()
    major, in_h, in_w, minor = x4.shape
    kh, kw = kernel.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    out = torch.empty(major, out_h, out_w, minor, device=x4.device, dtype=x4.dtype)
    _lib.check(lib.cips3d_upfirdn2d(_lib.dev_ptr(x4, "input"), _lib.dev_ptr(kernel, "kernel"), _lib.dev_ptr(out), major,
                                    in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                                    pad_y1, _lib.stream_ptr()), "cips3d_upfirdn2d")
    return out


# ------------------------------------------------------------------------------------------ fused leaky relu
class _FusedLeakyReLUBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, has_bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = bias_act_raw(grad_output, None, out, 3, 1, negative_slope, scale)
        if has_bias:
            dims = [0] + list(range(2, grad_input.ndim))
            grad_bias = grad_input.sum(dims).detach()
        else:
            grad_bias = grad_output.new_empty(0)
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gg_input, gg_bias):
        out, = ctx.saved_tensors
        gg_out = bias_act_raw(gg_input, gg_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gg_out, None, None, None, None


class _FusedLeakyReLU(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        ctx.has_bias = bias is not None
        out = bias_act_raw(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = _FusedLeakyReLUBackward.apply(grad_output.contiguous(), out, ctx.has_bias,
                                                              ctx.negative_slope, ctx.scale)
        return grad_input, (grad_bias if ctx.has_bias else None), None, None


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if input.device.type == "cpu":
        return _fused_leaky_relu_cpu(input, bias, negative_slope, scale)
    return _FusedLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


# ------------------------------------------------------------------------------------------ upfirdn2d
class _UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        gx0, gx1, gy0, gy1 = g_pad
        g = grad_output.reshape(-1, out_size[0], out_size[1], 1)
        grad_input = upfirdn2d_raw(g, grad_kernel, down[0], down[1], up[0], up[1], gx0, gx1, gy0, gy1)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, in_size, out_size)
        return grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])

    @staticmethod
    def backward(ctx, gg_input):
        kernel, = ctx.saved_tensors
        up, down, pad, in_size, out_size = ctx.cfg
        gg = gg_input.reshape(-1, in_size[2], in_size[3], 1)
        gg_out = upfirdn2d_raw(gg, kernel, up[0], up[1], down[0], down[1], *pad)
        return gg_out.view(in_size[0], in_size[1], out_size[0], out_size[1]), None, None, None, None, None, None, None, None


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        _, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
        out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (px0, px1, py0, py1)
        ctx.g_pad = (kw - px0 - 1, in_w * up_x - out_w * down_x + px0 - up_x + 1,
                     kh - py0 - 1, in_h * up_y - out_h * down_y + py0 - up_y + 1)
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        out = upfirdn2d_raw(input.reshape(-1, in_h, in_w, 1), kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        grad_input = _UpFirDn2dBackward.apply(grad_output.contiguous(), kernel, grad_kernel, ctx.up, ctx.down, ctx.pad,
                                              ctx.g_pad, ctx.in_size, ctx.out_size)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if input.device.type == "cpu":
        return _upfirdn2d_cpu(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
    return _UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
