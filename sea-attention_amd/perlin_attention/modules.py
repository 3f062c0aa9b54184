"""Predictor building blocks (reference: src/models/perlin_attention/modules.py:12-192).

Parameter / buffer names and shapes are kept so that trained SEA checkpoints load unchanged
(`weight` is (out, in, 2k-1, k) with a `weight_mask` buffer for the causal conv, `net` inside
KeepRes).  The computation is organised for MI355X/MIOpen instead of mirrored:

* CausalConv2d: the reference convolves with the full (2k-1) x k kernel whose lower k-1 rows are
  masked to zero on every call (`weight.masked_fill`, modules.py:173) and pads both sides of the
  time axis.  Here only the live k x k slice is convolved with top-only padding -- identical
  output, half the MACs, no per-call masked copy of the weight.
"""
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

# process-global switches read by callers (src/main/opt_generate.py:22-23, benchmark_opt_ablation.py:2-4)
BENCHMARKING = False
CAUSAL_CONV_FORCE_NON_CAUSAL = False


def interpolate(x: torch.Tensor, size, interp_mode: str = None):
    """modules.py:12-31: bilinear when widening, area when shrinking; fp32 detour for bf16 autocast."""
    if x.shape[-2:] == size:
        return x
    if interp_mode is None:
        interp_mode = 'bilinear' if size[-1] >= x.shape[-1] else 'area'
    if not BENCHMARKING and x.dtype == torch.bfloat16:
        return F.interpolate(x.float(), size, mode=interp_mode).to(torch.bfloat16)
    return F.interpolate(x, size, mode=interp_mode)


class Residual(nn.Module):
    def __init__(self, *args) -> None:
        super().__init__()
        self.net = nn.Sequential(*args)

    def forward(self, x):
        return x + self.net(x)


class KeepRes(nn.Module):
    """Run `net`, then resize back to the input height and `output_width` (modules.py:42-55)."""

    def __init__(self, *args, output_width=None):
        super().__init__()
        self.net = nn.Sequential(*args)
        self.output_width = output_width

    def forward(self, x):
        h, w = x.shape[-2:]
        y = self.net(x)
        return interpolate(y, (h, w if self.output_width is None else self.output_width))


class UpsampleFP32(nn.Module):
    """Nearest-neighbour upsample (modules.py:77-92).  `dtype` is kept for signature parity."""

    def __init__(self, scale, dtype=torch.float32):
        super().__init__()
        self.scale = scale
        self.dtype = dtype

    def forward(self, x):
        sh, sw = self.scale if isinstance(self.scale, (tuple, list)) else (self.scale, self.scale)
        if float(sh).is_integer() and float(sw).is_integer():
            # nearest upsample by integer factors == repeat_interleave (exact for every dtype)
            if int(sh) != 1:
                x = x.repeat_interleave(int(sh), dim=-2)
            if int(sw) != 1:
                x = x.repeat_interleave(int(sw), dim=-1)
            return x
        return F.interpolate(x, scale_factor=self.scale, mode='nearest')


class CausalConv2d(nn.Module):
    """2-D convolution that is causal along the height (= time) axis (modules.py:96-192)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 padding_mode: str = 'zeros', dilation: int = 1, causal: bool = False):
        super().__init__()
        self.causal = causal and not CAUSAL_CONV_FORCE_NON_CAUSAL
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = (padding, padding)
        self.padding_mode = padding_mode
        self.dilation = dilation

        init = nn.Conv2d(in_channels, out_channels, kernel_size)      # PyTorch's default initialiser
        self.bias = nn.Parameter(init.bias.data)
        if not causal:
            self.weight = nn.Parameter(init.weight.data)
        else:
            k = kernel_size
            full = torch.zeros((out_channels, in_channels, 2 * k - 1, k))
            full[:, :, :k, :] = init.weight.data
            self.weight = nn.Parameter(full)
            live = torch.zeros_like(full)
            live[:, :, :k, :] = 1.0
            self.register_buffer('weight_mask', live)
            d = dilation if isinstance(dilation, (int, float)) else dilation[0]
            self.padding = ((k - 1) * d, padding)

    def forward(self, x: torch.Tensor):
        if not self.causal:
            if self.padding_mode != 'zeros':
                ph, pw = self.padding
                x = F.pad(x, (ph, ph, pw, pw), mode=self.padding_mode)
                return F.conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation)
            return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)
        k = self.kernel_size
        ph, pw = self.padding
        w = self.weight[:, :, :k, :]                      # rows k.. are the masked (dead) half
        if w.dtype != x.dtype:
            w = w.to(x.dtype)
        b = self.bias if self.bias.dtype == x.dtype else self.bias.to(x.dtype)
        if k == 1 and self.stride in (1, (1, 1)):
            # a 1x1 convolution is a channel GEMM: (O x C) @ (C x T*W) per batch item.  Besides being the
            # cheaper form on rocBLAS it keeps pixels independent bit for bit -- the MIOpen/CK 1x1 kernel was
            # observed to couple vertically adjacent pixels at rounding level, which is enough to flip a
            # top-k decision in the row before a perturbed token (causality canary).
            if pw > 0:
                x = F.pad(x, (pw, pw))
            N, C, T, W = x.shape
            y = torch.matmul(w.view(self.out_channels, C), x.reshape(N, C, T * W))
            return (y + b.view(1, -1, 1)).view(N, self.out_channels, T, W)
        if ph > 0:
            x = F.pad(x, (0, 0, ph, 0))                   # past rows only
        return F.conv2d(x, w, b, self.stride, (0, pw), self.dilation)


class ResBlock(nn.Module):
    def __init__(self, ch, padding=1, lnorm_size=None, padding_mode='zeros', causal=False, dilation=1):
        super().__init__()
        self.net = KeepRes(
            CausalConv2d(ch, ch, 3, padding=padding, padding_mode=padding_mode, causal=causal, dilation=dilation),
            nn.ReLU(),
            CausalConv2d(ch, ch, 3, padding=padding, padding_mode=padding_mode, causal=causal, dilation=dilation),
        )
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.relu(self.net(x) + x)
