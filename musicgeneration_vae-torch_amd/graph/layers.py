"""Parameter-holding leaf layers of the MI355X build.

They keep torch's class NAMES (so the reference's ``weights_init`` name matching --
graph/weights_initializer.py:5-23 -- behaves identically), parameter names, shapes and
default initialisation, but their forward is a HIP launch through hipops.functional.
"""
import math

import torch
from torch import nn

from hipops import functional as HF


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _default_init(weight, bias, fan_in):
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    if bias is not None:
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        nn.init.uniform_(bias, -bound, bound)


class Conv2d(nn.Module):
    """``channels_last=True``: the layer works on channels-last activations and keeps its weight STORED [Cout, KH, KW, Cin]
    (torch.channels_last: same logical shape, same state_dict entry) -- the K-contiguous operand of csrc/conv_nhwc.inc."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, channels_last=False):
        super().__init__()
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.channels_last = bool(channels_last)
        w = torch.empty(out_channels, in_channels, *self.kernel_size)
        if self.channels_last:
            w = torch.empty(out_channels, *self.kernel_size, in_channels).permute(0, 3, 1, 2)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        _default_init(self.weight, self.bias, in_channels * self.kernel_size[0] * self.kernel_size[1])

    def forward(self, x, act=HF.ACT_NONE, slope=0.01, out=None, in_act=None, defer_act_grad=False):
        """in_act / defer_act_grad: see hipops.functional._ConvFn (activation gradient folded into the consumer)"""
        if self.channels_last:
            return HF.conv2d_cl(x, self.weight, self.bias, self.stride, self.padding, act, slope, out, in_act, defer_act_grad)
        return HF.conv2d(x, self.weight, self.bias, self.stride, self.padding, act, slope, out, in_act, defer_act_grad)


class ConvTranspose2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True,
                 channels_last=False):
        super().__init__()
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.output_padding = _pair(padding), _pair(output_padding)
        self.channels_last = bool(channels_last)
        w = torch.empty(in_channels, out_channels, *self.kernel_size)
        if self.channels_last:          # stored [Cin, KH, KW, Cout]: see Conv2d
            w = torch.empty(in_channels, *self.kernel_size, out_channels).permute(0, 3, 1, 2)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        # torch derives fan_in of a transposed-conv weight from dim 1
        _default_init(self.weight, self.bias, out_channels * self.kernel_size[0] * self.kernel_size[1])

    def forward(self, x, act=HF.ACT_NONE, slope=0.01, out=None, in_act=None, defer_act_grad=False):
        if self.channels_last:
            return HF.conv_transpose2d_cl(x, self.weight, self.bias, self.stride, self.padding, self.output_padding, act, slope, out)
        return HF.conv_transpose2d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding, act, slope, out,
                                   in_act, defer_act_grad)


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        _default_init(self.weight, self.bias, in_features)

    def forward(self, x, act=HF.ACT_NONE, slope=0.01, out=None):
        return HF.linear(x, self.weight, self.bias, act, slope, out)


class InstanceNorm2d(nn.Module):
    """affine, no running statistics (the reference passes momentum=0.01, which is unused
    without running stats)"""

    def __init__(self, num_features, eps=1e-5, momentum=0.01, affine=True, channels_last=False):
        super().__init__()
        if not affine:
            raise ValueError("the hot path only uses affine InstanceNorm2d")
        self.eps = eps
        self.channels_last = bool(channels_last)
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))

    def forward(self, x, act=HF.ACT_NONE, slope=0.01, out=None):
        if self.channels_last:
            return HF.instance_norm_cl(x, self.weight, self.bias, self.eps, act, slope, out)
        return HF.instance_norm(x, self.weight, self.bias, self.eps, act, slope, out)


class Embedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)

    def forward(self, idx, out=None):
        return HF.embedding(idx, self.weight, out)


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d state (weight, bias, running_mean, running_var, num_batches_tracked) and
    semantics; the class name matters: the reference's weights_init matches 'BatchNorm'."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, act=HF.ACT_NONE, slope=0.01):
        if self.training:
            self.num_batches_tracked += 1
        return HF.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                             self.momentum, self.eps, act, slope)
