"""Encoder building blocks on HIP kernels (reference: graph/encodingBlock.py).

Each block is: implicit-GEMM conv(s) with the inner activation fused in the conv
epilogue -> InstanceNorm -> fused CBAM + residual + activation.  ``out=`` lets a block
write its result straight into a channel slice of its consumer's concat buffer."""
from torch import nn

from hipops import blocks as HB
from hipops import functional as HF
from graph.cbam import CBAM
from graph.layers import Conv2d, InstanceNorm2d
from graph.weights_initializer import weights_init


class _Stem(nn.Module):
    """two thin 1-D convs; subclasses fix which axis goes first.  ``channels_last=True``: the stem belongs to the island --
    its first conv (one input channel) writes channels-last directly (hipops.functional.conv2d_c1_cl), the 32 -> 32 conv runs on
    the island's matrix kernels with its weight stored [Cout, KH, KW, Cin], InstanceNorm / CBAM are the channels-last kernels."""
    first, second = "time", "pitch"

    def __init__(self, channels_last=False):
        super().__init__()
        cl = self.channels_last = bool(channels_last)
        convs = {"time": lambda cin: Conv2d(cin, 32, (4, 1), stride=(2, 1), padding=(1, 0), bias=False, channels_last=cl and cin != 1),
                 "pitch": lambda cin: Conv2d(cin, 32, (1, 4), stride=(1, 2), padding=(0, 1), bias=False, channels_last=cl and cin != 1)}
        setattr(self, self.first, convs[self.first](1))
        setattr(self, self.second, convs[self.second](32))
        self.bn = InstanceNorm2d(32, eps=1e-5, momentum=0.01, affine=True)
        self.cbam = CBAM(32)
        self.apply(weights_init)

    def forward(self, x, out=None):
        c1, c2 = getattr(self, self.first), getattr(self, self.second)
        if self.channels_last:
            # the LeakyReLU gradient of the first conv is applied by the second conv's data gradient (its only consumer)
            o = HF.conv2d_c1_cl(x, c1.weight, c1.stride, c1.padding, HF.ACT_LEAKY, 0.01, defer_act_grad=True)
            o = c2(o, in_act=(HF.ACT_LEAKY, 0.01))
            return self.cbam.fused_norm(o, self.bn, 1, act=HF.ACT_LEAKY, slope=0.01, out=out, channels_last=True)
        # (no deferred activation gradient here: the second conv's stride-2 data gradient stores every other pixel, and
        # the masked epilogue costs it more -- 190 vs 106 us -- than the separate LeakyReLU-gradient pass it would save)
        o = c1(x, act=HF.ACT_LEAKY, slope=0.01)
        o = c2(o)
        return self.cbam.fused_norm(o, self.bn, 1, act=HF.ACT_LEAKY, slope=0.01, out=out)


class TimePitchModule(_Stem):
    """graph/encodingBlock.py:8-36: conv(4,1)s(2,1) -> LeakyReLU -> conv(1,4)s(1,2) -> IN -> +CBAM -> LeakyReLU"""
    first, second = "time", "pitch"


class PitchTimeModule(_Stem):
    """graph/encodingBlock.py:39-67: same with the axes swapped"""
    first, second = "pitch", "time"


class ResidualModule(nn.Module):
    """graph/encodingBlock.py:70-100: relu(x + CBAM(IN(conv2(relu(conv1(x))))))"""

    def __init__(self, channel, channels_last=False):
        super().__init__()
        self.channels_last = bool(channels_last)
        self.conv1 = Conv2d(channel, channel, 3, stride=1, padding=1, bias=False, channels_last=channels_last)
        self.conv2 = Conv2d(channel, channel, 3, stride=1, padding=1, bias=False, channels_last=channels_last)
        self.bn = InstanceNorm2d(channel, eps=1e-5, momentum=0.01, affine=True)
        self.cbam = CBAM(channel)
        self.apply(weights_init)

    def forward(self, x, out=None):
        if self.channels_last and out is None and HB.usable(x):
            return HB.residual_block(x, self.conv1, self.conv2, self.bn, self.cbam)      # one node, one launch chain
        # conv1's ReLU gradient is applied by conv2's data gradient (conv2 is the only consumer of relu(conv1(x)))
        o = self.conv2(self.conv1(x, act=HF.ACT_RELU, defer_act_grad=True), in_act=(HF.ACT_RELU, 0.0))
        return self.cbam.fused_norm(o, self.bn, 2, res=x, act=HF.ACT_RELU, out=out, channels_last=self.channels_last)


class PoolingModule(nn.Module):
    """graph/encodingBlock.py:103-126: relu(u + CBAM(u)), u = IN(conv3x3 s2 (x))"""

    def __init__(self, in_channel, out_channel, channels_last=False):
        super().__init__()
        self.channels_last = bool(channels_last)
        self.conv = Conv2d(in_channel, out_channel, 3, stride=2, padding=1, bias=False, channels_last=channels_last)
        self.bn = InstanceNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True)
        self.cbam = CBAM(out_channel)
        self.apply(weights_init)

    def forward(self, x, out=None):
        if self.channels_last and out is None and HB.usable(x):
            return HB.conv_norm_cbam_block(x, self.conv, self.bn, self.cbam)
        return self.cbam.fused_norm(self.conv(x), self.bn, 1, act=HF.ACT_RELU, out=out, channels_last=self.channels_last)
