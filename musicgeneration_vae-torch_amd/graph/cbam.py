"""CBAM for the MI355X build (reference: graph/cbam.py).  The three classes keep the
reference's names and parameters; ``CBAM`` runs channel + spatial attention AND the
residual/activation that every caller applies right after it as one fused HIP op."""
from torch import nn

from hipops import functional as HF
from graph.layers import Conv2d
from graph.weights_initializer import weights_init


class ChannelAttention(nn.Module):
    """parameters of graph/cbam.py:7-29: conv1 [C/16,C,1,1], conv2 [C,C/16,1,1]"""

    def __init__(self, channel):
        super().__init__()
        self.conv1 = Conv2d(channel, channel // 16, 1, bias=False)
        self.conv2 = Conv2d(channel // 16, channel, 1, bias=False)
        self.apply(weights_init)

    def forward(self, x):
        """x * sigmoid(MLP(avg_hw x) + MLP(max_hw x)) -- the channel half of the fused CBAM op"""
        return HF.cbam(x, self.conv1.weight, self.conv2.weight, None, 0, parts=1)


class SpatialAttention(nn.Module):
    """parameters of graph/cbam.py:32-52: conv [1,2,3,3]"""

    def __init__(self):
        super().__init__()
        self.conv = Conv2d(2, 1, 3, padding=1, bias=False)
        self.apply(weights_init)

    def forward(self, x):
        """x * sigmoid(conv3x3([mean_c x, max_c x])) -- the spatial half of the fused CBAM op"""
        return HF.cbam(x, None, None, self.conv.weight, 0, parts=2)


class CBAM(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.channel_attention = ChannelAttention(channel)
        self.spatial_attention = SpatialAttention()
        self.apply(weights_init)

    def fused(self, u, mode, res=None, act=HF.ACT_NONE, slope=0.01, out=None):
        """mode 1: act(u + cbam(u)); mode 2: act(res + cbam(u)); mode 0: cbam(u)"""
        ca, sa = self.channel_attention, self.spatial_attention
        return HF.cbam(u, ca.conv1.weight, ca.conv2.weight, sa.conv.weight, mode, res, act, slope, out)

    def fused_norm(self, x, norm, mode, res=None, act=HF.ACT_NONE, slope=0.01, out=None, channels_last=False):
        """fused(norm(x), ...) as one node: the InstanceNorm kernel also does the channel pooling"""
        ca, sa = self.channel_attention, self.spatial_attention
        if channels_last:
            return HF.norm_cbam_cl(x, norm.weight, norm.bias, ca.conv1.weight, ca.conv2.weight, sa.conv.weight, norm.eps, mode,
                                   res, act, slope, out)
        return HF.norm_cbam(x, norm.weight, norm.bias, ca.conv1.weight, ca.conv2.weight, sa.conv.weight, norm.eps, mode,
                            res, act, slope, out)

    def forward(self, x):
        return self.fused(x, 0)
