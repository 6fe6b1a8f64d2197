"""Bar decoder on HIP kernels (reference: graph/decoder.py).

Transposed convolutions run on the stride-phase data-gradient kernel (no zero taps);
every ``torch.cat`` of the reference is a pre-allocated buffer whose channel slices the
producers write directly."""
import os

import torch
from torch import nn

from hipops import blocks as HB
from hipops import functional as HF
from graph.cbam import CBAM
from graph.layers import Conv2d, ConvTranspose2d, Embedding, InstanceNorm2d, Linear
from graph.weights_initializer import weights_init


def _new(channels_last, n, c, h, w, like):
    """the buffer two branches are written into side by side: same layout AND storage type (fp32 / bf16 island) as ``like``"""
    if channels_last:
        return HF.new_channels_last(n, c, h, w, like.device, like.dtype)
    return torch.empty((n, c, h, w), device=like.device, dtype=torch.float32)


class _Stem(nn.Module):
    """[B,2304,1,1] -> [B,1024,6,3] through two non-overlapping transposed convs"""
    first, second = "time", "pitch"

    def __init__(self):
        super().__init__()
        mk = {"time": lambda cin: ConvTranspose2d(cin, 1024, (6, 1), stride=(6, 1), bias=False),
              "pitch": lambda cin: ConvTranspose2d(cin, 1024, (1, 3), stride=(1, 3), bias=False)}
        setattr(self, self.first, mk[self.first](2304))
        setattr(self, self.second, mk[self.second](1024))
        self.bn = InstanceNorm2d(1024, eps=1e-5, momentum=0.01, affine=True)
        self.cbam = CBAM(1024)
        self.apply(weights_init)

    def forward(self, x, out=None):
        o = getattr(self, self.first)(x, act=HF.ACT_RELU, defer_act_grad=True)      # ReLU' applied by the consumer below
        o = getattr(self, self.second)(o, in_act=(HF.ACT_RELU, 0.0))
        return self.cbam.fused_norm(o, self.bn, 1, act=HF.ACT_RELU, out=out)


class TimePitchModule(_Stem):
    """graph/decoder.py:8-36"""
    first, second = "time", "pitch"


class PitchTimeModule(_Stem):
    """graph/decoder.py:39-66"""
    first, second = "pitch", "time"


class DeConvModule(nn.Module):
    """graph/decoder.py:69-109: ConvT4x4 s2 || ConvT3x3 s2 (+bias) -> IN -> ReLU each; cat -> 1x1 -> IN -> +CBAM -> ReLU"""

    def __init__(self, in_channel, out_channel, channels_last=False):
        super().__init__()
        cl = self.channels_last = bool(channels_last)
        self.deConv1 = ConvTranspose2d(in_channel, out_channel, 4, stride=2, padding=1, bias=False, channels_last=cl)
        self.deConv2 = ConvTranspose2d(in_channel, out_channel, 3, stride=2, padding=1, output_padding=1, bias=True, channels_last=cl)
        self.conv = Conv2d(in_channel, out_channel, 1, stride=1, bias=False, channels_last=cl)
        self.bn1 = InstanceNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True, channels_last=cl)
        self.bn2 = InstanceNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True, channels_last=cl)
        self.bn3 = InstanceNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True, channels_last=cl)
        self.cbam = CBAM(out_channel)
        self.out_channel = out_channel
        self.apply(weights_init)

    def forward(self, x, out=None):
        if self.channels_last and out is None and HB.usable(x):
            return HB.deconv_block(x, self, False)
        co = self.out_channel
        n, _, h, w = x.shape
        cat = _new(self.channels_last, n, 2 * co, 2 * h, 2 * w, x)
        with HF.forked_branch(x, cat):            # the two transposed-conv branches are independent until the cat
            b = self.bn2(self.deConv2(x), act=HF.ACT_RELU, out=cat[:, co:])
        a = self.bn1(self.deConv1(x), act=HF.ACT_RELU, out=cat[:, :co])
        HF.join_side_streams(slot=0)
        return self.cbam.fused_norm(self.conv(HF.join(cat, a, b)), self.bn3, 1, act=HF.ACT_RELU, out=out,
                                    channels_last=self.channels_last)


class DeConvPitchPadding(nn.Module):
    """graph/decoder.py:112-154.  Literal about reference defect D5: ``bn2`` normalises BOTH
    branches and ``bn1`` is never used (its parameters never receive a gradient)."""

    def __init__(self, in_channel, out_channel, channels_last=False):
        super().__init__()
        cl = self.channels_last = bool(channels_last)
        self.deConv1 = ConvTranspose2d(in_channel, out_channel, 4, stride=2, padding=1, output_padding=(0, 1), bias=True, channels_last=cl)
        self.deConv2 = ConvTranspose2d(in_channel, out_channel, 4, stride=2, padding=1, output_padding=(0, 1), bias=True, channels_last=cl)
        self.conv = Conv2d(in_channel, out_channel, 1, stride=1, bias=False, channels_last=cl)
        self.bn1 = InstanceNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True, channels_last=cl)
        self.bn2 = InstanceNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True, channels_last=cl)
        self.bn3 = InstanceNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True, channels_last=cl)
        self.cbam1 = CBAM(out_channel)
        self.cbam2 = CBAM(out_channel)
        self.out_channel = out_channel
        self.apply(weights_init)

    def forward(self, x, out=None):
        if self.channels_last and out is None and HB.usable(x):
            return HB.deconv_block(x, self, True)
        co = self.out_channel
        n, _, h, w = x.shape
        cat = _new(self.channels_last, n, 2 * co, 2 * h, 2 * w + 1, x)
        with HF.forked_branch(x, cat):
            b = self.bn2(self.deConv2(x), act=HF.ACT_RELU, out=cat[:, co:])
        a = self.cbam1.fused_norm(self.deConv1(x), self.bn2, 1, act=HF.ACT_RELU, out=cat[:, :co], channels_last=self.channels_last)
        HF.join_side_streams(slot=0)
        return self.cbam2.fused_norm(self.conv(HF.join(cat, a, b)), self.bn3, 1, act=HF.ACT_RELU, out=out,
                                     channels_last=self.channels_last)


class Decoder(nn.Module):
    """graph/decoder.py:157-222: (z, pre_z, phrase_feature, position) -> gen [B,1,96,60] in (0,1)"""

    def __init__(self, layers, channels_last=None):
        super().__init__()
        # fit1 and the four up-sampling blocks (94 % of the decoder's MACs) run on channels-last tensors, like the encoder
        # trunks (graph/encoder.py), and so does fit2 (64 -> 1, hipops.functional.conv2d_to1_cl); the Linear head and the two
        # stems (1x1 maps: plain GEMMs) stay NCHW
        cl = self.channels_last = (os.environ.get("MGVAE_LAYOUT", "nhwc") != "nchw") if channels_last is None else bool(channels_last)
        self.bar_linear = Linear(1152 * 2, 1152)
        self.phrase_linear = Linear(1152 * 2, 1152)
        self.time = TimePitchModule()
        self.pitch = PitchTimeModule()
        self.fit1 = Conv2d(2048, 1024, 1, stride=1, bias=False, channels_last=cl)
        self.bn = InstanceNorm2d(1024, eps=1e-5, momentum=0.01, affine=True)
        self.fit2 = Conv2d(64, 1, 1, stride=1, bias=False)
        blocks = []
        for i in range(1, len(layers)):
            blocks.append((DeConvPitchPadding if i < 3 else DeConvModule)(layers[i - 1], layers[i], cl))
        self.layers = nn.ModuleList(blocks)
        self.cbam = CBAM(1024)
        self.position_embedding = Embedding(332, 1152)
        nn.init.uniform_(self.position_embedding.weight, -1.0, 1.0)
        self.dropout_p = 0.3
        self._drop_masks = None     # tests inject two pre-scaled masks [B,1152] here
        self.apply(weights_init)

    def _drop(self, t, i):
        m = self._drop_masks[i] if self._drop_masks is not None else None
        return HF.dropout(t, self.dropout_p, self.training, mask=m)

    def head(self, z, pre_z, phrase_feature, position):
        """graph/decoder.py:192-205: the two Linear + ReLU + Dropout branches and their concat -> [n, 2304, 1, 1]"""
        n, dev = z.shape[0], z.device
        new = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
        # cat(phrase_feature, embedding(position)) -> Linear -> ReLU -> Dropout
        pbuf = new(n, 2304)
        pa = HF.copy_into(phrase_feature, pbuf[:, :1152])
        pb = self.position_embedding(position, out=pbuf[:, 1152:])
        pf = self._drop(self.phrase_linear(HF.join(pbuf, pa, pb), act=HF.ACT_RELU), 0)
        # cat(z, pre_z) -> Linear -> ReLU -> Dropout
        bbuf = new(n, 2304)
        ba = HF.copy_into(z, bbuf[:, :1152])
        bb = HF.copy_into(pre_z, bbuf[:, 1152:])
        bf = self._drop(self.bar_linear(HF.join(bbuf, ba, bb), act=HF.ACT_RELU), 1)
        xbuf = new(n, 2304)
        xa = HF.copy_into(bf, xbuf[:, :1152])
        xb = HF.copy_into(pf, xbuf[:, 1152:])
        return HF.join(xbuf, xa, xb).view(n, 2304, 1, 1)

    def stems(self, x):
        """graph/decoder.py:207-211: the two stems side by side -> [n, 2048, 6, 3] (NCHW)"""
        n = x.shape[0]
        cat = torch.empty((n, 2048, 6, 3), device=x.device, dtype=torch.float32)
        with HF.forked_branch(x, cat):            # the two stems are independent until fit1
            time = self.time(x, out=cat[:, 1024:])
        pitch = self.pitch(x, out=cat[:, :1024])
        HF.join_side_streams(slot=0)
        return HF.join(cat, pitch, time)

    def fit_stage(self, o):
        """graph/decoder.py:213-215: fit1 -> InstanceNorm -> +CBAM -> ReLU (``o`` in the island's layout)"""
        if self.channels_last and HB.usable(o):
            return HB.conv_norm_cbam_block(o, self.fit1, self.bn, self.cbam)
        return self.cbam.fused_norm(self.fit1(o), self.bn, 1, act=HF.ACT_RELU, channels_last=self.channels_last)

    def forward(self, z, pre_z, phrase_feature, position):
        if self.channels_last and HB.front_usable(z):
            o = HB.decoder_front(self, z, pre_z, phrase_feature, position)     # head + stems + layout change: one node
        else:
            o = self.stems(self.head(z, pre_z, phrase_feature, position))
            if self.channels_last:
                o = HF.to_channels_last(o)
        o = self.fit_stage(o)
        for blk in self.layers:
            o = blk(o)
        if self.channels_last:             # fit2 + Sigmoid read the channels-last map directly: no layout change at the exit
            return HF.conv2d_to1_cl(o, self.fit2.weight, HF.ACT_SIGMOID)
        return self.fit2(o, act=HF.ACT_SIGMOID)
