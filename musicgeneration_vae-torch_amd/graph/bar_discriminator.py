"""Bar-pair discriminator on HIP kernels (reference: graph/bar_discriminator.py).

Three small conv towers over a 2-bar roll [B,1,192,60] -- chord (pitch folded 60 -> 12 groups
of 5 ADJACENT pitches, as the reference's view/sum really does), on/off (the reference's
``x[:, :-1]`` slices the size-1 CHANNEL axis, so the branch sees sum-over-pitch of x; restated
literally), basic -- each ending in a whole-map average, then Linear 192 -> 1 + sigmoid.
BatchNorm2d uses per-process batch statistics (like the reference under nn.DataParallel)."""
from torch import nn

from hipops import functional as HF
from hipops import netchain as NC
from graph.layers import BatchNorm2d, Conv2d, Linear
from graph.weights_initializer import weights_init

R = HF.ACT_RELU


def _pool_all(x, hw):
    if tuple(x.shape[2:]) != hw:
        raise RuntimeError("AvgPool2d%s expects a %s map, got %s" % (hw, hw, tuple(x.shape[2:])))
    return HF.global_avg_pool(x)


class ChordFeature(nn.Module):
    """graph/bar_discriminator.py:7-58"""

    def __init__(self):
        super().__init__()
        self.chord_conv1 = Conv2d(1, 8, (3, 1), stride=(2, 1), padding=(1, 0), bias=False)
        self.chord_conv2 = Conv2d(8, 16, (3, 1), stride=(2, 1), padding=(1, 0), bias=False)
        self.chord_fit = Conv2d(16, 16, 1, stride=1, bias=False)
        self.chord_conv3 = Conv2d(16, 32, 3, stride=2, padding=1, bias=False)
        self.chord_conv4 = Conv2d(32, 64, 3, stride=2, padding=1, bias=False)
        for i, c in enumerate((8, 16, 16, 32, 64)):
            setattr(self, "batch_norm%d" % (i + 1), BatchNorm2d(c, eps=1e-5, momentum=0.01, affine=True))
        self.apply(weights_init)

    def forward(self, x):
        o = HF.group_sum(x, 5)                                    # [B,1,192,12]
        o = self.batch_norm1(self.chord_conv1(o), act=R)
        o = self.batch_norm2(self.chord_conv2(o), act=R)
        o = self.batch_norm3(self.chord_fit(o), act=R)
        o = self.batch_norm4(self.chord_conv3(o), act=R)
        o = self.batch_norm5(self.chord_conv4(o), act=R)
        return _pool_all(o, (12, 3))


class OnOffFeature(nn.Module):
    """graph/bar_discriminator.py:61-100"""

    def __init__(self):
        super().__init__()
        self.onoff_conv1 = Conv2d(1, 8, 3, stride=(2, 1), padding=1, bias=False)
        self.onoff_conv2 = Conv2d(8, 8, 3, stride=(2, 1), padding=1, bias=False)
        self.batch_norm2 = BatchNorm2d(8)
        self.onoff_conv3 = Conv2d(8, 16, 3, stride=(2, 1), padding=1, bias=False)
        self.onoff_conv4 = Conv2d(16, 32, 3, stride=(2, 1), padding=1, bias=False)
        self.onoff_fit = Conv2d(32, 32, 1, stride=1, bias=False)
        self.onoff_conv5 = Conv2d(32, 64, 3, stride=(2, 1), padding=1, bias=False)
        self.apply(weights_init)

    def forward(self, x):
        if x.shape[1] != 1:
            raise RuntimeError("OnOffFeature restates the reference for single-channel rolls")
        o = HF.group_sum(x, x.shape[3])                           # [B,1,192,1]
        o = self.onoff_conv2(self.onoff_conv1(o, act=R), act=R)
        o = self.batch_norm2(o)
        o = self.onoff_conv4(self.onoff_conv3(o, act=R), act=R)
        o = self.onoff_conv5(self.onoff_fit(o, act=R), act=R)
        return _pool_all(o, (6, 1))


class ConvModule(nn.Module):
    """graph/bar_discriminator.py:103-134"""

    def __init__(self, in_channel, out_channel, isBasic=True):
        super().__init__()
        if not isBasic:
            self.conv1 = Conv2d(in_channel, in_channel, 3, stride=1, padding=1, bias=False)
        self.conv2 = Conv2d(in_channel, out_channel, 3, stride=2, padding=1, bias=False)
        self.bn1 = BatchNorm2d(in_channel, eps=1e-5, momentum=0.01, affine=True)
        self.bn2 = BatchNorm2d(out_channel, eps=1e-5, momentum=0.01, affine=True)
        self.isBasic = isBasic
        self.apply(weights_init)

    def forward(self, x):
        o = x if self.isBasic else self.bn1(self.conv1(x), act=R)
        return self.bn2(self.conv2(o), act=R)


class BasicFeature(nn.Module):
    """graph/bar_discriminator.py:137-183"""

    def __init__(self, layers):
        super().__init__()
        self.pitch1 = Conv2d(1, 8, (1, 4), stride=(1, 2), padding=(0, 1), bias=False)
        self.pitch2 = Conv2d(8, 8, (4, 1), stride=(2, 1), padding=(1, 0), bias=False)
        self.time1 = Conv2d(1, 8, (4, 1), stride=(2, 1), padding=(1, 0), bias=False)
        self.time2 = Conv2d(8, 8, (1, 4), stride=(1, 2), padding=(0, 1), bias=False)
        self.fit = Conv2d(16, 8, 1, stride=1, bias=False)
        self.bn = BatchNorm2d(8, eps=1e-5, momentum=0.01, affine=True)
        self.layers = nn.ModuleList([ConvModule(layers[i - 1], layers[i], i >= 3) for i in range(1, len(layers))])
        self.apply(weights_init)

    def forward(self, x):
        import torch
        n, _, h, w = x.shape
        cat = torch.empty((n, 16, h // 2, w // 2), device=x.device, dtype=torch.float32)
        p = self.pitch2(self.pitch1(x, act=R), act=R, out=cat[:, :8])
        t = self.time2(self.time1(x, act=R), act=R, out=cat[:, 8:])
        o = self.bn(self.fit(HF.join(cat, p, t)), act=R)
        for layer in self.layers:
            o = layer(o)
        return _pool_all(o, (12, 4))


class BarDiscriminator(nn.Module):
    """graph/bar_discriminator.py:186-217"""

    def __init__(self):
        super().__init__()
        self.chord = ChordFeature()
        self.onoff = OnOffFeature()
        self.basic = BasicFeature([8, 16, 32, 64])
        self.linear = Linear(64 * 3, 1, bias=False)
        self.apply(weights_init)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_mg_chain_cache", None)       # .to() / .cuda() re-create the buffers the launch chain points at
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x):
        import torch
        x = x.reshape(-1, 1, 96 * 2, 60)
        if NC.usable(self, x):
            return NC.bar_discriminator(self, x)      # the three towers as one autograd node and one launch chain per direction
        n = x.shape[0]
        feat = torch.empty((n, 192), device=x.device, dtype=torch.float32)
        a = HF.copy_into(self.chord(x), feat[:, :64])
        b = HF.copy_into(self.onoff(x), feat[:, 64:128])
        c = HF.copy_into(self.basic(x), feat[:, 128:])
        return self.linear(HF.join(feat, a, b, c), act=HF.ACT_SIGMOID)
