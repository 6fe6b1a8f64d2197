"""Bar encoder on HIP kernels (reference: graph/encoder.py:7-40)."""
import os

import torch
from torch import nn

from hipops import blocks as HB
from hipops import functional as HF
from graph.encodingBlock import PitchTimeModule, PoolingModule, ResidualModule, TimePitchModule
from graph.layers import Linear
from graph.weights_initializer import weights_init


class _ConvTrunk(nn.Module):
    """two stems -> concat -> (Residual, Pooling) x 4 -> whole-map average -> Linear"""
    pool_hw = (3, 2)
    linear_bias = True

    def __init__(self, layers, variational=False, channels_last=None):
        super().__init__()
        self.variational = bool(variational)
        # the residual / pooling trunk (78 % of the encoder's MACs) runs on channels-last tensors: K-contiguous conv
        # operands (csrc/conv_nhwc.inc) and the pixel-row InstanceNorm / CBAM kernels (csrc/norm_cbam_nhwc.inc).  The two
        # stems belong to the island too (round 3): their first conv reads the one-channel NCHW input and writes channels-last,
        # so there is no layout change at the entry.  MGVAE_LAYOUT=nchw switches the whole trunk back.
        self.channels_last = (os.environ.get("MGVAE_LAYOUT", "nhwc") != "nchw") if channels_last is None else bool(channels_last)
        self.time_pitch = TimePitchModule(self.channels_last)
        self.pitch_time = PitchTimeModule(self.channels_last)
        blocks = []
        for cin, cout in zip(layers[:-1], layers[1:]):
            blocks += [ResidualModule(cin, self.channels_last), PoolingModule(cin, cout, self.channels_last)]
        self.layers = nn.ModuleList(blocks)
        self.linear = Linear(1024, 1152, bias=self.linear_bias)
        if self.variational:
            # optional VAE head of the archived model (old/graphs/models/bar_v1/encoder.py:54-63):
            # ``linear`` gives the mean, ``var`` the log-variance; off by default so the parameter set
            # and semantics of graph/encoder.py are unchanged
            self.var = Linear(1024, 1152, bias=False)
        self.last_kl = None
        self.apply(weights_init)

    def stem_cat(self, x, cast=True):
        """the two stems side by side: [n, 1, h, w] -> [n, 64, h/2, w/2] (graph/encoder.py:27-29), in the island's layout
        (``cast=False``: left in the stems' own fp32 storage -- tests)"""
        n, _, h, w = x.shape
        if self.channels_last:
            # the stems stay fp32 storage (their 32-channel conv is below the bf16 kernels' 64-channel K tile); a bf16 island
            # begins with one cast of the concat -- the pass the layout change used to be
            cat = HF.new_channels_last(n, 64, h // 2, w // 2, x.device, torch.float32)
        else:
            cat = torch.empty((n, 64, h // 2, w // 2), device=x.device, dtype=torch.float32)
        pitch = self.pitch_time(x, out=cat[:, :32])
        time = self.time_pitch(x, out=cat[:, 32:])
        o = HF.join(cat, pitch, time)
        return HF.cast_cl(o, HF.island_dtype()) if (self.channels_last and cast) else o

    def features(self, x):
        """the conv trunk up to the pooled [n, 1024] features (everything but the final Linear)"""
        if self.channels_last and HB.entry_usable(x):
            o = HB.trunk_entry(x, self)            # both stems: one node, one launch chain
        else:
            o = self.stem_cat(x)
        # when the gradient of this tensor exists, every parameter gradient of ``layers`` and ``linear`` is enqueued:
        # the data-parallel step hooks it to start that range's all-reduce early (hipops/train.py)
        self.trunk_input = o if o.requires_grad else None
        for blk in self.layers:
            o = blk(o)
        if tuple(o.shape[2:]) != self.pool_hw:
            raise RuntimeError("AvgPool2d%s expects a %s map, got %s" % (self.pool_hw, self.pool_hw, tuple(o.shape[2:])))
        return HF.global_avg_pool_cl(o) if self.channels_last else HF.global_avg_pool(o)

    def forward(self, x):
        feat = self.features(x)
        mean = self.linear(feat)
        if not self.variational:
            return mean
        # reparameterisation sampler + KL term as one HIP op (eps from the Philox stream)
        z, self.last_kl = HF.reparam_kl(mean, self.var(feat))
        return z if self.training else mean


class Encoder(_ConvTrunk):
    """[B,1,96,60] -> z [B,1152]; AvgPool2d((3,2)) + Linear(1024,1152) with bias"""
    pool_hw = (3, 2)
    linear_bias = True
