"""Refiner on HIP kernels (reference: graph/refiner.py:7-58).

The reference's ``layer2`` is ``Conv2d(1, 8, ...)`` but is fed the 2-channel output of
``layer1`` (SURVEY defect D2), so ``graph.model.Model.forward`` raises as committed.  This module
implements the evident intent -- ``Conv2d(2, 8, ...)`` -- and is only used when the caller asks
for it (``Model(use_refiner=True)``).  Parity status: UNPINNED (the reference cannot produce a
number); checked against the oracle's restatement of the same intent."""
from torch import nn

from hipops import functional as HF
from graph.layers import BatchNorm2d, Conv2d, ConvTranspose2d, Linear
from graph.weights_initializer import weights_init


class _Slot(nn.Module):
    def forward(self, x):
        return x


def _seq(*mods):
    return nn.ModuleList(list(mods))


class Refiner(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer1 = _seq(Conv2d(1, 2, 4, padding=2), BatchNorm2d(2), _Slot(), _Slot())
        self.layer2 = _seq(Conv2d(2, 8, 4, padding=2), BatchNorm2d(8), _Slot(), _Slot())      # D2 fix: 2 input channels
        self.layer3 = _seq(Linear(2880, 1024), _Slot())
        self.layer4 = _seq(Linear(1024, 2880), _Slot())
        self.layer5 = _seq(ConvTranspose2d(8, 2, 4, stride=2, padding=1, bias=False), BatchNorm2d(2), _Slot())
        self.layer6 = _seq(ConvTranspose2d(2, 1, 4, stride=2, padding=1, bias=False), BatchNorm2d(1), _Slot())
        self.apply(weights_init)

    def forward(self, x):                                   # x [B,1,96,60]
        n = x.shape[0]
        x_2 = HF.maxpool2(self.layer1[1](self.layer1[0](x), act=HF.ACT_LEAKY, slope=0.2))        # [B,2,48,30]
        x_8 = HF.maxpool2(self.layer2[1](self.layer2[0](x_2), act=HF.ACT_LEAKY, slope=0.2))      # [B,8,24,15]
        f = self.layer3[0](x_8.reshape(n, 2880), act=HF.ACT_RELU)
        f = self.layer4[0](f, act=HF.ACT_RELU)
        x_8_t = HF.axpby(x_8, f.reshape(n, 8, 24, 15))
        x_2_t = HF.axpby(x_2, self.layer5[1](self.layer5[0](x_8_t), act=HF.ACT_RELU))
        y = HF.activation(self.layer6[1](self.layer6[0](x_2_t)), HF.ACT_SIGMOID)
        return HF.axpby(x, y, 0.5, 0.5)
