"""Latent discriminators (reference: graph/z_discriminator.py:7-54): MLP
1152-512-512-512-512-1 with ReLU between and a sigmoid head; every Linear (+activation)
is one implicit-GEMM HIP launch.  ``net`` keeps the reference's Sequential indices
(0,2,4,6,8) so state_dict keys match."""
from torch import nn

from hipops import blocks as HB
from hipops import functional as HF
from graph.layers import Linear
from graph.weights_initializer import weights_init


class _Slot(nn.Module):
    """parameter-free placeholder for the ReLU / Sigmoid entries of the reference's Sequential"""

    def forward(self, x):
        return x


class _ZDisc(nn.Module):
    def __init__(self, z_dim=1152):
        super().__init__()
        self.z_dim = z_dim
        dims = (z_dim, 512, 512, 512, 512, 1)
        mods = []
        for i in range(5):
            mods += [Linear(dims[i], dims[i + 1]), _Slot()]
        self.net = nn.ModuleList(mods)
        self.apply(weights_init)

    def forward(self, x):
        if HB.mlp_usable(x):         # the five Linears as one autograd node and one launch chain per direction
            return HB.mlp(x, [(self.net[i].weight, self.net[i].bias, HF.ACT_RELU if i < 8 else HF.ACT_SIGMOID) for i in (0, 2, 4, 6, 8)])
        for i in (0, 2, 4, 6):
            x = self.net[i](x, act=HF.ACT_RELU)
        return self.net[8](x, act=HF.ACT_SIGMOID)


class PhraseZDiscriminator(_ZDisc):
    pass


class BarZDiscriminator(_ZDisc):
    pass
