"""Feature discriminator (reference: graph/bar_discriminator_with_feature.py:6-25):
Linear 1152->512->1 without biases and WITHOUT an activation in between, sigmoid head."""
from torch import nn

from hipops import blocks as HB
from hipops import functional as HF
from graph.layers import Linear
from graph.weights_initializer import weights_init


class BarFeatureDiscriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.linear1 = Linear(1152, 512, bias=False)
        self.linear2 = Linear(512, 1, bias=False)
        self.apply(weights_init)

    def forward(self, x):
        x = x.reshape(-1, 1152)
        if HB.mlp_usable(x):
            return HB.mlp(x, [(self.linear1.weight, None, HF.ACT_NONE), (self.linear2.weight, None, HF.ACT_SIGMOID)])
        return self.linear2(self.linear1(x), act=HF.ACT_SIGMOID)
