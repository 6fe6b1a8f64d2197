"""Phrase encoder on HIP kernels (reference: graph/phrase_encoder.py:8-55)."""
from torch import nn

from graph.encoder import _ConvTrunk
from graph.weights_initializer import weights_init


class PhraseEncoder(_ConvTrunk):
    """[B,1,384,60] -> [B,1152]; AvgPool2d((12,2)) + Linear(1024,1152) WITHOUT bias"""
    pool_hw = (12, 2)
    linear_bias = False


class PhraseModel(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.phrase_encoder = PhraseEncoder(layers)
        self.apply(weights_init)

    def forward(self, phrase):
        return self.phrase_encoder(phrase)
