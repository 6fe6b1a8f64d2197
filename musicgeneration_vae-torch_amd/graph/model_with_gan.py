"""Generator of the GAN agents (reference: graph/model_with_gan.py:11-38): adds a third
encoder pass over the binarised output (no gradient flows through the threshold)."""
import torch
from torch import nn

from graph.decoder import Decoder
from graph.encoder import Encoder
from graph.phrase_encoder import PhraseModel
from graph.weights_initializer import weights_init


class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder([64, 128, 256, 512, 1024])
        self.decoder = Decoder([1024, 512, 256, 128, 64])
        self.phrase_encoder = PhraseModel([64, 128, 256, 512, 1024])
        self.apply(weights_init)

    def forward(self, note, pre_note, phrase, position, is_note=True):
        phrase_feature = self.phrase_encoder(phrase)
        pre_z = self.encoder(pre_note)
        if is_note:
            z = self.encoder(note)
            gen = self.decoder(z, pre_z, phrase_feature, position)
            fake = torch.gt(gen.detach(), 0.3).float()
            return gen, z, pre_z, phrase_feature, self.encoder(fake)
        gen = self.decoder(note, pre_z, phrase_feature, position)
        fake = torch.gt(gen.detach(), 0.3).float()
        return gen, self.encoder(fake)
