"""Generator of the GAN agents (reference: graph/model_with_gan.py:11-38): adds a third
encoder pass over the binarised output (no gradient flows through the threshold)."""
import torch
from torch import nn

from graph.decoder import Decoder
from graph.encoder import Encoder
from graph.phrase_encoder import PhraseModel
from graph.weights_initializer import weights_init
from graph import model as _plain


class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder([64, 128, 256, 512, 1024])
        self.decoder = Decoder([1024, 512, 256, 128, 64])
        self.phrase_encoder = PhraseModel([64, 128, 256, 512, 1024])
        self.apply(weights_init)

    def encode_pair(self, note, pre_note):
        """both bar-encoder passes of graph/model_with_gan.py:22,24 as one pass over the 2B stacked bars
        (exact: no cross-sample op in the encoder)"""
        if getattr(self.encoder, "variational", False) or note.shape != pre_note.shape:
            return self.encoder(note), self.encoder(pre_note)
        zz = self.encoder(torch.cat([note, pre_note], 0))
        b = note.shape[0]
        return zz[:b], zz[b:]

    # phrase trunk on a side stream beside the bar trunk (graph/model.py::Model.encode_phrase)
    encode_phrase = _plain.Model.encode_phrase
    # ... at NORMAL priority here: beside the side-stream weight gradients of the three extra networks of a GAN iteration a
    # high-priority phrase stream costs 48 -> 80 ms per iteration at 64 bars (tools/bench_gan.py), while the plain
    # pre-training step gains 1.3 % from it
    phrase_stream_priority = 0
    join_phrase = _plain.Model.join_phrase

    def forward(self, note, pre_note, phrase, position, is_note=True):
        phrase_feature = self.encode_phrase(phrase)
        if is_note:
            z, pre_z = self.encode_pair(note, pre_note)
            self.join_phrase()
            gen = self.decoder(z, pre_z, phrase_feature, position)
            fake = torch.gt(gen.detach(), 0.3).float()
            return gen, z, pre_z, phrase_feature, self.encoder(fake)
        pre_z = self.encoder(pre_note)
        self.join_phrase()
        gen = self.decoder(note, pre_z, phrase_feature, position)
        fake = torch.gt(gen.detach(), 0.3).float()
        return gen, self.encoder(fake)
