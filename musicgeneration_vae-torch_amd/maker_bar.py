"""Autoregressive bar-by-bar sampling (reference: maker_bar.py:19-55): load a checkpoint, generate
``music_length`` phrases of 4 bars each -- every bar decoded from N(0,1) noise conditioned on the
previous binarised bar, the previous 4-bar phrase and the phrase position -- and write the piano
roll padded from 60 to 128 pitches with (27, 41) like the reference.  The phrase encoder result is
computed once per phrase (the reference recomputes it for each of the 4 bars).  MIDI export needs
``pypianoroll`` (not a dependency here): without it the roll is saved as ``.npy``."""
import os
import sys

import numpy as np
import torch

from config import Config
from graph.model import Model
from hipops import functional as HF


def load_generator(config, device, use_refiner=False):
    gen = Model(use_refiner=use_refiner).to(device).eval()
    path = os.path.join(config.root_path, config.checkpoint_dir, config.checkpoint_file)
    ck = torch.load(path, map_location=device, weights_only=False)
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ck["generator_state_dict"].items()}
    # a checkpoint in another layout must not silently leave the generator at its random initialisation: only the
    # refiner's entries may be absent (checkpoints of this build's default, refiner-less generator) or surplus
    # (reference checkpoints loaded into it) -- the rule of agent/base.py::Net.load_state_dict
    res = gen.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if not k.startswith("refiner.")]
    unexpected = [k for k in res.unexpected_keys if not k.startswith("refiner.")]
    if missing or unexpected:
        raise RuntimeError("checkpoint %s does not match the generator: missing %s, unexpected %s"
                           % (path, missing[:5], unexpected[:5]))
    return gen


@torch.no_grad()
def sample(gen, music_length=10, songs=1, device="cuda"):
    """returns [songs, music_length * 384, 60] binary rolls; songs are independent batch entries"""
    pre_phrase = torch.zeros(songs, 1, 384, 60, device=device)
    pre_bar = torch.zeros(songs, 1, 96, 60, device=device)
    phrase_idx = [330] + list(range(music_length - 2, -1, -1))
    out = []
    for idx in range(music_length):
        pos = torch.full((songs,), phrase_idx[idx], device=device, dtype=torch.long)
        phrase_feature = gen.phrase_encoder(pre_phrase)          # reused by the 4 bars of this phrase
        bars = []
        for _ in range(4):
            pre_z = gen.encoder(pre_bar)
            bar = gen.decoder(HF.randn((songs, 1152), 1.0, device), pre_z, phrase_feature, pos)
            if getattr(gen, "use_refiner", False):
                bar = gen.refiner(bar)
            pre_bar = torch.gt(bar, 0.3).float()
            bars.append(pre_bar)
        pre_phrase = torch.cat(bars, dim=2)
        out.append(pre_phrase.reshape(songs, 384, 60))
    return torch.cat(out, dim=1)


class GraphSampler:
    """The same loop with the two device programs -- "encode the previous phrase" (once per phrase) and "decode one
    bar" (encoder of the previous bar, decoder, optional refiner, threshold) -- captured ONCE as HIP graphs over
    static buffers and replayed: a sampling call at batch 32 is ~200 short launches, i.e. bound by the host's launch
    rate, not by the GPU.  Prior noise is drawn outside the graph (its Philox offset advances per call).
    Capture needs every conv geometry to have been seen (gather tables, autotuner), hence the eager warm-up."""

    def __init__(self, gen, songs=1, device="cuda"):
        self.gen, self.songs, self.device = gen, songs, device
        self.pre_phrase = torch.zeros(songs, 1, 384, 60, device=device)
        self.pre_bar = torch.zeros(songs, 1, 96, 60, device=device)
        self.z = torch.zeros(songs, 1152, device=device)
        self.pos = torch.zeros(songs, device=device, dtype=torch.long)
        self.pf = torch.zeros(songs, 1152, device=device)
        with torch.no_grad():
            for _ in range(2):                       # eager warm-up on the capture inputs' shapes
                self._phrase_program()
                self._bar_program()
            torch.cuda.synchronize()
            self.g_phrase, self.g_bar = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_phrase):
                self._phrase_program()
            with torch.cuda.graph(self.g_bar, pool=self.g_phrase.pool()):
                self._bar_program()
        self.pre_bar.zero_()

    def _phrase_program(self):
        self.pf.copy_(self.gen.phrase_encoder(self.pre_phrase))

    def _bar_program(self):
        gen = self.gen
        bar = gen.decoder(self.z, gen.encoder(self.pre_bar), self.pf, self.pos)
        if getattr(gen, "use_refiner", False):
            bar = gen.refiner(bar)
        self.pre_bar.copy_(torch.gt(bar, 0.3).float())

    @torch.no_grad()
    def sample(self, music_length=10):
        """[songs, music_length * 384, 60] binary rolls, same schedule as ``sample``"""
        S = self.songs
        self.pre_phrase.zero_()
        self.pre_bar.zero_()
        phrase_idx = [330] + list(range(music_length - 2, -1, -1))
        out = torch.empty(S, music_length * 384, 60, device=self.device)
        for idx in range(music_length):
            self.pos.fill_(phrase_idx[idx])
            self.g_phrase.replay()
            for b in range(4):
                HF.randn((S, 1152), 1.0, self.device, out=self.z)
                self.g_bar.replay()
                out[:, idx * 384 + b * 96: idx * 384 + (b + 1) * 96].copy_(self.pre_bar.view(S, 96, 60))
            self.pre_phrase.copy_(out[:, idx * 384:(idx + 1) * 384].reshape(S, 1, 384, 60))
        return out


def main():
    config = Config()
    device = torch.device("cuda", 0)
    gen = load_generator(config, device)
    roll = sample(gen, music_length=int(sys.argv[1]) if len(sys.argv) > 1 else 10, device=device)[0].cpu().numpy()
    roll128 = np.pad(roll, ((0, 0), (27, 41)), mode="constant", constant_values=0.0)
    try:
        from pypianoroll import Multitrack, Track
        track = Track(pianoroll=roll128 > 0, program=0, is_drum=False, name="generated")
        Multitrack(tracks=[track], tempo=100.0, beat_resolution=24).write("generated.mid")
        print("wrote generated.mid")
    except ImportError:
        np.save("generated.npy", roll128.astype(np.uint8))
        print("pypianoroll not installed: wrote generated.npy", roll128.shape)


if __name__ == "__main__":
    main()
