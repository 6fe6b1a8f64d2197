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
    gen.load_state_dict(sd, strict=False)
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
