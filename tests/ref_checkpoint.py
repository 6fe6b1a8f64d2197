"""Builds a checkpoint in the REFERENCE's layout without the reference: plain torch modules whose parameter / buffer
names and shapes come from tests/golden/manifest.json (written from the reference import by oracle/make_golden.py) plus
the 30 Refiner entries of graph/refiner.py:11-47 (with the reference's own -- unrunnable, defect D2 -- shapes), keyed like
nn.DataParallel (``module.`` prefixes) and optimised by torch.optim.Adam, i.e. exactly what
agent/barGen2.py:168-177 hands to torch.save.  Used by tests/test_checkpoint_gpu.py; run as a script to write one:

    python tests/ref_checkpoint.py /tmp/checkpoint.pth.tar
"""
import json
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))

# graph/refiner.py:11-47 state_dict (names, shapes, buffer?) -- layer2.0.weight is [8, 1, 4, 4] there (defect D2)
REFINER = [("layer1.0.weight", (2, 1, 4, 4)), ("layer1.0.bias", (2,)), ("layer1.1", 2),
           ("layer2.0.weight", (8, 1, 4, 4)), ("layer2.0.bias", (8,)), ("layer2.1", 8),
           ("layer3.0.weight", (1024, 2880)), ("layer3.0.bias", (1024,)),
           ("layer4.0.weight", (2880, 1024)), ("layer4.0.bias", (2880,)),
           ("layer5.0.weight", (8, 2, 4, 4)), ("layer5.1", 2),
           ("layer6.0.weight", (2, 1, 4, 4)), ("layer6.1", 1)]


def _nest(root, name, tensor, buffer=False):
    parts = name.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor))


def bag(entries, rng, refiner=False):
    """an nn.Module tree holding ``entries`` = [(dotted name, shape)] as parameters (BatchNorm statistics as buffers)"""
    root = nn.Module()
    for name, shape in entries:
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            _nest(root, name, torch.tensor(3, dtype=torch.long), True)
        elif name.endswith("running_mean") or name.endswith("running_var"):
            _nest(root, name, torch.from_numpy(rng.random(shape).astype(np.float32) + 0.5), True)
        else:
            _nest(root, name, torch.from_numpy((rng.standard_normal(shape) * 0.05).astype(np.float32)))
    if refiner:
        for name, spec in REFINER:
            if isinstance(spec, int):        # BatchNorm2d
                for k in ("weight", "bias"):
                    _nest(root, "refiner.%s.%s" % (name, k), torch.from_numpy(rng.standard_normal(spec).astype(np.float32)))
                _nest(root, "refiner.%s.running_mean" % name, torch.zeros(spec), True)
                _nest(root, "refiner.%s.running_var" % name, torch.ones(spec), True)
                _nest(root, "refiner.%s.num_batches_tracked" % name, torch.tensor(0, dtype=torch.long), True)
            else:
                _nest(root, "refiner." + name, torch.from_numpy((rng.standard_normal(spec) * 0.05).astype(np.float32)))
    return root


def manifest():
    return json.load(open(os.path.join(HERE, "golden", "manifest.json")))


def generator_entries(man):
    return [["encoder." + n, s] for n, s in man["encoder"]] + [["decoder." + n, s] for n, s in man["decoder"]] + \
           [["phrase_encoder." + n, s] for n, s in man["phrase_encoder"]]


class DataParallelKeys(nn.Module):
    """what nn.DataParallel does to a state_dict -- every key gains the ``module.`` prefix -- without its side effect on
    a one-GPU box (it moves the wrapped module to that GPU; the fixture is built on the host)"""

    def __init__(self, module):
        super().__init__()
        self.module = module


def trained(module, rng, lr, steps=2):
    """DataParallel-style wrapper + torch.optim.Adam after ``steps`` steps on random gradients"""
    dp = DataParallelKeys(module)
    opt = torch.optim.Adam(dp.parameters(), lr=lr)
    for _ in range(steps):
        for p in dp.parameters():
            p.grad = torch.from_numpy((rng.standard_normal(tuple(p.shape)) * 0.01).astype(np.float32))
        opt.step()
    return dp, opt


def build(seed=0, lr=0.0016, with_refiner=True):
    """-> (checkpoint dict in agent/barGen2.py:168-177's layout, {name: (DataParallel module, Adam)})"""
    rng = np.random.default_rng(seed)
    man = manifest()
    nets = {"generator": trained(bag(generator_entries(man), rng, refiner=with_refiner), rng, lr),
            "z_discriminator_bar": trained(bag(man["z_discriminator_bar"], rng), rng, lr),
            "z_discriminator_phrase": trained(bag(man["z_discriminator_phrase"], rng), rng, lr)}
    ck = {"generator_state_dict": nets["generator"][0].state_dict(), "generator_optimizer": nets["generator"][1].state_dict(),
          "z_discriminator_bar_state_dict": nets["z_discriminator_bar"][0].state_dict(),
          "opt_Zdiscriminator_bar_optimizer": nets["z_discriminator_bar"][1].state_dict(),
          "z_discriminator_phrase_state_dict": nets["z_discriminator_phrase"][0].state_dict(),
          "opt_Zdiscriminator_phrase_optimizer": nets["z_discriminator_phrase"][1].state_dict()}
    return ck, nets


if __name__ == "__main__":
    ck, _ = build()
    torch.save(ck, sys.argv[1] if len(sys.argv) > 1 else "checkpoint.pth.tar")
    print(len(ck["generator_state_dict"]), "generator entries,", len(ck["generator_optimizer"]["state"]), "optimizer states")
