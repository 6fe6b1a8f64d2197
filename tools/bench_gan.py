"""BASELINE.json configs[3]: one `train_gan` iteration of agent/barGen_with_gan.py:462-537 on
synthetic bars -- discriminator step (generator forward incl. the third encoder pass, bar discriminator x2 on the
2-bar pairs, feature discriminator x2, two backward, two Adam) followed by the generator step from N(0, 1.5^2)
noise -- timed through the agent's own method.  Prints one JSON line (bench.py is the driver's contract).
usage: python tools/bench_gan.py [per_gpu_batch=16] [f32|bf16] [steps=10]"""
import json, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "musicgeneration_vae-torch_amd"))
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
from config import Config
from agent.barGen_with_gan import BarGen
from metrics import AverageMeter

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
root = tempfile.mkdtemp(prefix="mgvae_gan_bench_")
d = os.path.join(root, "data", "dataset"); os.makedirs(d)
rng = np.random.default_rng(0)
np.savez(os.path.join(d, "bar_000.npz"), note=(rng.random((2, 1, 96, 60)) < 0.05).astype(np.float32),
         pre_note=(rng.random((2, 1, 96, 60)) < 0.05).astype(np.float32),
         pre_phrase=(rng.random((2, 1, 384, 60)) < 0.05).astype(np.float32), position=rng.integers(0, 332, size=(2,)))


class Cfg(Config):
    root_path = root
    batch_size = B
    seed = 1
    compute_dtype = dtype
    log_file = os.path.join(root, "train_epoch.log")


agent = BarGen(Cfg())
agent.epoch = agent.pretraining_step_size + 1
dev = agent.device
g = torch.Generator().manual_seed(1234)
note = (torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev)
pre = (torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev)
phrase = (torch.rand(B, 1, 384, 60, generator=g) < 0.05).float().to(dev)
pos = torch.randint(0, 332, (B,), generator=g).to(dev)
meters = {k: AverageMeter() for k in ("generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase")}
odd_it = 1 - (agent.epoch % 2)            # (epoch + it) odd -> the iteration that also trains the discriminators
for _ in range(3):
    agent.train_gan(note, pre, phrase, pos, meters, odd_it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    agent.train_gan(note, pre, phrase, pos, meters, odd_it)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": "barGen_with_gan train_gan iteration (D step + G step), per-GPU batch %d, %s, 1 GPU" % (B, dtype),
                  "ms_per_iteration": 1e3 * dt, "bars_per_s": B / dt}))
