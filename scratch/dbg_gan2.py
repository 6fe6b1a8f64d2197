import sys, os, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/musicgeneration_vae-torch_amd'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import __graft_entry__ as g; g.build()
from test_agent_gpu import _make_dataset
from config import Config
root = tempfile.mkdtemp(); _make_dataset(root, 4, 2)
class Cfg(Config):
    root_path = root; batch_size = 2; epoch = 3; pretraining_step_size = 1; seed = 11; log_file = os.path.join(root, "l.log")
from agent.barGen_with_gan2 import BarGen
a = BarGen(Cfg())
w0 = a.opt_discriminator.flat.clone()
a.epoch = 2
batch = a.to_device(*next(iter(a.dataloader)))
from metrics import AverageMeter
meters = {k: AverageMeter() for k in ("generator", "discriminator", "discriminator_feature", "z_bar", "z_phrase")}
a.train_discriminator(*batch, meters)
torch.cuda.synchronize()
print("grad abs max", a.opt_discriminator.grad.abs().max().item(), "delta", (a.opt_discriminator.flat - w0).abs().max().item())
print("loss", float(meters["discriminator"].val), "requires_grad", [p.requires_grad for p in list(a.discriminator.parameters())[:3]])
print("step_count", a.opt_discriminator.step_count)
