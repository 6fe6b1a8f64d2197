"""cProfile of the host side of training steps (where the ~11 ms of enqueue time per step go)"""
import os, sys, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "musicgeneration_vae-torch_amd"))
import torch
import __graft_entry__ as ge; ge.build()
from hipops import functional as HF
from hipops.train import PretrainStep
from graph.model import Model
from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
from graph.loss.bar_loss import Loss, DLoss
dev = "cuda"
HF.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "f32")
B = 64
torch.manual_seed(0)
gen, zb, zp = Model().to(dev), BarZDiscriminator().to(dev), PhraseZDiscriminator().to(dev)
step = PretrainStep(gen, zb, zp, Loss().to(dev), DLoss().to(dev), lr=0.002)
g = torch.Generator().manual_seed(1)
batch = [(torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev), (torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev),
         (torch.rand(B, 1, 384, 60, generator=g) < 0.05).float().to(dev), torch.randint(0, 332, (B,), generator=g).to(dev)]
for _ in range(5):
    step(*batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(*batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
