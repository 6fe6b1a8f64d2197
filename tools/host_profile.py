"""host-side view of one training step: which aten ops (copies, fills, adds ...) the autograd tape and the host layer
issue besides the library's own launches (torch.profiler, CPU activity only)"""
import os, sys, collections
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
for _ in range(3):
    step(*batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(*batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25))
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::contiguous", "aten::empty_like", "aten::zeros"):
        st = [f for f in (ev.stack or []) if "musicgeneration" in f or "hipops" in f or "graph/" in f]
        cnt[(ev.name, st[0] if st else "(autograd / no python frame)")] += 1
for (name, where), n in cnt.most_common(40):
    print("%4d  %-18s %s" % (n, name, where))
