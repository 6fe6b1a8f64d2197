"""Which Python call sites issue device-to-device copies / ATen kernels in one pre-training step?  (torch.profiler with stacks)"""
import sys, os, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import bench
import __graft_entry__ as ge
ge.build()
from hipops import functional as HF
from hipops.train import PretrainStep
from graph.model import Model
from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
from graph.loss.bar_loss import Loss, DLoss
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen, zb, zp = Model().to(dev).train(), BarZDiscriminator().to(dev), PhraseZDiscriminator().to(dev)
for d in (zb, zp):
    for p in d.parameters():
        p.requires_grad = False
HF.manual_seed(1234, 0)
step = PretrainStep(gen, zb, zp, Loss().to(dev), DLoss(), lr=0.002)
batch = bench.synth_batch(int(os.environ.get("BENCH_B", "64")), 1234, dev)
for _ in range(3): step(*batch)
torch.cuda.synchronize()
import traceback
sites = collections.Counter()
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        st = [fr for fr in traceback.extract_stack(limit=8)[:-1] if "musicgeneration" in fr.filename or "bench.py" in fr.filename]
        key = "%s:%d" % (os.path.basename(st[-1].filename), st[-1].lineno) if st else "?"
        sites[(name, key)] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for n in ("copy_", "clone", "contiguous", "to", "zero_", "fill_", "add_", "mul_", "float", "bfloat16"):
    wrap(torch.Tensor, n)
for n in ("cat", "zeros", "zeros_like", "ones", "full", "add", "mul", "stack", "sum"):
    wrap(torch, n)
step(*batch)
torch.cuda.synchronize()
for (n, k), c in sites.most_common(50):
    print("%4d %-12s %s" % (c, n, k))
