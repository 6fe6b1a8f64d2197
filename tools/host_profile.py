"""cProfile of the host side of the pre-training step at a small batch (where the host bounds the step): top functions by
cumulative and by own time.  usage: python tools/host_profile.py [batch] [dtype]"""
import sys, os, cProfile, pstats, io
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
HF.set_compute_dtype(sys.argv[2] if len(sys.argv) > 2 else "bf16")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen, zb, zp = Model().to(dev).train(), BarZDiscriminator().to(dev), PhraseZDiscriminator().to(dev)
for d in (zb, zp):
    for p in d.parameters():
        p.requires_grad = False
HF.manual_seed(1234, 0)
step = PretrainStep(gen, zb, zp, Loss().to(dev), DLoss(), lr=0.002)
batch = bench.synth_batch(B, 1234, dev)
for _ in range(5): step(*batch)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("20 steps: host %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) * 50, (t2 - t0) * 50))
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step(*batch)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in sio.getvalue().splitlines()[4:44]))
