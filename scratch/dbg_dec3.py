import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/musicgeneration_vae-torch_amd')
import torch, numpy as np
import __graft_entry__ as g; g.build()
from oracle import restate as R, weights as W
import graph.decoder as DD
from hipops import functional as HF
dev='cuda'
B=3
def rel(a,b):
    a=a.detach().double().cpu(); b=b.detach().double().cpu(); return ((a-b).abs().max()/b.abs().max()).item()
torch.manual_seed(0)
gsd = W.make_state_dict(W.manifest_generator(), 0, 'wc')
_, dsd, _ = W.split_generator(gsd)
z, pz, pf = torch.randn(B,1152), torch.randn(B,1152), torch.randn(B,1152)
pos = torch.randint(0,332,(B,))
note = (torch.rand(B,1,96,60)<0.05).float()
osd = {k: v.clone().double().requires_grad_(True) for k,v in dsd.items()}
taps = {}
gen = R.decoder(osd, "", z.double(), pz.double(), pf.double(), pos, taps=taps)
for t in taps.values(): t.retain_grad()
lo = torch.nn.functional.binary_cross_entropy(gen, note.double()); lo.backward()
m = DD.Decoder([1024,512,256,128,64]); m.load_state_dict(dsd); m = m.to(dev).eval()
runs = []
for run in range(2):
    for p in m.parameters(): p.grad = None
    acts, grads = {}, {}
    def mk(name):
        def hook(mod, inp, out):
            acts[name] = out.detach().clone()
            out.register_hook(lambda g, name=name: grads.__setitem__(name, g.detach().clone()))
        return hook
    hs = [m.pitch.register_forward_hook(mk("pitch")), m.time.register_forward_hook(mk("time"))]
    for i, blk in enumerate(m.layers): hs.append(blk.register_forward_hook(mk("layers.%d" % i)))
    out = m(z.to(dev), pz.to(dev), pf.to(dev), pos.to(dev))
    l = HF.bce(out, note.to(dev)); l.backward(); torch.cuda.synchronize()
    for h in hs: h.remove()
    runs.append({n: p.grad.clone() for n,p in m.named_parameters() if p.grad is not None})
    if run == 0:
        for k in ["layers.3","layers.2","layers.1","layers.0","pitch","time"]:
            print("%-10s act rel %.2e   grad rel %.2e" % (k, rel(acts[k], taps[k]), rel(grads[k], taps[k].grad)))
a, b = runs
print("nondeterminism between two identical runs (max rel diff):")
for n in a:
    d = rel(a[n], b[n])
    if d > 1e-5: print("   ", n, d)
print("done")
