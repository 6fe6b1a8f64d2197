import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/musicgeneration_vae-torch_amd')
import torch, numpy as np
import torch.nn.functional as F
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
lo = F.binary_cross_entropy(gen, note.double()); lo.backward()
m = DD.Decoder([1024,512,256,128,64]); m.load_state_dict(dsd); m = m.to(dev).eval()
grads = {}
def mk(name):
    def hook(mod, inp, out):
        out.register_hook(lambda g, name=name: grads.__setitem__(name, g.detach().clone()))
    return hook
for i, blk in enumerate(m.layers): blk.register_forward_hook(mk("layers.%d" % i))
out = m(z.to(dev), pz.to(dev), pf.to(dev), pos.to(dev))
l = HF.bce(out, note.to(dev)); l.backward(); torch.cuda.synchronize()
gh = grads["layers.2"].double().cpu(); gr = taps["layers.2"].grad
err = (gh-gr).abs(); scale = gr.abs().max()
print("layers.2 grad rel %.3e; shape %s" % ((err.max()/scale).item(), tuple(gr.shape)))
bad = err > 1e-5*scale
print("n bad", int(bad.sum()), "of", bad.numel())
print("bad per sample", bad.sum(dim=(1,2,3)).tolist())
print("bad per (h%2,w%2):", [[int(bad[:,:,i::2,j::2].sum()) for j in range(2)] for i in range(2)])
cb = bad.sum(dim=(0,2,3)); print("bad channels (count>0):", int((cb>0).sum()), "top", torch.topk(cb,5))
hb = bad.sum(dim=(0,1,3)); print("bad rows:", hb.tolist())
wb = bad.sum(dim=(0,1,2)); print("bad cols:", wb.tolist())
# is the input of layers.3 an exact-zero there?
x_in = taps["layers.2"].detach()
print("fraction of bad where input==0: %.3f ; overall frac zero %.3f" % (((x_in==0)&bad).sum().item()/max(1,bad.sum().item()), (x_in==0).float().mean().item()))
print("ratio hip/ref at worst:", (gh.flatten()[err.argmax()]/gr.flatten()[err.argmax()]).item(), gh.flatten()[err.argmax()].item(), gr.flatten()[err.argmax()].item())
