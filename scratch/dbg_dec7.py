import sys, os, types
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
# ---- oracle with instrumented last block
osd = {k: v.clone().double().requires_grad_(True) for k,v in dsd.items()}
T = {}
def keep(n, t): t.retain_grad(); T[n] = t; return t
orig = R.deconv_module
def inst(sd, p, x):
    if p != "layers.3.": return orig(sd, p, x)
    keep("x", x)
    c1 = keep("c1", F.conv_transpose2d(x, sd[p+"deConv1.weight"], stride=2, padding=1))
    a = keep("a", F.relu(F.instance_norm(c1, None, None, sd[p+"bn1.weight"], sd[p+"bn1.bias"], True, 0.01, 1e-5)))
    c2 = keep("c2", F.conv_transpose2d(x, sd[p+"deConv2.weight"], sd[p+"deConv2.bias"], stride=2, padding=1, output_padding=1))
    b = keep("b", F.relu(F.instance_norm(c2, None, None, sd[p+"bn2.weight"], sd[p+"bn2.bias"], True, 0.01, 1e-5)))
    cat = keep("cat", torch.cat((a,b),1))
    c3 = keep("c3", F.conv2d(cat, sd[p+"conv.weight"]))
    u = keep("u", F.instance_norm(c3, None, None, sd[p+"bn3.weight"], sd[p+"bn3.bias"], True, 0.01, 1e-5))
    return keep("y", F.relu(u + R.cbam(sd, p+"cbam.", u)))
R.deconv_module = inst
gen = R.decoder(osd, "", z.double(), pz.double(), pf.double(), pos)
lo = F.binary_cross_entropy(gen, note.double()); lo.backward()
# ---- HIP with instrumented last block
m = DD.Decoder([1024,512,256,128,64]); m.load_state_dict(dsd); m = m.to(dev).eval()
G = {}; Hh = {}
def hk(n, t): Hh[n]=t; t.register_hook(lambda g, n=n: G.__setitem__(n, g.detach().clone())); return t
def fwd(self, x, out=None):
    hk("x", x)
    co = self.out_channel
    n, _, h, w = x.shape
    cat = torch.empty((n, 2 * co, 2 * h, 2 * w), device=x.device, dtype=torch.float32)
    c1 = hk("c1", self.deConv1(x)); a = hk("a", self.bn1(c1, act=HF.ACT_RELU, out=cat[:, :co]))
    c2 = hk("c2", self.deConv2(x)); b = hk("b", self.bn2(c2, act=HF.ACT_RELU, out=cat[:, co:]))
    j = hk("cat", HF.join(cat, a, b))
    c3 = hk("c3", self.conv(j)); u = hk("u", self.bn3(c3))
    return hk("y", self.cbam.fused(u, 1, act=HF.ACT_RELU, out=out))
m.layers[3].forward = types.MethodType(fwd, m.layers[3])
out = m(z.to(dev), pz.to(dev), pf.to(dev), pos.to(dev))
l = HF.bce(out, note.to(dev)); l.backward(); torch.cuda.synchronize()
for n in ["y","u","c3","cat","a","b","c1","c2","x"]:
    print("%-4s fwd rel %.2e   grad rel %.2e" % (n, rel(Hh[n], T[n]), rel(G[n], T[n].grad)))
