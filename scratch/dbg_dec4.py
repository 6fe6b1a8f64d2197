import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/musicgeneration_vae-torch_amd')
import torch, numpy as np
import torch.nn.functional as F
import __graft_entry__ as g; g.build()
from oracle import restate as R, weights as W
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
d64 = {k: v.double() for k,v in dsd.items()}
taps = {}
with torch.no_grad():
    R.decoder(d64, "", z.double(), pz.double(), pf.double(), pos, taps=taps)
xin = taps["layers.2"].float()          # real input of layers.3
print("input stats: min %.3g max %.3g mean %.3g frac0 %.3f" % (xin.min(), xin.max(), xin.mean(), (xin==0).float().mean()))
p = "layers.3."
sd = {k[len(p):]: v for k,v in dsd.items() if k.startswith(p)}
# ---- oracle, step by step, fp64
o = {k: v.double().requires_grad_(True) for k,v in sd.items()}
xr = xin.double().requires_grad_(True)
T = {}
def keep(n, t): t.retain_grad(); T[n] = t; return t
c1 = keep("c1", F.conv_transpose2d(xr, o["deConv1.weight"], stride=2, padding=1))
a = keep("a", F.relu(F.instance_norm(c1, None, None, o["bn1.weight"], o["bn1.bias"], True, 0.01, 1e-5)))
c2 = keep("c2", F.conv_transpose2d(xr, o["deConv2.weight"], o["deConv2.bias"], stride=2, padding=1, output_padding=1))
b = keep("b", F.relu(F.instance_norm(c2, None, None, o["bn2.weight"], o["bn2.bias"], True, 0.01, 1e-5)))
cat = keep("cat", torch.cat((a,b),1))
c3 = keep("c3", F.conv2d(cat, o["conv.weight"]))
u = keep("u", F.instance_norm(c3, None, None, o["bn3.weight"], o["bn3.bias"], True, 0.01, 1e-5))
y = keep("y", F.relu(u + R.cbam(o, "cbam.", u)))
pre = F.conv2d(y, d64["fit2.weight"].requires_grad_(False)); gen = torch.sigmoid(pre)
lo = F.binary_cross_entropy(gen, note.double()); lo.backward()
# plane variances
v1 = c1.detach().var(dim=(2,3), unbiased=False); v2 = c2.detach().var(dim=(2,3), unbiased=False)
print("c1 plane var min %.3g median %.3g ; c2 plane var min %.3g median %.3g" % (v1.min(), v1.median(), v2.min(), v2.median()))
# ---- HIP, step by step
P = {k: torch.nn.Parameter(v.to(dev)) for k,v in sd.items()}
xd = xin.to(dev).requires_grad_(True)
G = {}
def hk(n, t): t.register_hook(lambda g, n=n: G.__setitem__(n, g.detach().clone())); return t
N_, _, h, w = xd.shape
catd = torch.empty((N_, 128, 2*h, 2*w), device=dev)
hc1 = hk("c1", HF.conv_transpose2d(xd, P["deConv1.weight"], None, (2,2), (1,1), (0,0)))
ha = hk("a", HF.instance_norm(hc1, P["bn1.weight"], P["bn1.bias"], 1e-5, 1, 0.01, out=catd[:, :64]))
hc2 = hk("c2", HF.conv_transpose2d(xd, P["deConv2.weight"], P["deConv2.bias"], (2,2), (1,1), (1,1)))
hb = hk("b", HF.instance_norm(hc2, P["bn2.weight"], P["bn2.bias"], 1e-5, 1, 0.01, out=catd[:, 64:]))
hcat = hk("cat", HF.join(catd, ha, hb))
hc3 = hk("c3", HF.conv2d(hcat, P["conv.weight"]))
hu = hk("u", HF.instance_norm(hc3, P["bn3.weight"], P["bn3.bias"]))
hy = hk("y", HF.cbam(hu, P["cbam.channel_attention.conv1.weight"], P["cbam.channel_attention.conv2.weight"], P["cbam.spatial_attention.conv.weight"], 1, None, 1, 0.01))
fit2w = dsd["fit2.weight"].to(dev)
hgen = HF.conv2d(hy, fit2w, None, (1,1),(0,0), 3, 0.01)
l = HF.bce(hgen, note.to(dev)); l.backward(); torch.cuda.synchronize()
H = {"c1":hc1,"a":ha,"c2":hc2,"b":hb,"cat":hcat,"c3":hc3,"u":hu,"y":hy}
for n in ["y","u","c3","cat","a","b","c1","c2"]:
    print("%-4s fwd rel %.2e   grad rel %.2e" % (n, rel(H[n], T[n]), rel(G[n], T[n].grad)))
print("dx rel %.2e" % rel(xd.grad, xr.grad))
# isolate each backward op with ORACLE upstream gradient
def iso(name, fn):
    print("   isolated %-28s %.2e" % (name, fn()))
def t_in1():
    x_ = hc1.detach().requires_grad_(True); out = HF.instance_norm(x_, P["bn1.weight"], P["bn1.bias"], 1e-5, 1, 0.01); out.backward(T["a"].grad.float().to(dev)); return rel(x_.grad, T["c1"].grad)
def t_in2():
    x_ = hc2.detach().requires_grad_(True); out = HF.instance_norm(x_, P["bn2.weight"], P["bn2.bias"], 1e-5, 1, 0.01); out.backward(T["b"].grad.float().to(dev)); return rel(x_.grad, T["c2"].grad)
def t_ct1():
    x_ = xin.to(dev).requires_grad_(True); out = HF.conv_transpose2d(x_, P["deConv1.weight"], None, (2,2),(1,1),(0,0)); out.backward(T["c1"].grad.float().to(dev))
    xr2 = xin.double().requires_grad_(True); F.conv_transpose2d(xr2, sd["deConv1.weight"].double(), stride=2, padding=1).backward(T["c1"].grad); return rel(x_.grad, xr2.grad)
def t_ct2():
    x_ = xin.to(dev).requires_grad_(True); out = HF.conv_transpose2d(x_, P["deConv2.weight"], P["deConv2.bias"], (2,2),(1,1),(1,1)); out.backward(T["c2"].grad.float().to(dev))
    xr2 = xin.double().requires_grad_(True); F.conv_transpose2d(xr2, sd["deConv2.weight"].double(), sd["deConv2.bias"].double(), stride=2, padding=1, output_padding=1).backward(T["c2"].grad); return rel(x_.grad, xr2.grad)
iso("IN1+relu bwd (oracle dy)", t_in1); iso("IN2+relu bwd (oracle dy)", t_in2); iso("convT1 bwd_data (oracle dy)", t_ct1); iso("convT2 bwd_data (oracle dy)", t_ct2)
