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
torch.manual_seed(0)
gsd = W.make_state_dict(W.manifest_generator(), 0, 'wc')
_, dsd, _ = W.split_generator(gsd)
z, pz, pf = torch.randn(B,1152), torch.randn(B,1152), torch.randn(B,1152)
pos = torch.randint(0,332,(B,))
d64 = {k: v.double() for k,v in dsd.items()}
taps = {}
with torch.no_grad():
    R.decoder(d64, "", z.double(), pz.double(), pf.double(), pos, taps=taps)
    x = taps["layers.2"]; p="layers.3."
    c1 = F.conv_transpose2d(x, d64[p+"deConv1.weight"], stride=2, padding=1)
    a = F.relu(F.instance_norm(c1, None, None, d64[p+"bn1.weight"], d64[p+"bn1.bias"], True, 0.01, 1e-5))
    c2 = F.conv_transpose2d(x, d64[p+"deConv2.weight"], d64[p+"deConv2.bias"], stride=2, padding=1, output_padding=1)
    b = F.relu(F.instance_norm(c2, None, None, d64[p+"bn2.weight"], d64[p+"bn2.bias"], True, 0.01, 1e-5))
    u64 = F.instance_norm(F.conv2d(torch.cat((a,b),1), d64[p+"conv.weight"]), None, None, d64[p+"bn3.weight"], d64[p+"bn3.bias"], True, 0.01, 1e-5)
m = DD.Decoder([1024,512,256,128,64]); m.load_state_dict(dsd); m = m.to(dev).eval()
cap = {}
m.layers[3].bn3.register_forward_hook(lambda mod, i, o: cap.__setitem__("u", o.detach().clone()))
with torch.no_grad():
    m(z.to(dev), pz.to(dev), pf.to(dev), pos.to(dev))
u32 = cap["u"].cpu()
N,C,H,Wd = u64.shape
f64 = u64.view(N,C,-1); f32 = u32.view(N,C,-1)
am64 = f64.argmax(2); am32 = f32.argmax(2)
print("u rel err", ((u32.double()-u64).abs().max()/u64.abs().max()).item())
diff = (am64 != am32).nonzero()
print("channel-max argmax differs for (n,c):", diff.tolist())
top2 = f64.topk(2, dim=2).values
gap = ((top2[...,0]-top2[...,1])/top2[...,0].abs())
print("smallest relative top-2 gaps (fp64):", torch.sort(gap.flatten()).values[:8].tolist())
for n,c in diff.tolist():
    i64, i32 = am64[n,c].item(), am32[n,c].item()
    print("  (n=%d,c=%d): fp64 argmax %d (%d,%d) val %.9f ; fp32 argmax %d (%d,%d) val32 %.9f / %.9f" % (n,c,i64,i64//Wd,i64%Wd,f64[n,c,i64], i32,i32//Wd,i32%Wd, f32[n,c,i32], f32[n,c,i64]))
# spatial: max over channels of v = u*cg
def cg_of(u, sd):
    w1, w2 = sd[p+"cbam.channel_attention.conv1.weight"], sd[p+"cbam.channel_attention.conv2.weight"]
    a = F.conv2d(F.relu(F.conv2d(F.adaptive_avg_pool2d(u,1), w1)), w2); mm = F.conv2d(F.relu(F.conv2d(F.adaptive_max_pool2d(u,1), w1)), w2)
    return torch.sigmoid(a+mm)
v64 = u64*cg_of(u64, d64); v32 = u32*cg_of(u32, {k: v.float() for k,v in d64.items()})
d2 = (v64.argmax(1) != v32.argmax(1)).nonzero()
print("spatial argmax over C differs at", d2.tolist()[:10], "count", len(d2))
t2 = v64.topk(2, dim=1).values; g2 = ((t2[:,0]-t2[:,1])/t2[:,0].abs())
print("smallest spatial top-2 gaps:", torch.sort(g2.flatten()).values[:8].tolist())
