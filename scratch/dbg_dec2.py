import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/musicgeneration_vae-torch_amd')
import torch, numpy as np
import __graft_entry__ as g; g.build()
from oracle import restate as R, weights as W
import graph.decoder as DD
from hipops import functional as HF
dev='cuda'
B=3
torch.manual_seed(0)
gsd = W.make_state_dict(W.manifest_generator(), 0, 'wc')
_, dsd, _ = W.split_generator(gsd)
# last block alone, but with the REAL tail (fit2 + sigmoid + BCE)
x = torch.randn(B,128,48,30).relu_()
note = (torch.rand(B,1,96,60)<0.05).float()
def rel(a,b):
    a=a.detach().double().cpu(); b=b.detach().double().cpu(); return ((a-b).abs().max()/b.abs().max()).item()
for scale in (1.0,):
    osd = {k: v.clone().double().requires_grad_(True) for k,v in dsd.items()}
    xr = x.double().requires_grad_(True)
    o = R.deconv_module(osd, "layers.3.", xr); o.retain_grad()
    pre = torch.nn.functional.conv2d(o, osd["fit2.weight"]); pre.retain_grad()
    gen = torch.sigmoid(pre); gen.retain_grad()
    lo = torch.nn.functional.binary_cross_entropy(gen, note.double())
    lo.backward()
    blk = DD.DeConvModule(128,64); blk.load_state_dict({k[len("layers.3."):]:v for k,v in dsd.items() if k.startswith("layers.3.")}); blk=blk.to(dev)
    from graph.layers import Conv2d
    fit2 = Conv2d(64,1,1,bias=False); fit2.load_state_dict({"weight": dsd["fit2.weight"]}); fit2=fit2.to(dev)
    xd = x.to(dev).requires_grad_(True)
    grads = {}
    od = blk(xd); od.register_hook(lambda g: grads.__setitem__("o", g.clone()))
    gd = fit2(od, act=HF.ACT_SIGMOID); gd.register_hook(lambda g: grads.__setitem__("gen", g.clone()))
    l = HF.bce(gd, note.to(dev)); l.backward()
    print("fwd o", rel(od,o), "gen", rel(gd,gen), "loss", l.item(), lo.item())
    print("dgen", rel(grads["gen"], gen.grad), "do", rel(grads["o"], o.grad), "dx", rel(xd.grad, xr.grad))
    for n,p in list(blk.named_parameters()):
        print("  ", n, rel(p.grad, osd["layers.3."+n].grad))
    print("   fit2", rel(fit2.weight.grad, osd["fit2.weight"].grad))
    # isolate: feed the oracle's do into the block backward
    xd2 = x.to(dev).requires_grad_(True)
    for p in blk.parameters(): p.grad=None
    od2 = blk(xd2); od2.backward(o.grad.float().to(dev))
    print("block alone with oracle do: dx", rel(xd2.grad, xr.grad))
    for n,p in list(blk.named_parameters()):
        print("  ", n, rel(p.grad, osd["layers.3."+n].grad))
    # random dy of the same tiny magnitude
    dy = torch.randn_like(o)*o.grad.abs().mean()
    for v in osd.values(): v.grad=None
    xr.grad=None
    o2 = R.deconv_module(osd, "layers.3.", xr); o2.backward(dy)
    xd3 = x.to(dev).requires_grad_(True)
    for p in blk.parameters(): p.grad=None
    od3 = blk(xd3); od3.backward(dy.float().to(dev))
    print("block alone with tiny random dy: dx", rel(xd3.grad, xr.grad))
    for n,p in list(blk.named_parameters()):
        print("  ", n, rel(p.grad, osd["layers.3."+n].grad))
