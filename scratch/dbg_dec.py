import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/musicgeneration_vae-torch_amd')
import torch, numpy as np
import __graft_entry__ as g; g.build()
from oracle import restate as R, weights as W
from graph.decoder import Decoder
dev='cuda'
for B in (3, 4):
    torch.manual_seed(0)
    gsd = W.make_state_dict(W.manifest_generator(), 0, 'wc')
    _, dsd, _ = W.split_generator(gsd)
    z, pz, pf = torch.randn(B,1152), torch.randn(B,1152), torch.randn(B,1152)
    pos = torch.randint(0,332,(B,))
    note = (torch.rand(B,1,96,60)<0.05).float()
    osd = {k: v.clone().double().requires_grad_(True) for k,v in dsd.items()}
    gen = R.decoder(osd, "", z.double(), pz.double(), pf.double(), pos)
    lo = torch.nn.functional.binary_cross_entropy(gen, note.double())
    names = [k for k in osd]
    og = torch.autograd.grad(lo, [osd[n] for n in names], allow_unused=True)
    m = Decoder([1024,512,256,128,64]); m.load_state_dict(dsd); m = m.to(dev).eval()
    zd = z.to(dev).requires_grad_(True)
    out = m(zd, pz.to(dev), pf.to(dev), pos.to(dev))
    from hipops import functional as HF
    l = HF.bce(out, note.to(dev)); l.backward()
    print("B", B, "loss", l.item(), lo.item(), "gen rel", ((out.detach().cpu().double()-gen.detach()).abs().max()/gen.abs().max()).item())
    params = dict(m.named_parameters())
    for n, gg in zip(names, og):
        if gg is None: continue
        a = params[n].grad.detach().double().cpu()
        err = (a-gg).abs()
        r = (err.max()/gg.abs().max().clamp_min(1e-30)).item()
        if r > 1e-4 and gg.abs().max() > 1e-7:
            idx = np.unravel_index(err.argmax().item(), tuple(gg.shape))
            nbad = int((err > 1e-4*gg.abs().max()).sum())
            print("  %-50s rel %.2e  argmax idx %s  nbad %d / %d" % (n, r, idx, nbad, gg.numel()))
