import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/musicgeneration_vae-torch_amd')
import torch
import __graft_entry__ as g; g.build()
from oracle import restate as R, weights as W
import graph.encodingBlock as EB
dev='cuda'
torch.manual_seed(1)
gsd = W.make_state_dict(W.manifest_generator(), 0, 'd4')
pre = "encoder.time_pitch."
sub = {k[len(pre):]: v for k,v in gsd.items() if k.startswith(pre)}
x = (torch.rand(3,1,96,60)<0.05).float()
for dt in (torch.float64, torch.float32):
    osd = {k: v.clone().to(dt).requires_grad_(True) for k,v in sub.items()}
    xr = x.to(dt).requires_grad_(True)
    torch.manual_seed(5); yr = R.enc_time_pitch(osd, "", xr); dy = torch.randn(yr.shape, dtype=torch.float64); yr.backward(dy.to(dt))
    print(dt, "dbeta[:6]", osd["bn.bias"].grad[:6].tolist())
    if dt == torch.float64: ref = osd["bn.bias"].grad.clone(); refg = osd["bn.weight"].grad.clone()
    else: print("torch fp32 vs fp64 dbeta rel", ((osd["bn.bias"].grad.double()-ref).abs().max()/ref.abs().max()).item())
m = EB.TimePitchModule(); m.load_state_dict(sub); m = m.to(dev)
xd = x.to(dev).requires_grad_(True)
y = m(xd); y.backward(dy.float().to(dev))
print("hip dbeta[:6]", m.bn.bias.grad[:6].tolist())
print("hip vs fp64 dbeta rel", ((m.bn.bias.grad.double().cpu()-ref).abs().max()/ref.abs().max()).item(), "dgamma rel", ((m.bn.weight.grad.double().cpu()-refg).abs().max()/refg.abs().max()).item())
print("y stats", y.min().item(), y.max().item())
