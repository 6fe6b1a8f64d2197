"""every (tile, split) variant of the channels-last fp32 kernels against torch fp64 on the model's geometries at the batch
sizes of the 4-bar tests (N = 4, 8): MGVAE_NHWC_FORCE pins the variant"""
import sys, os, ctypes, itertools
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch, torch.nn.functional as F
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
def vp(t): return ctypes.c_void_p(t.data_ptr())
def cl(t): return t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
GEOMS = [(8, 512, 6, 4, 512, 3, 1, 1), (8, 256, 12, 8, 256, 3, 1, 1), (8, 512, 6, 4, 1024, 3, 2, 1), (4, 128, 48, 30, 256, 3, 2, 1),
         (4, 128, 48, 30, 256, 4, 2, 1), (4, 512, 12, 7, 1024, 4, 2, 1), (8, 64, 48, 30, 64, 3, 1, 1), (4, 2048, 6, 3, 1024, 1, 1, 0)]
bad = 0
for (N, Cx, H, W, Cy, k, st, p) in GEOMS:
    OH, OW = (H + 2*p - k)//st + 1, (W + 2*p - k)//st + 1
    torch.manual_seed(1)
    x = torch.randn(N, Cx, H, W).relu_(); w = torch.randn(Cy, Cx, k, k) * 0.1; dy = torch.randn(N, Cy, OH, OW)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, p); yr.backward(dy.double())
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 0, 0.0)
    xd, wd, dyd = cl(x), cl(w), cl(dy)
    for tile, split in itertools.product(range(4), (1, 2, 3, 8)):
        os.environ["MGVAE_NHWC_FORCE"] = "%d,%d" % (tile, split)
        yd = cl(torch.zeros(N, Cy, OH, OW)); dx = cl(torch.zeros(N, Cx, H, W)); dw = cl(torch.zeros(Cy, Cx, k, k))
        rc = [L.mgvae_conv2d_nhwc_fwd(ctypes.byref(d), vp(xd), vp(wd), None, vp(yd), None, s),
              L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d), vp(dyd), vp(wd), None, vp(dx), None, s),
              L.mgvae_conv2d_nhwc_bwd_weight(ctypes.byref(d), vp(xd), vp(dyd), vp(dw), s)]
        torch.cuda.synchronize()
        e = [float((a.double().cpu() - b).abs().max() / b.abs().max()) for a, b in ((yd, yr.detach()), (dx, xr.grad), (dw, wr.grad))]
        flag = "" if max(e) < 1e-4 and rc == [0, 0, 0] else "   <<<<<< BAD"
        bad += bool(flag)
        print("%-36s tile %d split %d rc %s  fwd %.1e dx %.1e dw %.1e%s" % ((N, Cx, H, W, Cy, k, st, p), tile, split, rc, e[0], e[1], e[2], flag), flush=True)
print("bad variants:", bad)
