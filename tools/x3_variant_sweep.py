"""every (loop form x tile shape) variant of the x3 kernels (MGVAE_X3_FORCE) against torch fp64 on the model's geometries at
the batch sizes of the 4-bar tests: phrase trunk at N = 4, stacked bar trunk at N = 8, decoder at N = 4"""
import sys, os, ctypes, itertools
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch, torch.nn.functional as F
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
def vp(t): return ctypes.c_void_p(t.data_ptr())
WS = torch.empty(512 << 20, device=dev, dtype=torch.uint8)      # caller-owned split-K workspace (include/mgvae.h), ample for every case
WS_P, WS_N = vp(WS), WS.numel()
def cl(t): return t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
GEOMS = [(4, 64, 192, 30, 64, 3, 1, 1), (4, 64, 192, 30, 128, 3, 2, 1), (4, 128, 96, 15, 128, 3, 1, 1), (4, 128, 96, 15, 256, 3, 2, 1),
         (4, 256, 48, 8, 256, 3, 1, 1), (4, 256, 48, 8, 512, 3, 2, 1), (4, 512, 24, 4, 512, 3, 1, 1), (4, 512, 24, 4, 1024, 3, 2, 1),
         (8, 64, 48, 30, 64, 3, 1, 1), (8, 512, 6, 4, 1024, 3, 2, 1), (4, 2048, 6, 3, 1024, 1, 1, 0),
         (4, 512, 12, 7, 1024, 4, 2, 1), (4, 256, 24, 15, 512, 4, 2, 1), (4, 64, 96, 60, 128, 3, 2, 1), (4, 128, 96, 60, 64, 1, 1, 0)]
bad = 0
for (N, Cx, H, W, Cy, k, st, p) in GEOMS:
    OH, OW = (H + 2*p - k)//st + 1, (W + 2*p - k)//st + 1
    torch.manual_seed(1)
    x = torch.randn(N, Cx, H, W).relu_(); w = torch.randn(Cy, Cx, k, k) * 0.1; dy = torch.randn(N, Cy, OH, OW)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, p); yr.backward(dy.double())
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 0, 0.0)
    xd, wd, dyd = cl(x), cl(w), cl(dy)
    wk3 = torch.empty(3 * w.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
    assert L.mgvae_pack_conv_weights_x3(vp(wd), vp(wk3), vp(wt3), Cy, k * k, Cx, s) == 0
    for tile, split in itertools.product(range(16), (1, 2, 5)):      # split: wgrad pixel splits AND the forward / data gradient's deterministic split-K
        os.environ["MGVAE_X3_FORCE"] = "%d,%d" % (tile, split)
        yd = cl(torch.zeros(N, Cy, OH, OW)); dx = cl(torch.zeros(N, Cx, H, W)); dw = cl(torch.zeros(Cy, Cx, k, k))
        rc = [L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(xd), vp(wk3), None, vp(yd), None, WS_P, WS_N, s),
              L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(dyd), vp(wt3), None, vp(dx), None, WS_P, WS_N, s),
              L.mgvae_conv2d_nhwc_x3_bwd_weight(ctypes.byref(d), vp(xd), vp(dyd), vp(dw), s)]
        torch.cuda.synchronize()
        e = [float((a.double().cpu() - b).abs().max() / b.abs().max()) for a, b in ((yd, yr.detach()), (dx, xr.grad), (dw, wr.grad))]
        flag = "" if max(e) < 2e-5 and rc == [0, 0, 0] else "   <<<<<< BAD"
        bad += bool(flag)
        if flag or split == 1:
            print("%-36s tile %2d split %d rc %s  fwd %.1e dx %.1e dw %.1e%s" % ((N, Cx, H, W, Cy, k, st, p), tile, split, rc, e[0], e[1], e[2], flag), flush=True)
print("bad variants:", bad)
