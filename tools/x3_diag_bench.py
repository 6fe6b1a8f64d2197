"""Timing-only diagnosis of the x3 forward / data-gradient kernels: forced (tile, split) on a few geometries (results of the
-DMGVAE_X3_DIAG builds are numerically meaningless -- only the times are read)."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
B = 64
CASES = [("res64 192x30", B, 64, 192, 30, 64, 3, 1, 1), ("res128 96x15", B, 128, 96, 15, 128, 3, 1, 1),
         ("res256 48x8", B, 256, 48, 8, 256, 3, 1, 1), ("res512 24x4", B, 512, 24, 4, 512, 3, 1, 1),
         ("convT4x4 256->128 48x30", B, 128, 48, 30, 256, 4, 2, 1)]
def vp(t): return ctypes.c_void_p(t.data_ptr())
WS = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
WS_P, WS_N = vp(WS), WS.numel()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    for _ in range(3): assert fn() == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
for name, N, Cx, H, W, Cy, k, st, p in CASES:
    OH = (H + 2*p - k)//st + 1; OW = (W + 2*p - k)//st + 1
    x = torch.randn(N, H, W, Cx, device=dev); y = torch.randn(N, OH, OW, Cy, device=dev)
    w = torch.randn(Cy, k * k, Cx, device=dev) * 0.1
    wk3 = torch.empty(3 * w.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
    assert L.mgvae_pack_conv_weights_x3(vp(w), vp(wk3), vp(wt3), Cy, k * k, Cx, s) == 0
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 0, 0.0)
    flops = 2.0*N*OH*OW*Cy*Cx*k*k
    out = []
    for tile, split in ((8, 1), (8, 2), (0, 1), (4, 1), (11, 1)):
        os.environ["MGVAE_X3_FORCE"] = "%d,%d" % (tile, split)
        f = timeit(lambda: L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(x), vp(wk3), None, vp(y), None, WS_P, WS_N, s))
        b = timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(y), vp(wt3), None, vp(x), None, WS_P, WS_N, s))
        out.append("t%d/s%d %4.0f %4.0f us (%3.0f %3.0f TF)" % (tile, split, f, b, flops/f/1e6, flops/b/1e6))
    print("%-24s | %s" % (name, " | ".join(out)), flush=True)
