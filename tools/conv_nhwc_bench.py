"""side by side: the NCHW implicit-GEMM kernels vs the channels-last family on the generator's conv geometries at batch 64
(HIP events, 20 back-to-back launches each, autotuned).  TFLOP/s forward / data gradient / weight gradient."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
CASES = [  # name, N, Cx, H, W, Cy, k, s, p
    ("res64 192x30", 64, 64, 192, 30, 64, 3, 1, 1), ("res128 96x15", 64, 128, 96, 15, 128, 3, 1, 1),
    ("res256 48x8", 64, 256, 48, 8, 256, 3, 1, 1), ("res512 24x4", 64, 512, 24, 4, 512, 3, 1, 1),
    ("res64 48x30 (2B)", 128, 64, 48, 30, 64, 3, 1, 1), ("res512 6x4 (2B)", 128, 512, 6, 4, 512, 3, 1, 1),
    ("pool64->128 192x30", 64, 64, 192, 30, 128, 3, 2, 1), ("pool128->256 96x15", 64, 128, 96, 15, 256, 3, 2, 1),
    ("pool256->512 48x8", 64, 256, 48, 8, 512, 3, 2, 1), ("pool512->1024 24x4", 64, 512, 24, 4, 1024, 3, 2, 1),
    ("pool512->1024 6x4 (2B)", 128, 512, 6, 4, 1024, 3, 2, 1),
    ("convT4x4 1024->512 12x7", 64, 512, 12, 7, 1024, 4, 2, 1), ("convT4x4 256->128 48x30", 64, 128, 48, 30, 256, 4, 2, 1),
    ("convT4x4 128->64 96x60", 64, 64, 96, 60, 128, 4, 2, 1), ("convT3x3 128->64 96x60", 64, 64, 96, 60, 128, 3, 2, 1),
    ("1x1 2048->1024 6x3", 64, 2048, 6, 3, 1024, 1, 1, 0), ("1x1 128->64 96x60", 64, 128, 96, 60, 64, 1, 1, 0),
]
only = sys.argv[1] if len(sys.argv) > 1 else ""
def vp(t): return ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def cl(t): return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
def timeit(fn):
    for _ in range(3): assert fn() == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
print("%-26s | %-26s | %-26s" % ("case", "NCHW fwd/dgrad/wgrad TF", "NHWC fwd/dgrad/wgrad TF"))
tot = [0.0, 0.0]
for name, N, Cx, H, W, Cy, k, st, p in CASES:
    if only and only not in name: continue
    OH = (H + 2*p - k)//st + 1; OW = (W + 2*p - k)//st + 1
    x = torch.randn(N, Cx, H, W, device=dev); y = torch.randn(N, Cy, OH, OW, device=dev); w = torch.randn(Cy, Cx, k, k, device=dev)
    dw = torch.zeros_like(w)
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 0, 0.0)
    flops = 2.0*N*OH*OW*Cy*Cx*k*k
    a = [timeit(lambda: L.mgvae_conv2d_fwd(ctypes.byref(d), vp(x), vp(w), None, vp(y), s)),
         timeit(lambda: L.mgvae_conv2d_bwd_data(ctypes.byref(d), vp(y), vp(w), None, vp(x), s)),
         timeit(lambda: L.mgvae_conv2d_bwd_weight(ctypes.byref(d), vp(x), vp(y), vp(dw), s))]
    xc, yc, wc, dwc = cl(x), cl(y), cl(w), cl(dw)
    b = [timeit(lambda: L.mgvae_conv2d_nhwc_fwd(ctypes.byref(d), vp(xc), vp(wc), None, vp(yc), None, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d), vp(yc), vp(wc), None, vp(xc), None, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_bwd_weight(ctypes.byref(d), vp(xc), vp(yc), vp(dwc), s))]
    tot[0] += sum(a); tot[1] += sum(b)
    print("%-26s | %5.1f %5.1f %5.1f  (%4.0f us) | %5.1f %5.1f %5.1f  (%4.0f us)" % ((name,) + tuple(flops/u/1e6 for u in a) + (sum(a),) + tuple(flops/u/1e6 for u in b) + (sum(b),)), flush=True)
print("sum of the three products over the cases: NCHW %.0f us, NHWC %.0f us" % tuple(tot))
