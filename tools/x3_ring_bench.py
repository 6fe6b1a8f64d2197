"""EXPERIMENT: the LDS-DMA ring form of the x3 forward product (csrc/conv_nhwc_x3_ring.inc, MGVAE_X3_RING=<stages>) against the
tuned implicit / halo forms: max error against the shipped path and fp32-equivalent TFLOP/s, trunk and decoder geometries."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
B = int(os.environ.get("BENCH_B", "64"))
CASES = [("res128 96x15", B, 128, 96, 15, 128, 3, 1, 1), ("res256 48x8", B, 256, 48, 8, 256, 3, 1, 1), ("res512 24x4", B, 512, 24, 4, 512, 3, 1, 1),
         ("res64 192x30", B, 64, 192, 30, 64, 3, 1, 1), ("pool256->512 48x8", B, 256, 48, 8, 512, 3, 2, 1), ("1x1 128->64 96x60", B, 128, 96, 60, 64, 1, 1, 0),
         ("small odd 3x(32->48) 9x7 s2", 3, 32, 9, 7, 48, 3, 2, 1)]
def vp(t): return ctypes.c_void_p(t.data_ptr())
WS = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
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
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Cx, device=dev); w = torch.randn(Cy, k * k, Cx, device=dev) * 0.1; bias = torch.randn(Cy, device=dev)
    wk3 = torch.empty(3 * w.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
    assert L.mgvae_pack_conv_weights_x3(vp(w), vp(wk3), vp(wt3), Cy, k * k, Cx, s) == 0
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 1, 0.0)
    flops = 2.0*N*OH*OW*Cy*Cx*k*k
    y0 = torch.zeros(N, OH, OW, Cy, device=dev)
    fn = lambda y: (lambda: L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(x), vp(wk3), vp(bias), vp(y), None, vp(WS), WS.numel(), s))
    os.environ.pop("MGVAE_X3_RING", None)
    t0 = timeit(fn(y0))
    out = ["tuned %4.0f us %3.0f TF" % (t0, flops / t0 / 1e6)]
    for S in (2, 3, 4, 6):
        os.environ["MGVAE_X3_RING"] = str(S)
        y1 = torch.full((N, OH, OW, Cy), 7.0, device=dev)
        t1 = timeit(fn(y1))
        err = float((y1 - y0).abs().max() / y0.abs().max())
        out.append("ring%d %4.0f us %3.0f TF err %.1e" % (S, t1, flops / t1 / 1e6, err))
    os.environ.pop("MGVAE_X3_RING", None)
    print("%-28s | %s" % (name, " | ".join(out)), flush=True)
