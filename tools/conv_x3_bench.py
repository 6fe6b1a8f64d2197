"""the fp32-storage channels-last conv kernels on the bf16 matrix pipe (six bf16 MFMAs per fp32 product, csrc/conv_nhwc_x3.inc)
against the fp32-MFMA channels-last kernels on the generator's geometries at batch 64 (HIP events, 20 launches):
fp32-equivalent TFLOP/s forward / data gradient / weight gradient."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
B = int(os.environ.get("BENCH_B", "64"))
CASES = [  # name, N, Cx, H, W, Cy, k, s, p
    ("res64 192x30", B, 64, 192, 30, 64, 3, 1, 1), ("res128 96x15", B, 128, 96, 15, 128, 3, 1, 1),
    ("res256 48x8", B, 256, 48, 8, 256, 3, 1, 1), ("res512 24x4", B, 512, 24, 4, 512, 3, 1, 1),
    ("res64 48x30 (2B)", 2 * B, 64, 48, 30, 64, 3, 1, 1), ("res128 24x15 (2B)", 2 * B, 128, 24, 15, 128, 3, 1, 1),
    ("res256 12x8 (2B)", 2 * B, 256, 12, 8, 256, 3, 1, 1), ("res512 6x4 (2B)", 2 * B, 512, 6, 4, 512, 3, 1, 1),
    ("pool64->128 192x30", B, 64, 192, 30, 128, 3, 2, 1), ("pool256->512 48x8", B, 256, 48, 8, 512, 3, 2, 1),
    ("pool512->1024 24x4", B, 512, 24, 4, 1024, 3, 2, 1),
    ("convT4x4 1024->512 12x7", B, 512, 12, 7, 1024, 4, 2, 1), ("convT4x4 256->128 48x30", B, 128, 48, 30, 256, 4, 2, 1),
    ("convT4x4 128->64 96x60", B, 64, 96, 60, 128, 4, 2, 1), ("convT3x3 128->64 96x60", B, 64, 96, 60, 128, 3, 2, 1),
    ("1x1 2048->1024 6x3", B, 2048, 6, 3, 1024, 1, 1, 0), ("1x1 128->64 96x60", B, 128, 96, 60, 64, 1, 1, 0),
]
def vp(t): return ctypes.c_void_p(t.data_ptr())
WS = torch.empty(512 << 20, device=dev, dtype=torch.uint8)      # caller-owned split-K workspace (include/mgvae.h), ample for every case
WS_P, WS_N = vp(WS), WS.numel()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    for _ in range(3): assert fn() == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
print("%-26s | x3 fwd / dgrad / wgrad TFLOP/s (us) | fp32-MFMA fwd / dgrad / wgrad TFLOP/s" % "case")
tot = 0.0; totf = 0.0; tot0 = 0.0
for name, N, Cx, H, W, Cy, k, st, p in CASES:
    OH = (H + 2*p - k)//st + 1; OW = (W + 2*p - k)//st + 1
    x = torch.randn(N, H, W, Cx, device=dev); y = torch.randn(N, OH, OW, Cy, device=dev)
    w = torch.randn(Cy, k * k, Cx, device=dev) * 0.1
    wk3 = torch.empty(3 * w.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
    assert L.mgvae_pack_conv_weights_x3(vp(w), vp(wk3), vp(wt3), Cy, k * k, Cx, s) == 0
    dw = torch.zeros(Cy, k * k, Cx, device=dev)
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 0, 0.0)
    flops = 2.0*N*OH*OW*Cy*Cx*k*k
    b = [timeit(lambda: L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(x), vp(wk3), None, vp(y), None, WS_P, WS_N, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(y), vp(wt3), None, vp(x), None, WS_P, WS_N, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_weight(ctypes.byref(d), vp(x), vp(y), vp(dw), s))]
    a = [timeit(lambda: L.mgvae_conv2d_nhwc_fwd(ctypes.byref(d), vp(x), vp(w), None, vp(y), None, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d), vp(y), vp(w), None, vp(x), None, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_bwd_weight(ctypes.byref(d), vp(x), vp(y), vp(dw), s))]
    tot += sum(b); tot0 += sum(a); totf += 3 * flops
    print("%-26s | %6.0f %6.0f %6.0f  (%4.0f %4.0f %4.0f us) | %6.0f %6.0f %6.0f" % ((name,) + tuple(flops/u/1e6 for u in b) + tuple(b) + tuple(flops/u/1e6 for u in a)), flush=True)
print("all cases: x3 %.0f us, %.0f TFLOP/s fp32-equivalent (peak 2500 / 6 = 417); fp32-MFMA %.0f us, %.0f TFLOP/s (peak 157)" % (
    tot, totf / tot / 1e6, tot0, totf / tot0 / 1e6))
