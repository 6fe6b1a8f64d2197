"""the encoder stems' second convs (32 -> 32 channels, 4x1 / 1x4 stride 2) on the channels-last x3 kernels (the bf16 family's K tile is 64 channels: not eligible)"""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
CASES = [("time  N64 384x30", 64, 32, 384, 30, 32, (4, 1), (2, 1), (1, 0)), ("pitch N64 192x60", 64, 32, 192, 60, 32, (1, 4), (1, 2), (0, 1)),
         ("time  N128 96x30", 128, 32, 96, 30, 32, (4, 1), (2, 1), (1, 0)), ("pitch N128 48x60", 128, 32, 48, 60, 32, (1, 4), (1, 2), (0, 1))]
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
    OH = (H + 2*p[0] - k[0])//st[0] + 1; OW = (W + 2*p[1] - k[1])//st[1] + 1
    T = k[0] * k[1]
    x = torch.randn(N, H, W, Cx, device=dev); y = torch.randn(N, OH, OW, Cy, device=dev)
    w = torch.randn(Cy, T, Cx, device=dev) * 0.1
    wk3 = torch.empty(3 * w.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
    assert L.mgvae_pack_conv_weights_x3(vp(w), vp(wk3), vp(wt3), Cy, T, Cx, s) == 0
    dw = torch.zeros(Cy, T, Cx, device=dev)
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k[0], k[1], st[0], st[1], p[0], p[1], Cx, 0, Cy, 0, 0, 0.0)
    flops = 2.0*N*OH*OW*Cy*Cx*T
    b = [timeit(lambda: L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(x), vp(wk3), None, vp(y), None, WS_P, WS_N, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(y), vp(wt3), None, vp(x), None, WS_P, WS_N, s)),
         timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_weight(ctypes.byref(d), vp(x), vp(y), vp(dw), s))]
    mb = (x.numel() + y.numel()) * 4 / 1e6
    print("%-18s | x3 fwd/dgrad/wgrad %4.0f %4.0f %4.0f us (%3.0f %3.0f %3.0f TF) | tensors %.0f MB fp32" % (
        (name,) + tuple(b) + tuple(flops/u/1e6 for u in b) + (mb,)), flush=True)
# the one-channel ends (csrc/thin_nhwc.hip): HBM-bound passes over the 32- / 64-channel map
for name, N, H, W, k, st, p in (("c1 time N64", 64, 384, 60, (4, 1), (2, 1), (1, 0)), ("c1 pitch N64", 64, 384, 60, (1, 4), (1, 2), (0, 1)),
                                ("c1 time N128", 128, 96, 60, (4, 1), (2, 1), (1, 0)), ("c1 time N32", 32, 384, 60, (4, 1), (2, 1), (1, 0))):
    OH = (H + 2*p[0] - k[0])//st[0] + 1; OW = (W + 2*p[1] - k[1])//st[1] + 1
    x = torch.randn(N, 1, H, W, device=dev); w = torch.randn(32, 1, *k, device=dev); y = torch.empty(N, OH, OW, 32, device=dev)
    dw = torch.zeros(32, k[0] * k[1], device=dev)
    d = nat.ConvDesc(N, 1, H, W, 32, OH, OW, k[0], k[1], st[0], st[1], p[0], p[1], 1, 0, 32, 0, 2, 0.01)
    f = timeit(lambda: L.mgvae_conv2d_c1_nhwc_fwd(ctypes.byref(d), vp(x), vp(w), vp(y), 0, s))
    b = timeit(lambda: L.mgvae_conv2d_c1_nhwc_bwd_weight(ctypes.byref(d), vp(x), vp(y), None, vp(dw), 0, s))
    mb = y.numel() * 4 / 1e6
    print("%-14s | fwd %5.1f us (%.2f TB/s)  wgrad %5.1f us (%.2f TB/s) | map %.0f MB" % (name, f, mb / f, b, mb / b, mb), flush=True)
for name, N in (("to1 N64", 64), ("to1 N32", 32), ("to1 N16", 16)):
    rows = N * 96 * 60
    x = torch.randn(rows, 64, device=dev); w = torch.randn(64, device=dev); y = torch.empty(rows, device=dev); dy = torch.randn(rows, device=dev)
    dx = torch.empty_like(x); dw = torch.zeros(64, device=dev)
    f = timeit(lambda: L.mgvae_conv2d_to1_nhwc_fwd(vp(x), vp(w), vp(y), rows, 64, 64, 0, 3, 0.0, 0, s))
    b = timeit(lambda: L.mgvae_conv2d_to1_nhwc_bwd(vp(x), vp(w), vp(y), vp(dy), vp(dx), vp(dw), rows, 64, 64, 0, 3, 0.0, 0, s))
    mb = x.numel() * 4 / 1e6
    print("%-14s | fwd %5.1f us (%.2f TB/s)  bwd %5.1f us (%.2f TB/s, read + write) | map %.0f MB" % (name, f, mb / f, b, 2 * mb / b, mb), flush=True)
