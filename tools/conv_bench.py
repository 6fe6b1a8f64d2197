"""micro-benchmark of individual conv launches through the C ABI (HIP events, 20 reps)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
arch = ctypes.create_string_buffer(64); cus = ctypes.c_int(0); L.mgvae_device_info(arch, 64, ctypes.byref(cus))
CASES = [  # N, Cx, H, W, Cy, k, s, p
    ("res256 48x8", 64, 256, 48, 8, 256, 3, 1, 1),
    ("res128 96x15", 64, 128, 96, 15, 128, 3, 1, 1),
    ("res64 192x30", 64, 64, 192, 30, 64, 3, 1, 1),
    ("res512 24x4", 64, 512, 24, 4, 512, 3, 1, 1),
    ("res512 6x4", 64, 512, 6, 4, 512, 3, 1, 1),
    ("pool256->512 48x8", 64, 256, 48, 8, 512, 3, 2, 1),
    ("deconv4x4 512<-1024 12x7", 64, 512, 12, 7, 1024, 4, 2, 1),
    ("deconv4x4 64<-128 96x60", 64, 64, 96, 60, 128, 4, 2, 1),
    ("1x1 2048->1024 6x3", 64, 2048, 6, 3, 1024, 1, 1, 0),
    ("1x1 128->64 96x60", 64, 128, 96, 60, 64, 1, 1, 0),
]
def vp(t): return ctypes.c_void_p(t.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("%-28s %10s %10s %10s   (TFLOP/s; us)" % ("case", "fwd", "bwd_data", "bwd_weight"))
for name, N, Cx, H, W, Cy, k, st, p in CASES:
    OH = (H + 2*p - k)//st + 1; OW = (W + 2*p - k)//st + 1
    x = torch.randn(N, Cx, H, W, device=dev); y = torch.randn(N, Cy, OH, OW, device=dev); w = torch.randn(Cy, Cx, k, k, device=dev)
    dw = torch.zeros_like(w)
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 0, 0.0)
    flops = 2.0*N*OH*OW*Cy*Cx*k*k
    res = []
    for fn in (lambda: L.mgvae_conv2d_fwd(ctypes.byref(d), vp(x), vp(w), None, vp(y), s),
               lambda: L.mgvae_conv2d_bwd_data(ctypes.byref(d), vp(y), vp(w), None, vp(x), s),
               lambda: L.mgvae_conv2d_bwd_weight(ctypes.byref(d), vp(x), vp(y), vp(dw), s)):
        for _ in range(3): assert fn() == 0
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)*1e3/20
        res.append((flops/us/1e6, us))
    print("%-28s %5.1f/%-6.0f %5.1f/%-6.0f %5.1f/%-6.0f" % (name, res[0][0], res[0][1], res[1][0], res[1][1], res[2][0], res[2][1]))

if os.environ.get("CONV_BENCH_NO_DIRECT"):
    sys.exit(0)
# ---- the direct (halo-tile, packed-weight) kernels on the same cases: forward only, pack time excluded / included
print("\n%-28s %12s %12s   (direct fwd: TFLOP/s kernel only; with weight packing)" % ("case", "direct", "direct+pack"))
for name, N, Cx, H, W, Cy, k, st, p in CASES:
    OH = (H + 2*p - k)//st + 1; OW = (W + 2*p - k)//st + 1
    x = torch.randn(N, Cx, H, W, device=dev); y = torch.empty(N, Cy, OH, OW, device=dev); w = torch.randn(Cy, Cx, k, k, device=dev)
    d = nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k, k, st, st, p, p, Cx, 0, Cy, 0, 0, 0.0)
    n = L.mgvae_conv_pack_floats(ctypes.byref(d), 0)
    if not n:
        print("%-28s %12s" % (name, "unsupported")); continue
    wp = torch.empty(n, device=dev)
    flops = 2.0*N*OH*OW*Cy*Cx*k*k
    pack = lambda: L.mgvae_conv_pack(ctypes.byref(d), 0, vp(w), vp(wp), s)
    run = lambda: L.mgvae_conv2d_fwd_packed(ctypes.byref(d), vp(x), vp(wp), None, vp(y), s)
    res = []
    for fn in (run, lambda: (pack(), run())):
        pack()
        for _ in range(3): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)*1e3/20
        res.append((flops/us/1e6, us))
    print("%-28s %5.1f/%-6.0f %5.1f/%-6.0f" % (name, res[0][0], res[0][1], res[1][0], res[1][1]))
