"""forced forms of the x3 forward / data gradient on the 3x3 stride-1 trunk layers: implicit K-tile-32 (tile 8, split 1 / 2 / 4) against
the halo forms (12: 128 x 128, 14: 128 x 64, 15: 256 x 64); fp32-equivalent TFLOP/s"""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
def vp(t): return ctypes.c_void_p(t.data_ptr())
WS = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    for _ in range(3): assert fn() == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
for name, N, C, H, W in (("res64 192x30", 64, 64, 192, 30), ("res128 96x15", 64, 128, 96, 15), ("res256 48x8", 64, 256, 48, 8), ("res512 24x4", 64, 512, 24, 4),
                         ("res128 24x15 (2B)", 128, 128, 24, 15), ("res256 12x8 (2B)", 128, 256, 12, 8)):
    x = torch.randn(N, H, W, C, device=dev); y = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(C, 9, C, device=dev) * 0.1
    wk3 = torch.empty(3 * w.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
    assert L.mgvae_pack_conv_weights_x3(vp(w), vp(wk3), vp(wt3), C, 9, C, s) == 0
    d = nat.ConvDesc(N, C, H, W, C, H, W, 3, 3, 1, 1, 1, 1, C, 0, C, 0, 0, 0.0)
    flops = 2.0 * N * H * W * C * C * 9
    out = []
    for tile, split in ((8, 1), (8, 2), (8, 4), (12, 1), (14, 1), (15, 1)):
        os.environ["MGVAE_X3_FORCE"] = "%d,%d" % (tile, split)
        f = timeit(lambda: L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(x), vp(wk3), None, vp(y), None, vp(WS), WS.numel(), s))
        b = timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(y), vp(wt3), None, vp(x), None, vp(WS), WS.numel(), s))
        out.append("t%d/s%d %3.0f %3.0f" % (tile, split, flops / f / 1e6, flops / b / 1e6))
    print("%-20s | %s" % (name, " | ".join(out)), flush=True)
