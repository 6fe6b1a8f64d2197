"""What merging DeConvPitchPadding's twin transposed convs (graph/decoder.py:116-120,135-142: same input, same 4x4 / stride-2
geometry) into one launch with 2 x Cout could buy: the two launches of today against ONE launch of the merged shape, for the
transposed conv itself (the stride-phase data-gradient kernel, image side = its output), its input gradient (the forward-conv
kernel, K doubled) and its weight gradient.  Timing only (random operands)."""
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
WS_P, WS_N = vp(WS), WS.numel()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    for _ in range(3): assert fn() == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
N = int(os.environ.get("BENCH_B", "64"))
for name, Ci, h, w, Co in (("1024 -> 2 x 512, 6x3 -> 12x7", 1024, 6, 3, 512), ("512 -> 2 x 256, 12x7 -> 24x15", 512, 12, 7, 256)):
    OH, OW = 2 * h, 2 * w + 1
    res = {}
    for tag, Cx in (("one branch", Co), ("merged", 2 * Co)):
        # conv-sense geometry: image side X = the transposed conv's OUTPUT (Cx channels, OH x OW), feature side Y = its input
        x = torch.randn(N, OH, OW, Cx, device=dev); y = torch.randn(N, h, w, Ci, device=dev)
        wt = torch.randn(Ci, 16, Cx, device=dev) * 0.05
        wk3 = torch.empty(3 * wt.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
        assert L.mgvae_pack_conv_weights_x3(vp(wt), vp(wk3), vp(wt3), Ci, 16, Cx, s) == 0
        dw = torch.zeros(Ci, 16, Cx, device=dev)
        d = nat.ConvDesc(N, Cx, OH, OW, Ci, h, w, 4, 4, 2, 2, 1, 1, Cx, 0, Ci, 0, 0, 0.0)
        res[tag] = (timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(y), vp(wt3), None, vp(x), None, WS_P, WS_N, s)),
                    timeit(lambda: L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(x), vp(wk3), None, vp(y), None, WS_P, WS_N, s)),
                    timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_weight(ctypes.byref(d), vp(x), vp(y), vp(dw), s)))
    a, m = res["one branch"], res["merged"]
    t = torch.empty(N * h * w * Ci, device=dev)
    add = timeit(lambda: L.mgvae_add_inplace(vp(t), vp(t), t.numel(), s))
    print("%-32s | transposed conv 2 x %4.0f vs %4.0f us | its input gradient 2 x %4.0f + add %3.0f vs %4.0f us | weight gradient 2 x %4.0f vs %4.0f us"
          " | block total %4.0f vs %4.0f us" % (name, a[0], m[0], a[1], add, m[1], a[2], m[2], 2 * sum(a) + add, sum(m)), flush=True)
