"""Does the half-empty second round of a 720-workgroup launch cost what its workgroup count suggests?  The 128-channel 96 x 15
3x3 layer with the 128 x 128 K-tile-32 form forced (tile 8, no split), at batch sizes that give 0.99 / 1.41 / 2.0 / 2.8 rounds of
512 resident workgroups: fp32-equivalent TFLOP/s forward / data gradient."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'musicgeneration_vae-torch_amd'))
import torch
import __graft_entry__ as g; g.build()
from hipops import _native as nat
L = nat.lib()
dev = 'cuda'
def vp(t): return ctypes.c_void_p(t.data_ptr())
WS = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    for _ in range(3): assert fn() == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20
os.environ["MGVAE_X3_FORCE"] = "8,1"
for C, H, W in ((128, 96, 15), (256, 48, 8)):
    for N in (45, 64, 91, 128, 182):
        x = torch.randn(N, H, W, C, device=dev); y = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(C, 9, C, device=dev) * 0.1
        wk3 = torch.empty(3 * w.numel(), device=dev, dtype=torch.bfloat16); wt3 = torch.empty_like(wk3)
        assert L.mgvae_pack_conv_weights_x3(vp(w), vp(wk3), vp(wt3), C, 9, C, s) == 0
        d = nat.ConvDesc(N, C, H, W, C, H, W, 3, 3, 1, 1, 1, 1, C, 0, C, 0, 0, 0.0)
        flops = 2.0 * N * H * W * C * C * 9
        f = timeit(lambda: L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(x), vp(wk3), None, vp(y), None, vp(WS), WS.numel(), s))
        b = timeit(lambda: L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(y), vp(wt3), None, vp(x), None, vp(WS), WS.numel(), s))
        wgs = -(-N * H * W // 128) * (C // 128)
        print("%d ch %dx%d batch %3d: %4d workgroups = %.2f rounds of 512 | %5.0f %5.0f us | %3.0f %3.0f TFLOP/s | %.3f %.3f us per workgroup" % (
            C, H, W, N, wgs, wgs / 512.0, f, b, flops / f / 1e6, flops / b / 1e6, f / wgs, b / wgs), flush=True)
