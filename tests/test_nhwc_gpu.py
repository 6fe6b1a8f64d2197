"""GPU suite: the channels-last (NHWC) kernel family through the C ABI against torch fp64 on the same logical tensors
(torch.channels_last activations AND weights: logical shapes unchanged).  Geometries: every conv of the encoder trunks
(graph/encodingBlock.py:87-126) at both map heights, the decoder's transposed convs and 1x1 convs, channel-sliced
inputs / outputs, split-K, stride phases with odd extents (15 -> 8, 7 -> 15), bias, fused activation and the deferred
activation-gradient mask."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from parity_util import check, flush_report

pytestmark = pytest.mark.gpu
dev = "cuda"
CL = torch.channels_last


@pytest.fixture(scope="module", autouse=True)
def _env():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()
    yield
    flush_report()


def vp(t):
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def cl(t):
    """device copy stored channels-last (also for H = W = 1, where torch considers every layout contiguous)"""
    n, c, h, w = t.shape
    return t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


GEOMS = [  # N, Cx, H, W, Cy, k, s, p
    (3, 64, 48, 30, 64, 3, 1, 1),        # residual 64
    (2, 128, 24, 15, 128, 3, 1, 1),      # residual 128, odd width
    (2, 256, 12, 8, 256, 3, 1, 1),
    (5, 512, 6, 4, 512, 3, 1, 1),        # deep K, small map -> split-K
    (3, 64, 48, 30, 128, 3, 2, 1),       # pooling 64 -> 128
    (5, 128, 24, 15, 256, 3, 2, 1),      # pooling, odd 15 -> 8
    (4, 512, 6, 4, 1024, 3, 2, 1),       # pooling 512 -> 1024, 3x2 out
    (2, 64, 192, 30, 64, 3, 1, 1),       # phrase trunk map
    (3, 2048, 6, 3, 1024, 1, 1, 0),      # fit1 (1x1)
    (2, 128, 96, 60, 64, 1, 1, 0),       # decoder 1x1
    (2, 128, 24, 15, 256, 4, 2, 1),      # 4x4 s2 (transposed conv 256 -> 128 in conv sense: Cx = out)
    (2, 32, 9, 7, 48, 3, 2, 1),          # small, odd, Cy not a multiple of 64
    (1, 16, 5, 4, 16, 3, 1, 1),          # minimum channel count
]


@pytest.mark.parametrize("g", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_nhwc_conv_three_products(g):
    from hipops import _native as nat
    L = nat.lib()
    N, Cx, H, W_, Cy, k, s, p = g
    OH, OW = (H + 2 * p - k) // s + 1, (W_ + 2 * p - k) // s + 1
    x = torch.randn(N, Cx, H, W_).relu_()
    w = torch.randn(Cy, Cx, k, k) * 0.2 - 0.05
    b = torch.randn(Cy)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd, wd, dyd = cl(x), cl(w), cl(dy.float())
    d = nat.ConvDesc(N, Cx, H, W_, Cy, OH, OW, k, k, s, s, p, p, Cx, 0, Cy, 0, 0, 0.0)
    yd = cl(torch.zeros(N, Cy, OH, OW))
    assert L.mgvae_conv2d_nhwc_fwd(ctypes.byref(d), vp(xd), vp(wd), vp(b.to(dev)), vp(yd), None, stream()) == 0
    check("nhwc fwd %s" % (g,), yd, yr)
    dx = cl(torch.zeros(N, Cx, H, W_))
    assert L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d), vp(dyd), vp(wd), None, vp(dx), None, stream()) == 0
    check("nhwc dx %s" % (g,), dx, xr.grad)
    dw = cl(torch.ones(Cy, Cx, k, k))                      # accumulates on top of what is there
    assert L.mgvae_conv2d_nhwc_bwd_weight(ctypes.byref(d), vp(xd), vp(dyd), vp(dw), stream()) == 0
    check("nhwc dw %s" % (g,), dw, 1.0 + wr.grad)
    # transposed conv forward == the data-gradient product with a bias over Cx and a fused activation
    bt = torch.randn(Cx)
    tr = F.relu(F.conv_transpose2d(dy, w.double(), bt.double(), stride=s, padding=p,
                                   output_padding=(H - ((OH - 1) * s - 2 * p + k), W_ - ((OW - 1) * s - 2 * p + k))))
    d2 = nat.ConvDesc(N, Cx, H, W_, Cy, OH, OW, k, k, s, s, p, p, Cx, 0, Cy, 0, 1, 0.0)
    tx = cl(torch.zeros(N, Cx, H, W_))
    assert L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d2), vp(dyd), vp(wd), vp(bt.to(dev)), vp(tx), None, stream()) == 0
    check("nhwc convT+bias+relu %s" % (g,), tx, tr)


def test_nhwc_channel_slices_and_mask():
    """input and output as channel slices of wider channels-last buffers (the zero-copy concat), LeakyReLU epilogue, and
    the deferred activation gradient: dX * act'(mask)"""
    from hipops import _native as nat
    L = nat.lib()
    N, Cx, H, W_, Cy, k = 3, 32, 10, 7, 48, 3
    xb = torch.randn(N, Cx + 16, H, W_); yb = torch.randn(N, Cy + 32, H, W_)
    w = torch.randn(Cy, Cx, k, k) * 0.2
    xs, xo, yo = xb[:, 8:8 + Cx].double(), 8, 16
    want = yb.double().clone()
    want[:, yo:yo + Cy] = F.leaky_relu(F.conv2d(xs, w.double(), None, 1, 1), 0.01)
    xd, yd, wd = cl(xb), cl(yb), cl(w)
    d = nat.ConvDesc(N, Cx, H, W_, Cy, H, W_, k, k, 1, 1, 1, 1, Cx + 16, xo, Cy + 32, yo, 2, 0.01)
    assert L.mgvae_conv2d_nhwc_fwd(ctypes.byref(d), vp(xd), vp(wd), None, vp(yd), None, stream()) == 0
    check("nhwc sliced fwd + leaky", yd, want)
    # data gradient into the x slice, multiplied by relu'(mask)
    mask_t = torch.randn(N, Cx, H, W_)
    dy = yb[:, yo:yo + Cy].double()
    gx = torch.nn.grad.conv2d_input(xs.shape, w.double(), dy, 1, 1) * (mask_t > 0).double()
    wantx = xb.double().clone(); wantx[:, xo:xo + Cx] = gx
    d0 = nat.ConvDesc(N, Cx, H, W_, Cy, H, W_, k, k, 1, 1, 1, 1, Cx + 16, xo, Cy + 32, yo, 0, 0.0)
    md = cl(mask_t)
    m = nat.ActMask(md.data_ptr(), Cx, 0, 1, 0.0)
    xd2 = cl(xb)
    assert L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d0), vp(cl(yb)), vp(wd), None, vp(xd2), ctypes.byref(m), stream()) == 0
    check("nhwc sliced dx * relu'(mask)", xd2, wantx)
    dw = cl(torch.zeros_like(w))
    assert L.mgvae_conv2d_nhwc_bwd_weight(ctypes.byref(d0), vp(cl(xb)), vp(cl(yb)), vp(dw), stream()) == 0
    check("nhwc sliced dw", dw, torch.nn.grad.conv2d_weight(xs, w.shape, dy, 1, 1))


def test_nhwc_rejects_unsupported_channel_counts():
    from hipops import _native as nat
    d = nat.ConvDesc(1, 17, 8, 8, 16, 8, 8, 3, 3, 1, 1, 1, 1, 17, 0, 16, 0, 0, 0.0)
    t = torch.zeros(8, device=dev)
    assert nat.lib().mgvae_conv2d_nhwc_fwd(ctypes.byref(d), vp(t), vp(t), None, vp(t), None, stream()) == -1
