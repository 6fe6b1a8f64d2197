"""GPU suite (-m gpu): the HIP path, called through the C ABI (ctypes), against
  (1) plain torch fp32/fp64 CPU ops for every kernel family and every geometry of the model,
  (2) the oracle restatement (oracle/restate.py) block by block and end to end,
  (3) the committed golden fixtures generated from the reference import.
Tolerances (max-norm relative error, written next to each check):
  kernels / blocks: 1e-3 (BASELINE.json north_star: "within 1e-3 rel fp32");
  end to end, ill-conditioned d4 weights: max(1e-3, 2 x torch's own fp32-vs-fp64 error)
  (SURVEY section 7 tolerance rule; the reference itself is off by 3.5e-3 there)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import restate as R
from oracle import weights as W

pytestmark = pytest.mark.gpu

from parity_util import REPORT, TOL, RoundBf16, RoundBf16Forward, check, check_grad, flush_report, pop_margins, rel


@pytest.fixture(scope="module", autouse=True)
def _env():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available(), "GPU suite needs a ROCm device"
    from hipops import _native as nat
    import ctypes
    buf = ctypes.create_string_buffer(64)
    cu = ctypes.c_int(0)
    nat.check(nat.lib().mgvae_device_info(buf, 64, ctypes.byref(cu)), "device_info")
    assert buf.value.decode().startswith("gfx950"), buf.value
    torch.manual_seed(0)
    yield
    flush_report()


dev = "cuda"


def HF():
    from hipops import functional
    return functional


# ------------------------------------------------------------------------- conv family
CONV_GEOMS = [  # (N, Cin, H, W, Cout, k, s, p, bias)  -- every Conv2d geometry of the model + odd cases
    (3, 1, 96, 60, 32, (4, 1), (2, 1), (1, 0), False),      # encoder stem time   (K1)
    (3, 32, 48, 60, 32, (1, 4), (1, 2), (0, 1), False),     # encoder stem pitch  (K1)
    (2, 1, 384, 60, 32, (1, 4), (1, 2), (0, 1), False),     # phrase stem
    (3, 64, 48, 30, 64, (3, 3), (1, 1), (1, 1), False),     # residual 64         (K2)
    (2, 256, 12, 8, 256, (3, 3), (1, 1), (1, 1), False),    # residual 256
    (5, 128, 24, 15, 256, (3, 3), (2, 2), (1, 1), False),   # pooling, odd 15 -> 8 (K3)
    (4, 512, 6, 4, 1024, (3, 3), (2, 2), (1, 1), False),    # pooling 512->1024, 3x2 out
    (3, 2048, 6, 3, 1024, (1, 1), (1, 1), (0, 0), False),   # fit1                (K4)
    (2, 128, 96, 60, 64, (1, 1), (1, 1), (0, 0), False),    # DeConvModule 1x1
    (2, 64, 96, 60, 1, (1, 1), (1, 1), (0, 0), False),      # fit2 (Cout = 1)
    (7, 2304, 1, 1, 1152, (1, 1), (1, 1), (0, 0), True),    # Linear as 1x1        (K13)
    (2, 5, 7, 9, 3, (3, 3), (1, 1), (1, 1), True),          # ragged tiny
    (1, 17, 5, 4, 70, (3, 3), (2, 2), (1, 1), True),        # ragged channels
]


@pytest.fixture(params=["direct", "igemm"])
def conv_path(request):
    """both conv implementations behind the op: the direct halo-tile kernels with packed weights
    and the implicit-GEMM kernels (fallback for geometries the direct kernel refuses)"""
    hf = HF()
    old = hf.USE_DIRECT
    hf.USE_DIRECT = request.param == "direct"
    yield request.param
    hf.USE_DIRECT = old


@pytest.mark.parametrize("g", CONV_GEOMS, ids=lambda g: "x".join(map(str, g[:5])) + "k%dx%d" % g[5])
def test_conv2d_fwd_bwd(g, conv_path):
    N, Ci, H, W_, Co, k, s, p, bias = g
    x = torch.randn(N, Ci, H, W_).relu_()
    w = torch.randn(Co, Ci, *k) * 0.2 - 0.1
    b = torch.randn(Co) if bias else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); wd = torch.nn.Parameter(w.to(dev)); bd = torch.nn.Parameter(b.to(dev)) if bias else None
    y = HF().conv2d(xd, wd, bd, s, p)
    g = (conv_path,) + tuple(g)
    check("conv2d fwd %s" % (g,), y, yr)
    y.backward(dy.float().to(dev))
    check("conv2d dx %s" % (g,), xd.grad, xr.grad)
    check("conv2d dw %s" % (g,), wd.grad, wr.grad)
    if bias:
        check("conv2d db %s" % (g,), bd.grad, br.grad)


@pytest.mark.parametrize("N,K,I", [(64, 512, 512), (128, 1152, 512), (5, 192, 1), (33, 1152, 70), (64, 2304, 1152), (96, 6144, 40)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_skinny_gemm(N, K, I, act):
    """nn.Linear at batch <= 128 takes the skinny-GEMM kernels (no LDS staging, batch = MFMA columns): forward,
    data gradient, weight gradient (accumulating), with column-sliced input and output buffers."""
    hf = HF()
    x = torch.randn(N, K); w = torch.randn(I, K) * 0.05; b = torch.randn(I)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    if act:
        yr = F.relu(yr)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xbuf = torch.zeros(N, K + 64, device=dev)
    xbuf[:, 32:32 + K] = x.to(dev)
    xd = xbuf[:, 32:32 + K].detach().requires_grad_(True)          # a column slice: row stride K + 64
    wd = torch.nn.Parameter(w.to(dev)); bd = torch.nn.Parameter(b.to(dev))
    obuf = torch.full((N, I + 8), 7.0, device=dev)
    y = hf.linear(xd, wd, bd, act=act, out=obuf[:, 4:4 + I])
    tag = "linear N%d K%d I%d act%d" % (N, K, I, act)
    check(tag + " fwd", y, yr)
    assert float(obuf[:, :4].min()) == 7.0 and float(obuf[:, 4 + I:].min()) == 7.0     # neighbours untouched
    y.backward(dy.float().to(dev))
    check(tag + " dx", xd.grad, xr.grad)
    check(tag + " dw", wd.grad, wr.grad)
    check(tag + " db", bd.grad, br.grad)
    # a second backward accumulates into the same gradient
    y2 = hf.linear(xd, wd, bd, act=act)
    y2.backward(dy.float().to(dev))
    check(tag + " dw x2", wd.grad, 2 * wr.grad)


def test_conv2d_fused_activations_and_channel_slices(conv_path):
    x = torch.randn(3, 24, 10, 7)
    w = torch.randn(16, 8, 3, 3) * 0.3
    big = x.to(dev)
    xs = big[:, 8:16]                                  # input is a channel slice
    out = torch.zeros(3, 40, 10, 7, device=dev)
    wd = torch.nn.Parameter(w.to(dev))
    for act, fn in ((1, F.relu), (2, lambda t: F.leaky_relu(t, 0.01)), (3, torch.sigmoid)):
        xin = xs.detach().requires_grad_(True)
        y = HF().conv2d(xin, wd, None, (1, 1), (1, 1), act, 0.01, out=out[:, 5:21])
        xr = x[:, 8:16].double().requires_grad_(True); wr = w.double().requires_grad_(True)
        yr = fn(F.conv2d(xr, wr, padding=1))
        check("conv2d act=%d slice fwd" % act, y, yr)
        assert out[:, :5].abs().max().item() == 0 and out[:, 21:].abs().max().item() == 0
        dy = torch.randn_like(yr)
        yr.backward(dy)
        wd.grad = None
        y.backward(dy.float().to(dev))
        check("conv2d act=%d slice dx" % act, xin.grad, xr.grad)
        check("conv2d act=%d slice dw" % act, wd.grad, wr.grad)


CONVT_GEOMS = [  # (N, Cin, h, w, Cout, k, s, p, op, bias) -- every ConvTranspose2d geometry of the model
    (3, 2304, 1, 1, 1024, (1, 3), (1, 3), (0, 0), (0, 0), False),    # decoder stem (K6)
    (3, 1024, 1, 3, 1024, (6, 1), (6, 1), (0, 0), (0, 0), False),
    (3, 2304, 1, 1, 1024, (6, 1), (6, 1), (0, 0), (0, 0), False),
    (2, 1024, 6, 1, 1024, (1, 3), (1, 3), (0, 0), (0, 0), False),
    (3, 1024, 6, 3, 512, (4, 4), (2, 2), (1, 1), (0, 1), True),      # DeConvPitchPadding (K7) -> 12x7
    (2, 512, 12, 7, 256, (4, 4), (2, 2), (1, 1), (0, 1), True),      # -> 24x15
    (2, 256, 24, 15, 128, (4, 4), (2, 2), (1, 1), (0, 0), False),    # DeConvModule 4x4
    (2, 256, 24, 15, 128, (3, 3), (2, 2), (1, 1), (1, 1), True),     # DeConvModule 3x3 op1 (K8)
    (1, 128, 48, 30, 64, (3, 3), (2, 2), (1, 1), (1, 1), True),
    (2, 6, 5, 4, 3, (4, 4), (2, 2), (1, 1), (0, 1), True),           # ragged tiny
]


@pytest.mark.parametrize("g", CONVT_GEOMS, ids=lambda g: "x".join(map(str, g[:5])) + "k%dx%d" % g[5])
def test_conv_transpose2d_fwd_bwd(g, conv_path):
    N, Ci, h, w_, Co, k, s, p, op, bias = g
    x = torch.randn(N, Ci, h, w_).relu_()
    w = torch.randn(Ci, Co, *k) * 0.1
    b = torch.randn(Co) if bias else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=op)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); wd = torch.nn.Parameter(w.to(dev)); bd = torch.nn.Parameter(b.to(dev)) if bias else None
    y = HF().conv_transpose2d(xd, wd, bd, s, p, op)
    assert tuple(y.shape) == tuple(yr.shape)
    g = (conv_path,) + tuple(g)
    check("convT fwd %s" % (g,), y, yr)
    y.backward(dy.float().to(dev))
    check("convT dx %s" % (g,), xd.grad, xr.grad)
    check("convT dw %s" % (g,), wd.grad, wr.grad)
    if bias:
        check("convT db %s" % (g,), bd.grad, br.grad)


# ---------------------------------------------------------------------- norm / cbam
@pytest.mark.parametrize("shape", [(3, 32, 48, 30), (2, 1024, 3, 2), (2, 512, 12, 7), (2, 64, 96, 60), (1, 16, 1, 6),
                                   (3, 20, 4, 6), (1, 7, 12, 8), (5, 3, 5, 5)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_instance_norm(shape, act):
    x = torch.randn(shape) * 3 + 1
    g = torch.randn(shape[1]); b = torch.randn(shape[1])
    fn = (lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01))[act]
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = fn(F.instance_norm(xr, None, None, gr, br, True, 0.01, 1e-5))
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); gd = torch.nn.Parameter(g.to(dev)); bd = torch.nn.Parameter(b.to(dev))
    y = HF().instance_norm(xd, gd, bd, 1e-5, act, 0.01)
    check("instnorm fwd %s act%d" % (shape, act), y, yr)
    y.backward(dy.float().to(dev))
    check("instnorm dx %s act%d" % (shape, act), xd.grad, xr.grad)
    check("instnorm dgamma %s act%d" % (shape, act), gd.grad, gr.grad)
    check("instnorm dbeta %s act%d" % (shape, act), bd.grad, br.grad)


@pytest.mark.parametrize("shape", [(3, 32, 48, 30), (2, 1024, 6, 3), (2, 128, 24, 15), (2, 64, 12, 7)])
@pytest.mark.parametrize("mode,act", [(0, 0), (1, 1), (1, 2), (2, 1)])
def test_cbam(shape, mode, act):
    N, C, H, W_ = shape
    u = torch.randn(shape); res = torch.randn(shape)
    sd = {"channel_attention.conv1.weight": torch.randn(C // 16, C, 1, 1) * 0.2,
          "channel_attention.conv2.weight": torch.randn(C, C // 16, 1, 1) * 0.2,
          "spatial_attention.conv.weight": torch.randn(1, 2, 3, 3) * 0.3}
    sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    ur, rr = u.double().requires_grad_(True), res.double().requires_grad_(True)
    fn = (lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01))[act]
    o = R.cbam(sdr, "", ur)
    yr = o if mode == 0 else fn(ur + o) if mode == 1 else fn(rr + o)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    ud = u.to(dev).requires_grad_(True); rd = res.to(dev).requires_grad_(True)
    ps = {k: torch.nn.Parameter(v.to(dev)) for k, v in sd.items()}
    y = HF().cbam(ud, ps["channel_attention.conv1.weight"], ps["channel_attention.conv2.weight"],
                  ps["spatial_attention.conv.weight"], mode, rd if mode == 2 else None, act, 0.01)
    tag = "cbam %s mode%d act%d" % (shape, mode, act)
    check(tag + " fwd", y, yr)
    y.backward(dy.float().to(dev))
    check(tag + " du", ud.grad, ur.grad)
    if mode == 2:
        check(tag + " dres", rd.grad, rr.grad)
    for k in sd:
        check(tag + " d" + k, ps[k].grad, sdr[k].grad)


@pytest.mark.parametrize("shape", [(3, 32, 48, 30), (2, 1024, 6, 3), (2, 128, 24, 15), (2, 64, 12, 7), (2, 64, 96, 61),
                                   (1, 256, 12, 8), (3, 32, 6, 4), (5, 16, 3, 2)])
@pytest.mark.parametrize("mode,act", [(1, 1), (1, 2), (2, 1)])
def test_norm_cbam_fused(shape, mode, act):
    """InstanceNorm -> CBAM -> residual -> activation as one node (pooling fused into the norm
    forward, CBAM backward tail fused into the norm backward) against the fp64 oracle."""
    N, C, H, W_ = shape
    x = torch.randn(shape) * 2 + 0.5; res = torch.randn(shape)
    g = torch.randn(C); b = torch.randn(C)
    sd = {"channel_attention.conv1.weight": torch.randn(C // 16, C, 1, 1) * 0.2,
          "channel_attention.conv2.weight": torch.randn(C, C // 16, 1, 1) * 0.2,
          "spatial_attention.conv.weight": torch.randn(1, 2, 3, 3) * 0.3}
    sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    fn = (lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01))[act]
    ur = F.instance_norm(xr, None, None, gr, br, True, 0.01, 1e-5)
    o = R.cbam(sdr, "", ur)
    yr = fn(ur + o) if mode == 1 else fn(rr + o)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); rd = res.to(dev).requires_grad_(True)
    gd = torch.nn.Parameter(g.to(dev)); bd = torch.nn.Parameter(b.to(dev))
    ps = {k: torch.nn.Parameter(v.to(dev)) for k, v in sd.items()}
    y = HF().norm_cbam(xd, gd, bd, ps["channel_attention.conv1.weight"], ps["channel_attention.conv2.weight"],
                       ps["spatial_attention.conv.weight"], 1e-5, mode, rd if mode == 2 else None, act, 0.01)
    tag = "norm_cbam %s mode%d act%d" % (shape, mode, act)
    check(tag + " fwd", y, yr)
    y.backward(dy.float().to(dev))
    check(tag + " dx", xd.grad, xr.grad)
    check(tag + " dgamma", gd.grad, gr.grad)
    check(tag + " dbeta", bd.grad, br.grad)
    if mode == 2:
        check(tag + " dres", rd.grad, rr.grad)
    for k in sd:
        check(tag + " d" + k, ps[k].grad, sdr[k].grad)


# ------------------------------------------------------------------------- small ops
def test_small_ops():
    hf = HF()
    # whole-map average pooling
    x = torch.randn(5, 1024, 3, 2)
    xr = x.double().requires_grad_(True); yr = F.avg_pool2d(xr, (3, 2)).view(5, 1024); dy = torch.randn_like(yr); yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); y = hf.global_avg_pool(xd); y.backward(dy.float().to(dev))
    check("avgpool fwd", y, yr); check("avgpool dx", xd.grad, xr.grad)
    # embedding into a column slice + scatter-add gradient (repeated indices)
    tab = torch.randn(332, 1152); idx = torch.tensor([3, 331, 3, 0, 17])
    tr = tab.double().requires_grad_(True); er = F.embedding(idx, tr); de = torch.randn_like(er); er.backward(de)
    td = torch.nn.Parameter(tab.to(dev)); buf = torch.zeros(5, 2304, device=dev)
    e = hf.embedding(idx.to(dev), td, out=buf[:, 1152:]); e.backward(de.float().to(dev))
    check("embedding fwd", e, er); check("embedding dtable", td.grad, tr.grad)
    assert buf[:, :1152].abs().max().item() == 0
    # zero-copy concat + its backward split
    a = torch.randn(4, 6, 5, 3, device=dev, requires_grad=True); b = torch.randn(4, 10, 5, 3, device=dev, requires_grad=True)
    cat = torch.empty(4, 16, 5, 3, device=dev)
    g1 = torch.nn.Parameter(torch.ones(6, device=dev)); g2 = torch.nn.Parameter(torch.ones(10, device=dev))
    z1 = torch.nn.Parameter(torch.zeros(6, device=dev)); z2 = torch.nn.Parameter(torch.zeros(10, device=dev))
    pa = hf.instance_norm(a, g1, z1, out=cat[:, :6]); pb = hf.instance_norm(b, g2, z2, out=cat[:, 6:])
    j = hf.join(cat, pa, pb)
    ar, br_ = a.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    jr = torch.cat((F.instance_norm(ar), F.instance_norm(br_)), 1)
    dj = torch.randn_like(jr); jr.backward(dj); j.backward(dj.float().to(dev))
    check("join fwd", j, jr); check("join da", a.grad, ar.grad); check("join db", b.grad, br_.grad)
    # dropout: keep probability, scaling, backward uses the same mask
    hf.manual_seed(123)
    x = torch.ones(64, 1152, device=dev, requires_grad=True)
    y = hf.dropout(x, 0.3, True)
    keep = (y > 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01, keep
    assert abs(y.max().item() - 1 / 0.7) < 1e-6
    y.sum().backward()
    assert torch.equal(x.grad, y.detach())
    y2 = hf.dropout(x, 0.3, True)
    assert not torch.equal(y2, y), "Philox offset must advance"
    assert hf.dropout(x, 0.3, False) is x
    # Gaussian prior noise
    n = hf.randn((64, 1152), 1.5)
    assert abs(n.mean().item()) < 0.02 and abs(n.std().item() - 1.5) < 0.02


def test_losses():
    hf = HF()
    from graph.loss.bar_loss import Loss, DLoss
    gen = torch.rand(4, 1, 96, 60) * 0.98 + 0.01
    gen[0, 0, 0, :5] = torch.tensor([0.0, 1.0, 1e-30, 1 - 1e-8, 0.3])      # clamp / saturation edge cases
    lab = (torch.rand(4, 1, 96, 60) < 0.05).float()
    for pre in (True, False):
        gr = gen.double().requires_grad_(True)
        lr_ = R.bar_loss(gr.float(), lab, pre)
        g32 = gen.clone().requires_grad_(True); lr32 = R.bar_loss(g32, lab, pre); lr32.backward()
        gd = gen.to(dev).requires_grad_(True)
        l = Loss().to(dev)(gd, lab.to(dev), pre)
        check("Loss pretraining=%s value" % pre, l, lr32)
        l.backward()
        check("Loss pretraining=%s grad" % pre, gd.grad, g32.grad)
    d = torch.rand(64) * 0.9 + 0.05
    for val in (0.0, 1.0):
        dr = d.clone().requires_grad_(True); lr_ = R.dloss(dr, torch.full((64,), val)); lr_.backward()
        dd = d.to(dev).requires_grad_(True); l = DLoss.constant(dd, val); l.backward()
        check("DLoss const %g" % val, l, lr_); check("DLoss const %g grad" % val, dd.grad, dr.grad)
        dd2 = d.to(dev).requires_grad_(True); l2 = DLoss()(dd2, torch.full((64,), val, device=dev))
        check("DLoss tensor %g" % val, l2, lr_)
    # reparameterisation + KL (closed form: mu = 0, logvar = 0 -> KL = 0, z = eps)
    mu = torch.randn(8, 1152); lv = torch.randn(8, 1152) * 0.3; eps = torch.randn(8, 1152)
    mr, lvr = mu.double().requires_grad_(True), lv.double().requires_grad_(True)
    zr = R.reparameterize(mr, lvr, eps.double()); kr = R.kl_term(mr, lvr)
    (zr.sum() * 0.3 + kr * 0.01).backward()
    md, lvd = mu.to(dev).requires_grad_(True), lv.to(dev).requires_grad_(True)
    z, kl = hf.reparam_kl(md, lvd, eps.to(dev))
    (z.sum() * 0.3 + kl * 0.01).backward()
    check("reparam z", z, zr); check("kl", kl, kr); check("reparam dmean", md.grad, mr.grad); check("reparam dlogvar", lvd.grad, lvr.grad)
    z0, kl0 = hf.reparam_kl(torch.zeros(4, 1152, device=dev), torch.zeros(4, 1152, device=dev), eps[:4].to(dev))
    assert kl0.item() == 0.0 and torch.equal(z0.cpu(), eps[:4])


def test_flat_adam_matches_torch():
    from hipops import FlatParams
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in ((33, 7), (1000,), (5, 5, 3, 3), (1,))]
    ref = [p.detach().cpu().clone().requires_grad_(True) for p in ps]
    opt = FlatParams(ps, lr=0.002)
    topt = torch.optim.Adam(ref, lr=0.002)
    for step in range(3):
        opt.zero_grad(); topt.zero_grad()
        for p, r in zip(ps, ref):
            g = torch.randn_like(r)
            r.grad = g.clone(); p.grad.copy_(g.to(dev))
        opt.step(); topt.step()
    for i, (p, r) in enumerate(zip(ps, ref)):
        check("adam param %d" % i, p, r, 1e-5)
        assert p.data_ptr() >= opt.flat.data_ptr() and p.data_ptr() < opt.flat.data_ptr() + opt.flat.numel() * 4


# ------------------------------------------------------------------------------ blocks
def _load(mod, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    mod.load_state_dict(sub)
    return mod.to(dev), {k: v.clone().double().requires_grad_(True) for k, v in sub.items()}


def _block_check(tag, mod, osd, ofn, x, tol=TOL):
    xr = x.double().requires_grad_(True)
    yr = ofn(osd, "", xr)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    o32 = {k: v.detach().float().requires_grad_(True) for k, v in osd.items()}
    x32 = x.clone().requires_grad_(True)
    ofn(o32, "", x32).backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    y = mod(xd)
    check(tag + " fwd", y, yr, tol)
    y.backward(dy.float().to(dev))
    check_grad(tag + " dx", xd.grad, xr.grad, tol, ref32=x32.grad)
    # absolute floor for gradients that are analytically zero / pure cancellation (bias in
    # front of an InstanceNorm, d4 affine terms): 1e-6 of the block's largest gradient entry
    gscale = max(v.grad.abs().max().item() for v in osd.values() if v.grad is not None)
    for n, p in mod.named_parameters():
        if osd[n].grad is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, n
            continue
        check_grad(tag + " d" + n, p.grad, osd[n].grad, tol, atol=1e-6 * gscale, ref32=o32[n].grad)


@pytest.mark.parametrize("mode", ["wc", "d4"])
def test_blocks_against_oracle(mode):
    import graph.encodingBlock as EB
    import graph.decoder as DD
    gsd = W.make_state_dict(W.manifest_generator(), 0, mode)
    # d4 (N(-1,1) weights): gradients pass through InstanceNorm of huge pre-activations; the
    # fp64 oracle is the judge and fp32 noise is larger -> 5e-3 on that weight set only
    tol = TOL if mode == "wc" else 5e-3
    torch.manual_seed(1)
    B = 3
    cases = [
        ("enc.time_pitch", EB.TimePitchModule(), "encoder.time_pitch.", R.enc_time_pitch, (torch.rand(B, 1, 96, 60) < 0.05).float()),
        ("enc.pitch_time", EB.PitchTimeModule(), "encoder.pitch_time.", R.enc_pitch_time, (torch.rand(B, 1, 96, 60) < 0.05).float()),
        ("enc.residual64", EB.ResidualModule(64), "encoder.layers.0.", R.residual_module, torch.randn(B, 64, 48, 30).relu_()),
        ("enc.pooling64", EB.PoolingModule(64, 128), "encoder.layers.1.", R.pooling_module, torch.randn(B, 64, 48, 30).relu_()),
        ("enc.residual512", EB.ResidualModule(512), "encoder.layers.6.", R.residual_module, torch.randn(B, 512, 6, 4).relu_()),
        ("enc.pooling512", EB.PoolingModule(512, 1024), "encoder.layers.7.", R.pooling_module, torch.randn(B, 512, 6, 4).relu_()),
        ("dec.pitch_time", DD.PitchTimeModule(), "decoder.pitch.", R.dec_pitch_time, torch.randn(B, 2304, 1, 1).relu_()),
        ("dec.time_pitch", DD.TimePitchModule(), "decoder.time.", R.dec_time_pitch, torch.randn(B, 2304, 1, 1).relu_()),
        ("dec.deconv_pp1024", DD.DeConvPitchPadding(1024, 512), "decoder.layers.0.", R.deconv_pitch_padding, torch.randn(B, 1024, 6, 3).relu_()),
        ("dec.deconv_pp512", DD.DeConvPitchPadding(512, 256), "decoder.layers.1.", R.deconv_pitch_padding, torch.randn(B, 512, 12, 7).relu_()),
        ("dec.deconv256", DD.DeConvModule(256, 128), "decoder.layers.2.", R.deconv_module, torch.randn(2, 256, 24, 15).relu_()),
        ("dec.deconv128", DD.DeConvModule(128, 64), "decoder.layers.3.", R.deconv_module, torch.randn(2, 128, 48, 30).relu_()),
    ]
    for tag, mod, prefix, ofn, x in cases:
        mod, osd = _load(mod, gsd, prefix)
        _block_check("%s[%s]" % (tag, mode), mod, osd, ofn, x, tol)


# -------------------------------------------------------------------------- end to end
def _generator(mode):
    from graph.model import Model
    gsd = W.make_state_dict(W.manifest_generator(), 0, mode)
    m = Model()
    m.load_state_dict(gsd)
    return m.to(dev).eval(), gsd


@pytest.mark.parametrize("mode", ["wc", "d4"])
def test_generator_forward_against_golden(golden_dir, mode):
    fx = np.load(os.path.join(golden_dir, "generator_%s.npz" % mode))
    m, _ = _generator(mode)
    note, pre, phrase, pos = W.make_inputs(4, seed=1234)
    with torch.no_grad():
        gen, z, pz, pf = m(note.to(dev), pre.to(dev), phrase.to(dev), pos.to(dev))
    t = lambda k: torch.from_numpy(fx[k])
    # encoders: well inside 1e-3 on both weight sets
    check("e2e[%s] z vs reference fp32" % mode, z, t("z"))
    check("e2e[%s] pre_z vs reference fp32" % mode, pz, t("pre_z"))
    check("e2e[%s] phrase_feature vs reference fp32" % mode, pf, t("phrase_feature"))
    # decoder: tolerance rule of SURVEY section 7 against the fp64 run
    ref_err = rel(t("gen"), t("gen64"))
    tol = max(TOL, 2 * ref_err)
    check("e2e[%s] gen vs fp64 oracle (torch fp32 itself: %.2e)" % (mode, ref_err), gen, t("gen64"), tol)
    # ... and against the reference's OWN fp32 output (north_star: "1e-3 rel vs reference CPU forward"): two fp32
    # evaluations of an ill-conditioned decoder can each be ref_err from the truth, so that is the bar on the d4 weights
    check("e2e[%s] gen vs the reference's fp32 fixture" % mode, gen, t("gen"), max(TOL, 2 * ref_err))
    flips = int(((gen.cpu().numpy() > 0.3) != (fx["gen64"] > 0.3)).sum())
    REPORT.append("e2e[%s] binarised mismatches vs fp64: %d of %d" % (mode, flips, gen.numel()))
    assert flips <= (3 if mode == "d4" else 0)


def _fp32_floor(gsd, zsd, inputs, names, seeds=(101, 102, 103)):
    """gradients of the fp32 oracle step and of runs whose generator weights were moved by <= 2e-6 relative (2^-19: the size
    of an fp32 dot product's own accumulation error at the model's K -- "the same arithmetic summed in another order / on
    another pipe"): per name a list of tensors, the noise floor parity_util.check_grad judges a row against"""
    note, pre, phrase, pos = inputs
    runs = []
    for seed in (None,) + tuple(seeds):
        osd32 = {k: v.clone() for k, v in gsd.items()}
        if seed is not None:
            g = torch.Generator().manual_seed(seed)
            for t in osd32.values():
                if t.is_floating_point():
                    t.mul_(1.0 + 2.0 ** -19 * (2.0 * torch.rand(t.shape, generator=g) - 1.0))
        osd32 = {k: (v.requires_grad_(True) if v.is_floating_point() else v) for k, v in osd32.items()}
        lo32, _ = R.pretrain_step_loss(osd32, zsd, zsd, note, pre, phrase, pos, True)
        runs.append(torch.autograd.grad(lo32, [osd32[n] for n in names], allow_unused=True))
        del osd32
    return [[r[i] for r in runs] if runs[0][i] is not None else None for i in range(len(names))]


def test_train_step_against_oracle_and_golden(golden_dir):
    """one barGen2 pre-training generator step (agent/barGen2.py:267-292): loss, every
    parameter gradient, and the post-Adam parameters."""
    mode = "wc"
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.loss.bar_loss import Loss, DLoss
    from hipops import FlatParams
    fx = np.load(os.path.join(golden_dir, "generator_%s.npz" % mode))
    gn = json.load(open(os.path.join(golden_dir, "gradnorm_%s.json" % mode)))
    m, gsd = _generator(mode)
    zsd = W.make_state_dict(W.manifest_z_discriminator(), 0, mode)
    zb, zp = BarZDiscriminator(), PhraseZDiscriminator()
    zb.load_state_dict(zsd); zp.load_state_dict(zsd)
    zb, zp = zb.to(dev), zp.to(dev)
    for d in (zb, zp):
        for p in d.parameters():
            p.requires_grad = False
    note, pre, phrase, pos = W.make_inputs(4, seed=1234)
    opt = FlatParams(list(m.parameters()), lr=0.002)
    opt.zero_grad()
    gen, z, pz, pf = m(note.to(dev), pre.to(dev), phrase.to(dev), pos.to(dev))
    loss = DLoss.constant(zp(pf).view(-1), 1.0)
    loss = loss + DLoss.constant(zb(z).view(-1), 1.0) + DLoss.constant(zb(pz).view(-1), 1.0)
    loss = loss + Loss().to(dev)(gen, note.to(dev), True)
    loss.backward()
    check("step loss vs golden", loss, torch.tensor(float(fx["loss_pretrain"])))
    # oracle gradients in fp64 (the judge) and in torch fp32 (what the reference itself achieves)
    osd = {k: v.clone().double().requires_grad_(True) for k, v in gsd.items()}
    z64 = {k: v.double() for k, v in zsd.items()}
    lo, _ = R.pretrain_step_loss(osd, z64, z64, note.double(), pre.double(), phrase.double(), pos, True)
    names = list(gn["grad"].keys())
    og = torch.autograd.grad(lo, [osd[n] for n in names])
    og32 = _fp32_floor(gsd, zsd, (note, pre, phrase, pos), names)       # plain fp32 + three perturbed runs
    params = dict(m.named_parameters())
    gscale = max(g.abs().max().item() for g in og)
    worst, worst_ref = 0.0, 0.0
    for n, g, g32 in zip(names, og, og32):
        e, e32 = rel(params[n].grad, g), max(rel(t, g) for t in g32)
        if g.abs().max().item() > 1e-6 * gscale:     # skip analytically-zero gradients (bias before an InstanceNorm)
            worst, worst_ref = max(worst, e), max(worst_ref, e32)
        REPORT.append("  grad %-60s hip-vs-fp64=%.3e torch32-vs-fp64=%.3e |g|=%.3e golden|g|=%.3e" % (
            n, e, e32, params[n].grad.double().norm().item(), gn["grad"][n][0]))
        # whole-step gradients cross ~40 layers of ReLU / max-pool masks: 2e-3 max-norm here,
        # 1e-3 stays the bar for every kernel and block above; beyond that only what torch fp32 itself
        # misses against fp64 is admitted (parity_util.check_grad)
        check_grad("step d" + n, params[n].grad, g, 2 * TOL, atol=1e-6 * gscale, ref32=g32)
        assert abs(params[n].grad.double().norm().item() - gn["grad"][n][0]) <= 3e-2 * gn["grad"][n][0] + 1e-6 * gscale, n
    REPORT.append("step: worst max-norm gradient error vs fp64: hip %.3e, torch fp32 itself %.3e (mask flips; see check_grad) over %d tensors" % (
        worst, worst_ref, len(names)))
    pop_margins("4-bar step vs fp64 oracle", 10)
    for n in gn["unused"]:
        assert params[n].grad.abs().max().item() == 0.0, n
    # Adam step (first step of torch.optim.Adam defaults) applied to the HIP gradients themselves
    before = {n: p.detach().clone() for n, p in params.items()}
    hipg = {n: p.grad.detach().clone() for n, p in params.items()}
    opt.step()
    torch.cuda.synchronize()
    for n in names[::17]:
        p0, g = before[n].double().cpu(), hipg[n].double().cpu()
        mm = 0.1 * g; vv = 0.001 * g * g
        want = p0 - (0.002 / 0.1) * mm / (vv.sqrt() / (0.001 ** 0.5) + 1e-8)
        check("adam " + n, params[n].detach().double().cpu() - p0, want - p0, 1e-4)


def test_full_size_step_parity_and_batch_properties():
    """BASELINE.json configs[1] at its full size (64 bars per GPU), three ways:
      * against the oracle's fp32 step on the host (loss, gradient norms, gradients with the mask-flip rule);
      * per-sample independence (InstanceNorm + per-sample CBAM, dropout off): the first 4 rows of the 64-bar forward
        equal the 4-bar forward of those rows -- other tiles, split-K factors and kernel variants are chosen at 4;
      * linearity: every loss term is a batch mean, so the 64-bar gradient is the mean of the 16 4-bar gradients."""
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.loss.bar_loss import Loss, DLoss
    from hipops import FlatParams
    mode, B = "wc", 64
    m, gsd = _generator(mode)
    zsd = W.make_state_dict(W.manifest_z_discriminator(), 0, mode)
    zb, zp = BarZDiscriminator(), PhraseZDiscriminator()
    zb.load_state_dict(zsd); zp.load_state_dict(zsd)
    zb, zp = zb.to(dev), zp.to(dev)
    for d in (zb, zp):
        for prm in d.parameters():
            prm.requires_grad = False
    note, pre, phrase, pos = W.make_inputs(B, seed=4321)
    opt = FlatParams(list(m.parameters()), lr=0.002)
    lossf = Loss().to(dev)

    def step(sl):
        opt.zero_grad()
        n_, p_, ph_, po_ = (t[sl].to(dev) for t in (note, pre, phrase, pos))
        gen, z, pz, pf = m(n_, p_, ph_, po_)
        loss = DLoss.constant(zp(pf).view(-1), 1.0) + DLoss.constant(zb(z).view(-1), 1.0) + DLoss.constant(zb(pz).view(-1), 1.0)
        loss = loss + lossf(gen, n_, True)
        loss.backward()
        return loss.detach(), gen.detach(), opt.grad.clone()

    loss64, gen64, g64 = step(slice(0, B))
    # (1) the oracle on the same 64 bars: fp64 is the judge, its fp32 run says what the reference's own arithmetic
    # achieves (rules of parity_util.check_grad: strict 2e-3, else 3 x torch fp32, else flip-tolerant)
    params = dict(m.named_parameters())
    names = [n for n in params if n in gsd]
    osd = {k: v.clone().double().requires_grad_(True) for k, v in gsd.items()}
    z64 = {k: v.double() for k, v in zsd.items()}
    lo, _ = R.pretrain_step_loss(osd, z64, z64, note.double(), pre.double(), phrase.double(), pos, True)
    og = torch.autograd.grad(lo, [osd[n] for n in names], allow_unused=True)
    del osd
    og32 = _fp32_floor(gsd, zsd, (note, pre, phrase, pos), names, seeds=(101,))      # plain fp32 + one perturbed run
    check("full-size step loss vs oracle fp64", loss64, lo.detach(), 1e-4)
    gscale = max(g.abs().max().item() for g in og if g is not None)
    pop_margins("(rows before the full-size step)", 0)
    rules = {}
    for n, g, g32 in zip(names, og, og32):
        if g is None or g.abs().max().item() <= 1e-6 * gscale:
            continue
        r = check_grad("full-size d" + n, params[n].grad, g, 2 * TOL, atol=1e-6 * gscale, ref32=g32)
        rules[r] = rules.get(r, 0) + 1
    pop_margins("full-size step (64 bars) vs fp64 oracle", 10)
    REPORT.append("full-size step: admitted by rule: %s" % rules)
    # most rows must carry information: the uninformative ones (torch fp32 itself > 1 %% off) are the stems' tied planes
    assert rules.get("uninformative", 0) <= 0.35 * sum(rules.values()), rules
    # (2) per-sample independence
    _, gen4, _ = step(slice(0, 4))
    check("rows 0-3 of the 64-bar forward == the 4-bar forward", gen64[:4], gen4, 1e-4)
    # (3) linearity over the batch
    acc = torch.zeros_like(g64)
    for c in range(B // 4):
        acc += step(slice(4 * c, 4 * c + 4))[2]
    acc /= B // 4
    err = float((acc - g64).norm() / g64.norm())
    REPORT.append("full size: |mean of 16 4-bar gradients - 64-bar gradient| / |64-bar gradient| = %.3e" % err)
    assert err < 2e-3, err


# ------------------------------------------------------------------------- discriminators
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act", [0, 1])
def test_batch_norm(training, act):
    x = torch.randn(5, 16, 12, 7) * 2 + 0.5
    g = torch.randn(16); b = torch.randn(16)
    rm, rv = torch.randn(16) * 0.1, torch.rand(16) + 0.5
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    rmr, rvr = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rmr, rvr, gr, br, training, 0.01, 1e-5)
    yr = F.relu(yr) if act else yr
    dy = torch.randn_like(yr); yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); gd = torch.nn.Parameter(g.to(dev)); bd = torch.nn.Parameter(b.to(dev))
    rmd, rvd = rm.to(dev), rv.to(dev)
    y = HF().batch_norm(xd, gd, bd, rmd, rvd, training, 0.01, 1e-5, act)
    tag = "batchnorm train=%s act=%d" % (training, act)
    check(tag + " fwd", y, yr)
    y.backward(dy.float().to(dev))
    check(tag + " dx", xd.grad, xr.grad); check(tag + " dgamma", gd.grad, gr.grad); check(tag + " dbeta", bd.grad, br.grad)
    check(tag + " running_mean", rmd, rmr); check(tag + " running_var", rvd, rvr)


def test_group_sum_and_cat_time():
    x = torch.randn(3, 1, 192, 60)
    xr = x.double().requires_grad_(True)
    cr = torch.sum(xr.view(-1, 1, 192, 12, 5), 4); orr = torch.sum(xr, 3, keepdim=True)
    d1, d2 = torch.randn_like(cr), torch.randn_like(orr)
    (cr * d1).sum().backward(retain_graph=True); g1 = xr.grad.clone(); xr.grad = None
    (orr * d2).sum().backward(); g2 = xr.grad.clone()
    xd = x.to(dev).requires_grad_(True)
    c = HF().group_sum(xd, 5); check("group_sum(5) fwd", c, cr)
    (c * d1.float().to(dev)).sum().backward(); check("group_sum(5) bwd", xd.grad, g1); xd.grad = None
    o = HF().group_sum(xd, 60); check("group_sum(60) fwd", o, orr)
    (o * d2.float().to(dev)).sum().backward(); check("group_sum(60) bwd", xd.grad, g2)
    a = torch.randn(3, 1, 96, 60, device=dev, requires_grad=True); b = torch.randn(3, 1, 96, 60, device=dev, requires_grad=True)
    ct = HF().cat_time(a, b)
    assert torch.equal(ct, torch.cat((a, b), dim=2))
    w = torch.randn_like(ct); (ct * w).sum().backward()
    assert torch.equal(a.grad, w[:, :, :96]) and torch.equal(b.grad, w[:, :, 96:])


@pytest.mark.parametrize("mode", ["wc", "d4"])
def test_discriminators_against_oracle_and_golden(golden_dir, mode):
    from graph.bar_discriminator import BarDiscriminator
    from graph.z_discriminator import BarZDiscriminator
    from graph.bar_discriminator_with_feature import BarFeatureDiscriminator
    fx = np.load(os.path.join(golden_dir, "generator_%s.npz" % mode))
    z = torch.from_numpy(fx["z"])
    zsd = W.make_state_dict(W.manifest_z_discriminator(), 0, mode)
    fsd = W.make_state_dict(W.manifest_bar_feature_discriminator(), 0, mode)
    bsd = W.make_state_dict(W.manifest_bar_discriminator(), 0, mode)
    zb = BarZDiscriminator(); zb.load_state_dict(zsd); zb = zb.to(dev)
    fd = BarFeatureDiscriminator(); fd.load_state_dict(fsd); fd = fd.to(dev)
    check("BarZDiscriminator[%s] vs golden" % mode, zb(z.to(dev)), torch.from_numpy(fx["d_zbar"]))
    check("BarFeatureDiscriminator[%s] vs golden" % mode, fd(z.to(dev)), torch.from_numpy(fx["d_feature"]))
    # bar discriminator: forward vs golden (reference import), gradients vs the fp64 oracle
    note, pre, _, _ = W.make_inputs(4, seed=1234)
    pair = torch.cat((pre, note), dim=2)
    bd = BarDiscriminator(); bd.load_state_dict(bsd); bd = bd.to(dev).train()
    pd_ = pair.to(dev).requires_grad_(True)
    out = bd(pd_)
    tol = TOL if mode == "wc" else 5e-3
    check("BarDiscriminator[%s] fwd vs golden" % mode, out, torch.from_numpy(fx["d_bar"]), tol)
    run = torch.cat([v.detach().flatten().cpu() for k, v in bd.state_dict().items() if "running" in k])
    check("BarDiscriminator[%s] running stats vs golden" % mode, run, torch.from_numpy(fx["bd_running"]), tol)
    osd = {k: (v.clone().double().requires_grad_(True) if v.is_floating_point() and "running" not in k else
               (v.clone().double() if v.is_floating_point() else v.clone())) for k, v in bsd.items()}
    pr = pair.double().requires_grad_(True)
    o = R.bar_discriminator(osd, "", pr, train=True)
    dy = torch.randn_like(o); o.backward(dy)
    o32sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in bsd.items()}
    p32 = pair.clone().requires_grad_(True)
    R.bar_discriminator(o32sd, "", p32, train=True).backward(dy.float())
    out.backward(dy.float().to(dev))
    gscale = max(v.grad.abs().max().item() for v in osd.values() if getattr(v, "grad", None) is not None)
    check_grad("BarDiscriminator[%s] dx" % mode, pd_.grad, pr.grad, tol, ref32=p32.grad)
    for n, p in bd.named_parameters():
        if osd[n].grad is None:      # basic.layers.2.bn1.* is constructed but never used (isBasic=True)
            assert p.grad is None or p.grad.abs().max().item() == 0, n
            continue
        check_grad("BarDiscriminator[%s] d%s" % (mode, n), p.grad, osd[n].grad, tol, atol=1e-6 * gscale, ref32=o32sd[n].grad)


def test_standalone_channel_and_spatial_attention():
    """graph.cbam.ChannelAttention / SpatialAttention used on their own (reference graph/cbam.py:7-52)"""
    from graph.cbam import ChannelAttention, SpatialAttention
    x = torch.randn(3, 64, 12, 7)
    dy = torch.randn(3, 64, 12, 7, dtype=torch.float64)
    sd = {"conv1.weight": torch.randn(4, 64, 1, 1) * 0.2, "conv2.weight": torch.randn(64, 4, 1, 1) * 0.2}
    ca = ChannelAttention(64); ca.load_state_dict(sd); ca = ca.to(dev)
    xr = x.double().requires_grad_(True); sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    yr = R.channel_attention(sdr, "", xr); yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); y = ca(xd); y.backward(dy.float().to(dev))
    check("ChannelAttention fwd", y, yr); check("ChannelAttention dx", xd.grad, xr.grad)
    check("ChannelAttention dconv1", ca.conv1.weight.grad, sdr["conv1.weight"].grad)
    check("ChannelAttention dconv2", ca.conv2.weight.grad, sdr["conv2.weight"].grad)
    sd = {"conv.weight": torch.randn(1, 2, 3, 3) * 0.3}
    sa = SpatialAttention(); sa.load_state_dict(sd); sa = sa.to(dev)
    xr = x.double().requires_grad_(True); sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    yr = R.spatial_attention(sdr, "", xr); yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); y = sa(xd); y.backward(dy.float().to(dev))
    check("SpatialAttention fwd", y, yr); check("SpatialAttention dx", xd.grad, xr.grad)
    check("SpatialAttention dconv", sa.conv.weight.grad, sdr["conv.weight"].grad)


def test_refiner_d2_fixed_against_oracle():
    """graph.refiner.Refiner with the D2 fix (layer2 takes 2 channels).  Parity UNPINNED by the
    reference (it raises there); checked against the oracle's restatement of the same intent."""
    from graph.refiner import Refiner
    sd = W.make_state_dict(W.manifest_refiner(), 0, "wc")
    m = Refiner(); m.load_state_dict(sd); m = m.to(dev).train()
    torch.manual_seed(11)            # (inputs used to depend on what ran before this test)
    x = torch.rand(3, 1, 96, 60)
    osd = {k: (v.clone().double().requires_grad_(True) if v.is_floating_point() and "running" not in k else
               (v.clone().double() if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    yr = R.refiner(osd, "", xr, train=True)
    dy = torch.randn_like(yr); yr.backward(dy)
    xd = x.to(dev).requires_grad_(True)
    y = m(xd)
    check("Refiner fwd", y, yr)
    y.backward(dy.float().to(dev))
    check_grad("Refiner dx", xd.grad, xr.grad)
    gscale = max(v.grad.abs().max().item() for v in osd.values() if getattr(v, "grad", None) is not None)
    for n, p in m.named_parameters():
        # atol: the conv biases in front of a BatchNorm have an analytically zero gradient; what the kernels return is
        # the fp32 rounding of a 17 280-term sum that cancels (observed up to 1.3e-6 of the largest gradient)
        check_grad("Refiner d" + n, p.grad, osd[n].grad, atol=5e-6 * gscale)
    for k, v in m.state_dict().items():
        if "running" in k:
            check("Refiner " + k, v, osd[k])
    # and inside the generator wrapper (graph/model.py:22-41 with use_refiner=True)
    from graph.model import Model
    g = Model(use_refiner=True).to(dev).eval()
    note, pre, phrase, pos = W.make_inputs(2, seed=5)
    with torch.no_grad():
        out = g(torch.randn(2, 1152, device=dev), pre.to(dev), phrase.to(dev), pos.to(dev), False)
    assert tuple(out.shape) == (2, 1, 96, 60) and torch.isfinite(out).all()


def test_variational_encoder_flag():
    """Encoder(variational=True): mean / log-variance heads + fused reparameterise + KL
    (old/graphs/models/bar_v1/encoder.py:60-63, old/graphs/losses/loss.py:14-17); default stays off."""
    from graph.encoder import Encoder
    enc0 = Encoder([64, 128, 256, 512, 1024])
    assert "var.weight" not in enc0.state_dict()
    enc = Encoder([64, 128, 256, 512, 1024], variational=True).to(dev).train()
    with torch.no_grad():
        enc.var.weight.mul_(1e-3)         # keep exp(logvar) finite under the N(-1,1) init
        enc.linear.weight.mul_(1e-3)
    x = (torch.rand(2, 1, 96, 60) < 0.05).float().to(dev)
    z = enc(x)
    kl = enc.last_kl
    assert tuple(z.shape) == (2, 1152) and torch.isfinite(z).all() and torch.isfinite(kl) and kl.item() >= 0
    (z.sum() * 1e-3 + kl * 1e-3).backward()
    assert enc.var.weight.grad.abs().max().item() > 0 and enc.linear.weight.grad.abs().max().item() > 0
    enc.eval()
    with torch.no_grad():
        # eval returns the mean (no sampling); split-K atomics / autotuned configurations make the last
        # bits run-dependent and the N(-1,1)-initialised trunk amplifies them to ~1e-4 relative
        assert torch.allclose(enc(x), enc(x), rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("geom", [(3, 8, 10, 7, 16, 3, 1, 1), (2, 64, 12, 8, 128, 3, 2, 1), (4, 40, 1, 1, 24, 1, 1, 0),
                                  (2, 1, 20, 12, 8, 3, 1, 1), (2, 256, 6, 4, 512, 3, 1, 1)])
def test_c_abi_channel_offsets(geom):
    """x_coff / y_coff of MgvaeConvDesc (the host layer always passes 0 and offsets the pointer instead): all three
    conv entry points must address [N, ctot, H, W] buffers at a channel offset, tiled, split-K, thin and skinny."""
    import ctypes
    from hipops import _native as nat
    L = nat.lib()
    N, Cx, H, W_, Cy, k, st, pd = geom
    OH, OW = (H + 2 * pd - k) // st + 1, (W_ + 2 * pd - k) // st + 1
    xc, xo, yc, yo = Cx + 5, 3, Cy + 7, 4
    xb = torch.randn(N, xc, H, W_); yb = torch.randn(N, yc, OH, OW); w = torch.randn(Cy, Cx, k, k) * 0.1
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = nat.ConvDesc(N, Cx, H, W_, Cy, OH, OW, k, k, st, st, pd, pd, xc, xo, yc, yo, 0, 0.0)
    xs, ys = xb[:, xo:xo + Cx].double(), yb[:, yo:yo + Cy].double()
    # forward into the y slice
    xd, yd, wd = xb.to(dev), yb.to(dev), w.to(dev)
    assert L.mgvae_conv2d_fwd(ctypes.byref(d), vp(xd), vp(wd), None, vp(yd), s) == 0
    want = yb.double().clone(); want[:, yo:yo + Cy] = F.conv2d(xs, w.double(), None, st, pd)
    check("abi coff fwd %s" % (geom,), yd, want)
    # data gradient into the x slice
    xd2 = xb.to(dev); yd2 = yb.to(dev)
    assert L.mgvae_conv2d_bwd_data(ctypes.byref(d), vp(yd2), vp(wd), None, vp(xd2), s) == 0
    wantx = xb.double().clone()
    wantx[:, xo:xo + Cx] = torch.nn.grad.conv2d_input(xs.shape, w.double(), ys, st, pd)
    check("abi coff bwd_data %s" % (geom,), xd2, wantx)
    # weight gradient (accumulates)
    dw = torch.ones_like(wd)
    assert L.mgvae_conv2d_bwd_weight(ctypes.byref(d), vp(xb.to(dev)), vp(yb.to(dev)), vp(dw), s) == 0
    check("abi coff bwd_weight %s" % (geom,), dw, 1.0 + torch.nn.grad.conv2d_weight(xs, w.shape, ys, st, pd))


@pytest.mark.parametrize("geom", [(3, 16, 10, 7, 24, (3, 3), (1, 1), (1, 1)), (2, 32, 24, 10, 32, (1, 4), (1, 2), (0, 1)),
                                  (4, 512, 6, 4, 512, (3, 3), (1, 1), (1, 1)), (2, 20, 9, 9, 12, (3, 3), (2, 2), (1, 1))])
@pytest.mark.parametrize("act", [1, 2])
def test_deferred_activation_gradient(geom, act):
    """conv1 -> act -> conv2 with the activation's gradient folded into conv2's data gradient (masked epilogue,
    incl. split-K and strided phases) equals the unfused chain and the fp64 reference."""
    N, C, H, W_, Co, k, s, p = geom
    hf = HF()
    x = torch.randn(N, C, H, W_); w1 = torch.randn(C, C, 3, 3) * 0.2; w2 = torch.randn(Co, C, *k) * 0.2
    fn = (None, F.relu, lambda t: F.leaky_relu(t, 0.01))[act]
    xr, w1r, w2r = (t.double().requires_grad_(True) for t in (x, w1, w2))
    yr = F.conv2d(fn(F.conv2d(xr, w1r, None, 1, 1)), w2r, None, s, p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = x.to(dev).requires_grad_(True); w1d = torch.nn.Parameter(w1.to(dev)); w2d = torch.nn.Parameter(w2.to(dev))
    h = hf.conv2d(xd, w1d, None, (1, 1), (1, 1), act, 0.01, defer_act_grad=True)
    y = hf.conv2d(h, w2d, None, s, p, in_act=(act, 0.01))
    tag = "deferred act grad %s act%d" % (geom, act)
    check(tag + " fwd", y, yr)
    y.backward(dy.float().to(dev))
    check_grad(tag + " dx", xd.grad, xr.grad)
    check(tag + " dw2", w2d.grad, w2r.grad)
    check_grad(tag + " dw1", w1d.grad, w1r.grad)


BF16_GEOMS = [
    (3, 64, 48, 30, 64, (3, 3), (1, 1), (1, 1)),      # residual 64 (128x128 / 64-wide tiles via the autotuner)
    (2, 256, 12, 8, 256, (3, 3), (1, 1), (1, 1)),
    (5, 128, 24, 15, 256, (3, 3), (2, 2), (1, 1)),    # strided, odd 15 -> 8
    (4, 512, 6, 4, 1024, (3, 3), (2, 2), (1, 1)),     # deep K, split-K
    (3, 2048, 6, 3, 1024, (1, 1), (1, 1), (0, 0)),
    (2, 32, 48, 60, 32, (1, 4), (1, 2), (0, 1)),      # stem second conv
    (2, 17, 9, 7, 70, (3, 3), (2, 2), (1, 1)),        # ragged channels / odd K
    (2, 128, 24, 15, 64, (4, 4), (2, 2), (1, 1)),     # 4x4 s2
]


@pytest.mark.parametrize("g", BF16_GEOMS, ids=lambda g: "x".join(map(str, g[:5])) + "k%dx%d" % g[5])
def test_conv2d_bf16_compute(g):
    """the NCHW kernels' bf16-operand mode (set_nchw_operand_dtype; no model path uses it): operands rounded to bf16 (RNE), fp32 accumulation.  The reference rounds the
    same operands to bf16 and multiplies in fp64, so what remains is fp32 accumulation error: the fp32 tolerance."""
    N, Ci, H, W_, Co, k, s, p = g
    hf = HF()
    bf = lambda t: t.bfloat16().double()
    x = torch.randn(N, Ci, H, W_).relu_(); w = torch.randn(Co, Ci, *k) * 0.2 - 0.1
    xr, wr = bf(x).requires_grad_(True), bf(w).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=s, padding=p)
    dy = torch.randn_like(yr).bfloat16().double()          # exactly representable: both gradients see the same dy
    want_dx = torch.nn.grad.conv2d_input(xr.shape, wr.detach(), dy, s, p)
    want_dw = torch.nn.grad.conv2d_weight(xr.detach(), wr.shape, dy, s, p)
    hf.set_nchw_operand_dtype("bf16")
    try:
        xd = x.to(dev).requires_grad_(True); wd = torch.nn.Parameter(w.to(dev))
        y = hf.conv2d(xd, wd, None, s, p)
        check("bf16 conv fwd %s" % (g,), y, yr)
        y.backward(dy.float().to(dev))
        check("bf16 conv dx %s" % (g,), xd.grad, want_dx)
        check("bf16 conv dw %s" % (g,), wd.grad, want_dw)
        # transposed conv rides the same kernels with the roles swapped
        wt = torch.randn(Ci, Co, *k) * 0.2
        zr = F.conv_transpose2d(xr.detach(), bf(wt), None, stride=s, padding=p)
        z = hf.conv_transpose2d(x.to(dev), wt.to(dev), None, s, p)
        check("bf16 convT fwd %s" % (g,), z, zr)
    finally:
        hf.set_nchw_operand_dtype("f32")
    assert hf.get_nchw_operand_dtype() == "f32"


def test_train_step_bf16_close_to_fp32():
    """one generator pre-training step with the bf16-storage island against the same step in fp32 (well-conditioned
    weights): a sanity bound, not the parity test (that is test_bf16_storage_step_against_bf16_rounding_oracle): loss
    within 3 %; the flat gradient points the same way (cosine > 0.93, relative L2 < 0.4: every stored activation carries
    2^-9 relative rounding, twice per layer, and ReLU / arg-max decisions near ties flip through ~40 layers)."""
    import __graft_entry__ as ge
    ge.build()
    from graph.model import Model
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.loss.bar_loss import Loss, DLoss
    from hipops.train import PretrainStep
    hf = HF()
    B = 4
    g = torch.Generator().manual_seed(5)
    note = (torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev)
    pre = (torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev)
    phrase = (torch.rand(B, 1, 384, 60, generator=g) < 0.05).float().to(dev)
    pos = torch.randint(0, 332, (B,), generator=g).to(dev)
    grads = {}
    losses = {}
    for mode in ("f32", "bf16"):
        torch.manual_seed(0)
        gen, zb, zp = Model().to(dev), BarZDiscriminator().to(dev), PhraseZDiscriminator().to(dev)
        with torch.no_grad():
            for m in (gen, zb, zp):
                for name, prm in m.named_parameters():      # well-conditioned: N(0, 1/fan_in)-like instead of D4's N(-1,1)
                    if prm.dim() > 1:
                        fan = prm[0].numel()
                        prm.copy_(torch.randn(prm.shape, generator=torch.Generator().manual_seed(__import__("zlib").crc32(name.encode()))) / fan ** 0.5)
                    else:
                        prm.copy_(torch.full(prm.shape, 0.5 if name.endswith("weight") else 0.0))
        gen.eval()                                           # no dropout: the two runs must see the same function
        hf.set_compute_dtype(mode)
        try:
            step = PretrainStep(gen, zb, zp, Loss().to(dev), DLoss().to(dev), lr=0.0)
            loss, _ = step(note, pre, phrase, pos)
            losses[mode] = float(loss.detach())
            grads[mode] = step.opt.grad.detach().double().cpu().clone()
        finally:
            hf.set_compute_dtype("f32")
    assert abs(losses["bf16"] - losses["f32"]) <= 3e-2 * abs(losses["f32"]), losses
    rel = float((grads["bf16"] - grads["f32"]).norm() / grads["f32"].norm())
    cos = float((grads["bf16"] * grads["f32"]).sum() / (grads["bf16"].norm() * grads["f32"].norm()))
    assert rel < 0.4 and cos > 0.93, (rel, cos)


def test_bf16_storage_step_against_bf16_rounding_oracle():
    """BASELINE.json configs 3-4 (bf16): one pre-training step at 32 bars with the channels-last island in bf16 STORAGE
    against the oracle run in fp64 with the SAME rounding model -- every island convolution sees bf16-rounded activations
    and bf16 copies of the fp32 master weights, stores a bf16 result, and receives a bf16-rounded gradient
    (oracle.restate.ISLAND_ROUNDING).  What that model does not cover (the bf16 stores of the fused norm/CBAM outputs are
    the next conv's rounding, but its residual branch and the gradients inside the fused backward are rounded once more) is
    the same size as what it covers, so the bar is relative: the HIP step must be as close to the rounding oracle as the
    rounding oracle is to exact arithmetic (x 1.5), tensor by tensor on the large tensors and on the flat gradient, and
    closer to the rounding oracle than to exact arithmetic's opposite side (cosine)."""
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.loss.bar_loss import Loss, DLoss
    from hipops import FlatParams
    hf = HF()
    mode, B = "wc", 32
    zsd = W.make_state_dict(W.manifest_z_discriminator(), 0, mode)
    note, pre, phrase, pos = W.make_inputs(B, seed=977)
    hf.set_compute_dtype("bf16")
    try:
        m, gsd = _generator(mode)
        zb, zp = BarZDiscriminator(), PhraseZDiscriminator()
        zb.load_state_dict(zsd); zp.load_state_dict(zsd)
        zb, zp = zb.to(dev), zp.to(dev)
        for d in (zb, zp):
            for prm in d.parameters():
                prm.requires_grad = False
        opt = FlatParams(list(m.parameters()), lr=0.002)
        opt.zero_grad()
        n_, p_, ph_, po_ = (t.to(dev) for t in (note, pre, phrase, pos))
        gen, z, pz, pf = m(n_, p_, ph_, po_)
        loss = DLoss.constant(zp(pf).view(-1), 1.0) + DLoss.constant(zb(z).view(-1), 1.0) + DLoss.constant(zb(pz).view(-1), 1.0)
        loss = loss + Loss().to(dev)(gen, n_, True)
        loss.backward()
        torch.cuda.synchronize()
        params = dict(m.named_parameters())
        hip = {n: p.grad.detach().double().cpu() for n, p in params.items() if p.grad is not None}
        hip_loss = float(loss.detach())
    finally:
        hf.set_compute_dtype("f32")
    names = [n for n in params if n in gsd]
    z64 = {k: v.double() for k, v in zsd.items()}

    def oracle(rounding):
        osd = {k: v.clone().double().requires_grad_(True) for k, v in gsd.items()}
        R.ISLAND_ROUNDING = rounding
        try:
            lo, _ = R.pretrain_step_loss(osd, z64, z64, note.double(), pre.double(), phrase.double(), pos, True)
            og = torch.autograd.grad(lo, [osd[n] for n in names], allow_unused=True)
        finally:
            R.ISLAND_ROUNDING = None
        return float(lo.detach()), dict(zip(names, og))

    lo_r, g_r = oracle((RoundBf16.apply, RoundBf16Forward.apply))
    lo_x, g_x = oracle(None)
    REPORT.append("bf16 step (32 bars): loss hip %.6f  rounding oracle %.6f  exact %.6f" % (hip_loss, lo_r, lo_x))
    # The HIP loss itself is reproducible only to ~1 % at this size: measured on one box, same binary, four runs in a row:
    # 5.6419, 5.6767, 5.6817, 5.6668 (per-op path and chained path alike; another box: 5.7119) around the rounding
    # oracle's 5.6367 and exact arithmetic's 5.6666 -- the statistics kernels and the NCHW stems' split-K sum with fp32
    # atomics, and in bf16 STORAGE that last-bit noise moves bf16 roundings, which 40 layers amplify.  The bound is therefore
    # the model's own distance from exact arithmetic (x 2) or 2.5 %, whichever is larger; the segmented test
    # (tests/test_gan_parity_gpu.py::test_bf16_generator_step_segmented_against_rounding_oracle) is the tight one.
    assert abs(hip_loss - lo_r) <= max(2 * abs(lo_r - lo_x), 2.5e-2 * abs(lo_x)), (hip_loss, lo_r, lo_x)
    keep = [n for n in names if g_x[n] is not None and n in hip]
    flat = lambda d: torch.cat([d[n].reshape(-1).double() for n in keep])
    fh, fr, fx = flat(hip), flat(g_r), flat(g_x)
    e_hip = float((fh - fr).norm() / fr.norm()); e_model = float((fr - fx).norm() / fx.norm()); e_exact = float((fh - fx).norm() / fx.norm())
    cos = float((fh * fr).sum() / (fh.norm() * fr.norm()))
    REPORT.append("bf16 step: flat gradient  |hip - rounding oracle| %.3e   |rounding oracle - exact| %.3e   |hip - exact| %.3e   cos(hip, rounding oracle) %.5f"
                  % (e_hip, e_model, e_exact, cos))
    worst = []
    for n in keep:
        if g_x[n].numel() < 4096:
            continue                                   # small aggregates: covered by the flat vector
        a = float((hip[n] - g_r[n]).norm() / g_r[n].norm().clamp_min(1e-300))
        b = float((g_r[n] - g_x[n]).norm() / g_x[n].norm().clamp_min(1e-300))
        worst.append((a / max(b, 1e-3), n, a, b))
    worst.sort(reverse=True)
    for ratio, n, a, b in worst[:8]:
        REPORT.append("    %-60s |hip - rounding oracle| %.3e   |rounding oracle - exact| %.3e   ratio %.2f" % (n, a, b, ratio))
    assert e_hip <= 1.5 * max(e_model, 2e-3), (e_hip, e_model)
    assert cos >= 1.0 - 2.0 * max(e_model, 2e-3) ** 2, cos
    assert worst[0][0] <= 3.0, worst[0]


def test_graphed_train_step_matches_eager():
    """the whole step replayed as one HIP graph == the eager step (same launches, same order): parameters after the
    same number of Adam steps agree to split-K summation noise; Adam's bias correction and lr follow the host-side
    step counter through the pinned hyper-parameter vector; dropout masks are refreshed between replays."""
    import __graft_entry__ as ge
    ge.build()
    from graph.model import Model
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.loss.bar_loss import Loss, DLoss
    from hipops.train import PretrainStep, GraphedPretrainStep
    hf = HF()
    B = 2
    g = torch.Generator().manual_seed(9)
    batch = [(torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev), (torch.rand(B, 1, 96, 60, generator=g) < 0.05).float().to(dev),
             (torch.rand(B, 1, 384, 60, generator=g) < 0.05).float().to(dev), torch.randint(0, 332, (B,), generator=g).to(dev)]

    def build():
        torch.manual_seed(0)
        gen, zb, zp = Model().to(dev), BarZDiscriminator().to(dev), PhraseZDiscriminator().to(dev)
        with torch.no_grad():
            for m in (gen, zb, zp):
                for prm in m.parameters():
                    prm.mul_(0.02)
        gen.eval()                      # dropout off: both runs compute the same function
        for d in (zb, zp):
            for prm in d.parameters():
                prm.requires_grad = False
        return PretrainStep(gen, zb, zp, Loss().to(dev), DLoss().to(dev), lr=1e-3)

    eager = build()
    for _ in range(5):
        le, _ = eager(*batch)
    graphed_base = build()
    w_before = graphed_base.opt.flat.clone()
    gs = GraphedPretrainStep(graphed_base, *batch, warmup=3)
    # construction trains nothing: the warm-up steps are undone (weights, Adam moments, step counter)
    assert torch.equal(graphed_base.opt.flat, w_before) and graphed_base.opt.step_count == 0
    assert float(graphed_base.opt.exp_avg.abs().max()) == 0.0

    def eager_forward_is_current(step):
        """an eager forward of the graphed model == a forward of a fresh model holding the same flat weights (the
        matrix-pipe copies of the conv weights must not be the warm-up's or the previous replay's)"""
        fresh = build()
        with torch.no_grad():
            fresh.opt.flat.copy_(step.opt.flat)
            a = step.gen(*batch)[0]
            b = fresh.gen(*batch)[0]
        torch.cuda.synchronize()
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max())), float((a - b).abs().max())

    eager_forward_is_current(graphed_base)
    for _ in range(5):
        lg, _ = gs(*batch)
    eager_forward_is_current(graphed_base)
    lg, _ = gs(*batch); lg, _ = gs(*batch)         # and replays after an eager forward still train the same weights
    for _ in range(2):
        le, _ = eager(*batch)
    assert graphed_base.opt.step_count == eager.opt.step_count == 7
    a, b = eager.opt.flat.double().cpu(), graphed_base.opt.flat.double().cpu()
    # Adam normalises every gradient to ~+-lr per step, so parameters whose gradient is pure summation noise (atomics
    # order) may move by up to 2*lr*steps apart: compare in L2 over the whole vector, and through the loss
    assert float((a - b).norm() / a.norm()) < 2e-2, float((a - b).norm() / a.norm())
    assert float((a - b).abs().max()) <= 2 * 1e-3 * 7 + 1e-6
    # (1e-3: with weights scaled to ~1e-3 and lr 1e-3, five Adam steps amplify the atomics-order noise of the split-K
    # weight gradients; observed 1e-5 .. 3e-4 from run to run, depending on the split counts the autotuner picked)
    assert abs(float(lg) - float(le)) <= 1e-3 * abs(float(le)), (float(lg), float(le))
    # training mode: the two dropout masks change between replays
    graphed_base.gen.train()
    gs2 = GraphedPretrainStep(graphed_base, *batch, warmup=1)
    gs2(*batch); m0 = gs2.masks[0].clone()
    gs2(*batch)
    assert not torch.equal(m0, gs2.masks[0]) and set(gs2.masks[0].unique().tolist()) <= {0.0, 1.0 / 0.7} | {float(torch.tensor(1.0 / 0.7, dtype=torch.float32))}


@pytest.mark.parametrize("g", BF16_GEOMS, ids=lambda g: "x".join(map(str, g[:5])) + "k%dx%d" % g[5])
def test_conv2d_f32_via_bf16x3(g):
    """opt-in fp32-accurate mode on the bf16 matrix pipe (exact three-term operand split, six products): against the
    fp64 reference it must be as close as the native fp32-MFMA kernels are -- same tolerance, UNROUNDED operands."""
    N, Ci, H, W_, Co, k, s, p = g
    hf = HF()
    x = torch.randn(N, Ci, H, W_).relu_(); w = torch.randn(Co, Ci, *k) * 0.2 - 0.1
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=s, padding=p)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    errs = {}
    for mode in ("f32", "f32_bf16x3"):
        hf.set_nchw_operand_dtype(mode)
        try:
            xd = x.to(dev).requires_grad_(True); wd = torch.nn.Parameter(w.to(dev))
            y = hf.conv2d(xd, wd, None, s, p)
            y.backward(dy.float().to(dev))
            errs[mode] = [float((a.double().cpu() - b).abs().max() / b.abs().max()) for a, b in
                          ((y, yr.detach()), (xd.grad, xr.grad), (wd.grad, wr.grad))]
            if mode != "f32":
                check("f32 via bf16x3 fwd %s" % (g,), y, yr)
                check("f32 via bf16x3 dx %s" % (g,), xd.grad, xr.grad)
                check("f32 via bf16x3 dw %s" % (g,), wd.grad, wr.grad)
        finally:
            hf.set_nchw_operand_dtype("f32")
    # fp32 accuracy, not bf16 accuracy: within a small factor of the native kernel's own error (both ~1e-7..1e-6)
    for e_split, e_native in zip(errs["f32_bf16x3"], errs["f32"]):
        assert e_split <= max(4 * e_native, 2e-6), (errs, g)
