"""GPU suite: the channels-last (NHWC) kernel family through the C ABI against torch fp64 on the same logical tensors
(torch.channels_last activations AND weights: logical shapes unchanged).  Geometries: every conv of the encoder trunks
(graph/encodingBlock.py:87-126) at both map heights, the decoder's transposed convs and 1x1 convs, channel-sliced
inputs / outputs, split-K, stride phases with odd extents (15 -> 8, 7 -> 15), bias, fused activation and the deferred
activation-gradient mask."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from parity_util import REPORT, check, flush_report

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


@pytest.mark.parametrize("shape", [(3, 64, 48, 30), (2, 128, 24, 15), (2, 256, 12, 8), (3, 512, 6, 4), (2, 1024, 3, 2), (2, 64, 192, 30), (3, 32, 48, 30),
                                   (2, 32, 192, 30)])
@pytest.mark.parametrize("mode,act", [(1, 1), (2, 1), (1, 2), (0, 0)])
def test_norm_cbam_channels_last(shape, mode, act):
    """InstanceNorm -> CBAM -> (+residual) -> activation on channels-last tensors (u never materialised, channel
    pooling from the statistics pass) against the fp64 oracle: forward, dx, dres, affine and all CBAM weight gradients;
    output written into a channel slice of a wider channels-last buffer; negative and zero gammas included (max_hw(u)
    then sits at the minimum of x / at pixel 0)"""
    from hipops import functional as HF
    from oracle import restate as R
    N, C, H, W_ = shape
    x = torch.randn(shape) * 2 + 0.5; res = torch.randn(shape)
    g = torch.randn(C); b = torch.randn(C)
    g[3] = 0.0; g[5] = -abs(g[5]) - 0.1
    sd = {"channel_attention.conv1.weight": torch.randn(C // 16, C, 1, 1) * 0.2,
          "channel_attention.conv2.weight": torch.randn(C, C // 16, 1, 1) * 0.2,
          "spatial_attention.conv.weight": torch.randn(1, 2, 3, 3) * 0.3}
    sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    fn = (lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01))[act]
    ur = F.instance_norm(xr, None, None, gr, br, True, 0.01, 1e-5)
    o = R.cbam(sdr, "", ur)
    yr = o if mode == 0 else (fn(ur + o) if mode == 1 else fn(rr + o))
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = cl(x).requires_grad_(True); rd = cl(res).requires_grad_(True)
    gd = torch.nn.Parameter(g.to(dev)); bd = torch.nn.Parameter(b.to(dev))
    ps = {k: torch.nn.Parameter(v.to(dev)) for k, v in sd.items()}
    buf = cl(torch.full((N, C + 32, H, W_), 7.0))
    y = HF.norm_cbam_cl(xd, gd, bd, ps["channel_attention.conv1.weight"], ps["channel_attention.conv2.weight"],
                        ps["spatial_attention.conv.weight"], 1e-5, mode, rd if mode == 2 else None, act, 0.01, out=buf[:, 16:16 + C])
    tag = "norm_cbam_cl %s mode%d act%d" % (shape, mode, act)
    check(tag + " fwd", y, yr)
    assert float(buf[:, :16].min()) == 7.0 and float(buf[:, 16 + C:].max()) == 7.0          # neighbours untouched
    y.backward(cl(dy.float()))
    check(tag + " dx", xd.grad, xr.grad)
    check(tag + " dgamma", gd.grad, gr.grad); check(tag + " dbeta", bd.grad, br.grad)
    if mode == 2:
        check(tag + " dres", rd.grad, rr.grad)
    for k in sd:
        check(tag + " d" + k, ps[k].grad, sdr[k].grad)


def test_layout_round_trip_and_mean():
    from hipops import functional as HF
    x = torch.randn(3, 72, 9, 7)
    big = torch.randn(3, 100, 9, 7).to(dev)
    big[:, 10:82] = x.to(dev)
    xs = big[:, 10:82].detach().requires_grad_(True)                # an NCHW channel slice
    y = HF.to_channels_last(xs)
    assert HF.cl_pitch(y) == 72 and torch.equal(y.cpu(), x)
    z = HF.to_nchw(y)
    assert z.is_contiguous() and torch.equal(z.cpu(), x)
    w = torch.randn(3, 72, 9, 7)
    (z * w.to(dev)).sum().backward()
    assert torch.equal(xs.grad.cpu(), w)
    t = cl(torch.randn(4, 1024, 3, 2)).requires_grad_(True)
    m = HF.global_avg_pool_cl(t)
    check("mean_nhwc fwd", m, t.detach().double().cpu().mean(dim=(2, 3)))
    dm = torch.randn(4, 1024)
    m.backward(dm.to(dev))
    check("mean_nhwc bwd", t.grad, (dm.double() / 6).view(4, 1024, 1, 1).expand(4, 1024, 3, 2))


@pytest.mark.parametrize("mode", ["wc", "d4"])
def test_channels_last_blocks_against_oracle(mode):
    """the encoder trunk's two block types on the channels-last kernels (conv -> [ReLU -> conv] -> fused norm / CBAM /
    residual) against the fp64 oracle, with channels-last weights inside a flat parameter buffer"""
    import graph.encodingBlock as EB
    import graph.decoder as DD
    from hipops import FlatParams
    from hipops import functional as HF
    from oracle import restate as R, weights as W
    from parity_util import check_grad, TOL
    gsd = W.make_state_dict(W.manifest_generator(), 0, mode)
    tol = TOL if mode == "wc" else 5e-3
    cases = [("cl.residual64", EB.ResidualModule(64, True), "encoder.layers.0.", R.residual_module, torch.randn(3, 64, 48, 30).relu_()),
             ("cl.pooling64", EB.PoolingModule(64, 128, True), "encoder.layers.1.", R.pooling_module, torch.randn(3, 64, 48, 30).relu_()),
             ("cl.residual512", EB.ResidualModule(512, True), "encoder.layers.6.", R.residual_module, torch.randn(3, 512, 6, 4).relu_()),
             ("cl.pooling512", EB.PoolingModule(512, 1024, True), "encoder.layers.7.", R.pooling_module, torch.randn(3, 512, 6, 4).relu_()),
             ("cl.residual128 odd", EB.ResidualModule(128, True), "encoder.layers.2.", R.residual_module, torch.randn(2, 128, 24, 15).relu_()),
             ("cl.deconv_pp1024", DD.DeConvPitchPadding(1024, 512, True), "decoder.layers.0.", R.deconv_pitch_padding, torch.randn(3, 1024, 6, 3).relu_()),
             ("cl.deconv_pp512", DD.DeConvPitchPadding(512, 256, True), "decoder.layers.1.", R.deconv_pitch_padding, torch.randn(3, 512, 12, 7).relu_()),
             ("cl.deconv256", DD.DeConvModule(256, 128, True), "decoder.layers.2.", R.deconv_module, torch.randn(2, 256, 24, 15).relu_()),
             ("cl.deconv128", DD.DeConvModule(128, 64, True), "decoder.layers.3.", R.deconv_module, torch.randn(2, 128, 48, 30).relu_())]
    for tag, mod, prefix, ofn, x in cases:
        sub = {k[len(prefix):]: v for k, v in gsd.items() if k.startswith(prefix)}
        mod.load_state_dict(sub)
        mod = mod.to(dev)
        wkey = "conv1.weight" if "residual" in tag else ("deConv1.weight" if "deconv" in tag else "conv.weight")
        assert mod.state_dict()[wkey].stride(1) == 1                                                       # stored channels-last
        opt = FlatParams(list(mod.parameters()))
        opt.zero_grad()
        osd = {k: v.clone().double().requires_grad_(True) for k, v in sub.items()}
        o32 = {k: v.clone().requires_grad_(True) for k, v in sub.items()}
        xr = x.double().requires_grad_(True)
        yr = ofn(osd, "", xr)
        dy = torch.randn_like(yr)
        yr.backward(dy)
        x32 = x.clone().requires_grad_(True)
        ofn(o32, "", x32).backward(dy.float())
        xd = x.to(dev).requires_grad_(True)
        y = mod(HF.to_channels_last(xd))
        check("%s[%s] fwd" % (tag, mode), y, yr, tol)
        y.backward(dy.float().to(dev))
        check_grad("%s[%s] dx" % (tag, mode), xd.grad, xr.grad, tol, ref32=x32.grad)
        gscale = max(v.grad.abs().max().item() for v in osd.values() if v.grad is not None)
        for n, p in mod.named_parameters():
            if osd[n].grad is None:                                   # reference defect D5: bn1 of DeConvPitchPadding is never used
                assert p.grad.abs().max().item() == 0, n
                continue
            check_grad("%s[%s] d%s" % (tag, mode, n), p.grad, osd[n].grad, tol, atol=1e-6 * gscale, ref32=o32[n].grad)
            assert p.grad.data_ptr() >= opt.grad.data_ptr()          # accumulated straight into the flat gradient


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_chained_blocks_equal_the_per_op_path(storage):
    """hipops/blocks.py (one autograd node + one mgvae_chain_run per block and direction) against the per-op autograd
    functions it replaces, on the same inputs: the chain issues the SAME entry points in the same order, so the forward is
    bit-identical and the gradients agree to the weight gradients' atomic summation order -- for every chained block type,
    in fp32 (x3 engine) and bf16 storage, with the gradient flowing into the input (need_dx) and not."""
    import graph.encodingBlock as EB
    import graph.decoder as DD
    from graph.cbam import CBAM
    from graph.layers import Conv2d, InstanceNorm2d
    from hipops import FlatParams
    from hipops import blocks as HB
    from hipops import functional as HF

    class Fit(torch.nn.Module):                 # the decoder's fit1 stage: 1x1 conv -> InstanceNorm -> +CBAM -> ReLU
        def __init__(self):
            super().__init__()
            self.fit1 = Conv2d(256, 128, 1, stride=1, bias=False, channels_last=True)
            self.bn = InstanceNorm2d(128)
            self.cbam = CBAM(128)

        def forward(self, o):
            if HB.usable(o):
                return HB.conv_norm_cbam_block(o, self.fit1, self.bn, self.cbam)
            return self.cbam.fused_norm(self.fit1(o), self.bn, 1, act=HF.ACT_RELU, channels_last=True)

    cases = [("residual64", lambda: EB.ResidualModule(64, True), (3, 64, 24, 30)),
             ("residual512", lambda: EB.ResidualModule(512, True), (4, 512, 6, 4)),
             ("pooling64", lambda: EB.PoolingModule(64, 128, True), (3, 64, 48, 30)),
             ("pooling odd", lambda: EB.PoolingModule(128, 256, True), (2, 128, 24, 15)),
             ("fit", Fit, (3, 256, 6, 3)),
             ("deconv_pp", lambda: DD.DeConvPitchPadding(512, 256, True), (3, 512, 12, 7)),
             ("deconv", lambda: DD.DeConvModule(128, 64, True), (2, 128, 24, 30))]
    HF.set_compute_dtype(storage)
    fork_min = HF.FORK_MIN_BATCH
    HF.FORK_MIN_BATCH = 1           # fork the weight gradients onto the side stream even at these small batches
    try:
        for tag, mk, shape in cases:
            torch.manual_seed(11)
            mod = mk().to(dev)
            with torch.no_grad():
                for prm in mod.parameters():
                    if prm.dim() > 1:
                        prm.mul_(0.05).add_(0.01 * torch.randn_like(prm))      # tame the N(-1,1) init: informative outputs
            opt = FlatParams(list(mod.parameters()))
            x = torch.randn(shape).relu_()
            dy = None
            res = {}
            for need_dx in (True, False):
                for chained in (False, True):
                    HB.ENABLED = chained
                    opt.zero_grad()
                    xd = x.to(dev).requires_grad_(need_dx)
                    xi = HF.to_channels_last(xd) if need_dx else HF.to_channels_last(xd).detach()
                    y = mod(xi)
                    if dy is None:
                        dy = torch.randn(y.shape, device=dev)
                    y.backward(dy.to(y.dtype) if y.dtype != dy.dtype else dy)
                    torch.cuda.synchronize()
                    res[(need_dx, chained)] = (y.detach().float().clone(), xd.grad.clone() if need_dx else None, opt.grad.clone())
                    fn = type(y.grad_fn).__name__
                    assert ("Residual" in fn or "ConvNormCbam" in fn or "DeConv" in fn) == chained, (tag, chained, fn)
                (y0, dx0, g0), (y1, dx1, g1) = res[(need_dx, False)], res[(need_dx, True)]
                assert torch.equal(y0, y1), "%s [%s]: chained forward differs from the per-op forward" % (tag, storage)
                gs = float(g0.abs().max())
                assert gs > 0
                tol = 1e-5 if storage == "f32" else 2e-2      # bf16: dx of a two-consumer input is summed in bf16 (one more rounding)
                # bf16 storage: the InstanceNorm backward's atomically summed statistics move the last bit of a few bf16-stored
                # gradient values from run to run, and an analytically ZERO weight-like gradient (the bias of a conv in front
                # of an InstanceNorm) is the sum of exactly that noise: measured 3e-4 of the largest gradient entry
                gtol = 1e-5 if storage == "f32" else 2e-3
                assert float((g0 - g1).abs().max()) <= gtol * gs, (tag, storage, need_dx, float((g0 - g1).abs().max()) / gs)
                if need_dx:
                    assert float((dx0 - dx1).abs().max()) <= tol * float(dx0.abs().max()), (tag, storage, float((dx0 - dx1).abs().max()))
            # a frozen block (the generator inside a discriminator step) still back-propagates to its input
            HB.ENABLED = True
            for prm in mod.parameters():
                prm.requires_grad = False
            opt.zero_grad()
            xd = x.to(dev).requires_grad_(True)
            mod(HF.to_channels_last(xd)).backward(dy.to(HF.island_dtype()))
            torch.cuda.synchronize()
            assert float(opt.grad.abs().max()) == 0.0
            assert float((xd.grad - res[(True, True)][1]).abs().max()) <= 1e-5 * float(xd.grad.abs().max()) + (2e-2 * float(xd.grad.abs().max()) if storage == "bf16" else 0)
    finally:
        HB.ENABLED = True
        HF.FORK_MIN_BATCH = fork_min
        HF.set_compute_dtype("f32")


def test_chained_ends_equal_the_per_op_path():
    """the chains outside the island -- the latent / feature discriminators' MLPs (hipops.blocks.mlp) and an encoder trunk's
    entry (both stems + the layout change, hipops.blocks.trunk_entry) -- against the per-op path on the same inputs: same
    entry points in the same order, so forward values are bit-identical and gradients agree to atomic summation order;
    trainable and frozen (the generator step back-propagates THROUGH frozen discriminators), dense and column-sliced inputs."""
    from graph.encoder import Encoder
    from graph.z_discriminator import BarZDiscriminator
    from graph.bar_discriminator_with_feature import BarFeatureDiscriminator
    from hipops import FlatParams
    from hipops import blocks as HB
    from hipops import functional as HF
    fork_min = HF.FORK_MIN_BATCH
    HF.FORK_MIN_BATCH = 1
    try:
        for name, mk in (("zdisc", BarZDiscriminator), ("feature", BarFeatureDiscriminator)):
            torch.manual_seed(5)
            m = mk().to(dev)
            with torch.no_grad():           # (the reference's N(-1, 1) init leaves every ReLU dead: nothing to compare)
                for prm in m.parameters():
                    prm.copy_(torch.randn_like(prm) * (2.0 / prm.shape[-1] ** 0.5 if prm.dim() > 1 else 0.1))
            opt = FlatParams(list(m.parameters()))
            wide = torch.randn(6, 2304)
            for sliced in (False, True):
                for frozen in (False, True):
                    for prm in m.parameters():
                        prm.requires_grad = not frozen
                    res = {}
                    for chained in (False, True):
                        HB.ENABLED = chained
                        opt.zero_grad()
                        wd = wide.to(dev).requires_grad_(True)
                        x = wd[:, 1152:] if sliced else wd[:, :1152].contiguous()
                        y = m(x)
                        assert ("Mlp" in type(y.grad_fn).__name__) == chained
                        (y * torch.linspace(-1, 2, y.numel(), device=dev).view_as(y)).sum().backward()
                        torch.cuda.synchronize()
                        res[chained] = (y.detach().clone(), wd.grad.clone(), opt.grad.clone())
                    assert float((res[False][0] - res[True][0]).abs().max()) <= 1e-6, (name, sliced, frozen)
                    for a, b in ((res[False][1], res[True][1]), (res[False][2], res[True][2])):
                        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-30), (name, sliced, frozen)
                    assert (float(res[True][2].abs().max()) == 0.0) == frozen
                    assert float(res[True][1].abs().max()) > 0
        # the decoder's front (head + stems + layout change, hipops.blocks.decoder_front) against the same three per-op calls of
        # Decoder.forward: Philox dropout (same seed and offsets -> same masks), injected masks, eval mode; a fixed cotangent on the
        # front's output, the gradients of z, pre_z, the phrase feature and the parameters.  (The comparison stops at the front's
        # output on purpose: through the four up-sampling blocks with random weights the per-op path differs from ITSELF by 1e-2
        # in fp32 and 30 % in bf16 storage from one run to the next -- tools/r3_front_noise.py -- which says nothing about the
        # chain; the step and agent tests hold the whole decoder to the fp64 oracle with trained-scale weights.)
        from graph.decoder import Decoder
        for storage in ("f32", "bf16"):
            HF.set_compute_dtype(storage)
            for how in ("rng", "masks", "eval"):
                res = {}
                for run in ("per-op", "per-op again", "chained"):
                    HB.ENABLED = run == "chained"
                    torch.manual_seed(8)
                    dec = Decoder([1024, 512, 256, 128, 64]).to(dev)
                    with torch.no_grad():
                        for prm in dec.parameters():
                            if prm.dim() > 1:
                                prm.copy_(torch.randn_like(prm) * (1.2 / (prm[0].numel() ** 0.5)))
                    dec.train(how != "eval")
                    gm = torch.Generator().manual_seed(4)
                    dec._drop_masks = [((torch.rand(3, 1152, generator=gm) >= 0.3).float() / 0.7).to(dev) for _ in range(2)] if how == "masks" else None
                    HF.manual_seed(77)
                    opt = FlatParams(list(dec.parameters()))
                    opt.zero_grad()
                    gi = torch.Generator().manual_seed(5)
                    zz = torch.randn(6, 1152, generator=gi).to(dev).requires_grad_(True)
                    pf = torch.randn(3, 1152, generator=gi).to(dev).requires_grad_(True)
                    pos = torch.tensor([3, 330, 17], device=dev)
                    if run == "chained":
                        assert HB.front_usable(zz[:3])
                        o = HB.decoder_front(dec, zz[:3], zz[3:], pf, pos)
                        assert "DecoderFront" in type(o.grad_fn).__name__
                    else:
                        o = HF.to_channels_last(dec.stems(dec.head(zz[:3], zz[3:], pf, pos)))
                    assert tuple(o.shape) == (3, 2048, 6, 3) and o.dtype == (torch.bfloat16 if storage == "bf16" else torch.float32)
                    ct = torch.linspace(-1, 1, o.numel(), device=dev).view(3, 6, 3, 2048).permute(0, 3, 1, 2).to(o.dtype)
                    o.backward(ct)
                    torch.cuda.synchronize()
                    res[run] = (o.detach().float().clone(), zz.grad.clone(), pf.grad.clone(), opt.grad.clone())
                # The per-op path against ITSELF gives the noise of this comparison (the stems' GEMMs split K with fp32 atomics and
                # the InstanceNorm backward over an 18-pixel map divides by standard deviations that random weights make small);
                # the chained path must sit inside 4 x that noise -- a wrong call sequence is off by O(1).
                for i, name in enumerate(("front", "d(z, pre_z)", "d(phrase feature)", "parameter gradients")):
                    a, a2, b = res["per-op"][i], res["per-op again"][i], res["chained"][i]
                    scale = float(a.abs().max())
                    assert scale > 0, (storage, how, name)
                    noise = float((a - a2).abs().max()) / scale
                    floor = 8e-3 if (storage == "bf16" and i == 0) else 1e-5      # one bf16 ulp of the stored front
                    tol = max(floor, 4 * noise)
                    assert noise < 2e-2 and float((a - b).abs().max()) <= tol * scale, (storage, how, name, float((a - b).abs().max()) / scale, noise)
        HF.set_compute_dtype("f32")
        # the bar-pair discriminator (hipops/netchain.py): train / eval BatchNorm, trainable / frozen, gradient into the pair or not
        from graph.bar_discriminator import BarDiscriminator
        for training in (True, False):
            for frozen in (False, True):
                for need_dx in (False, True):
                    if frozen and not need_dx:
                        continue
                    res = {}
                    for chained in (False, True):
                        HB.ENABLED = chained
                        torch.manual_seed(7)
                        m = BarDiscriminator().to(dev)
                        with torch.no_grad():
                            for prm in m.parameters():
                                if prm.dim() > 1:
                                    prm.copy_(torch.randn_like(prm) * (1.5 / (prm[0].numel() ** 0.5)))
                            for mod in m.modules():
                                if type(mod).__name__ == "BatchNorm2d":
                                    mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
                        m.train(training)
                        for prm in m.parameters():
                            prm.requires_grad = not frozen
                        opt = FlatParams(list(m.parameters()))
                        opt.zero_grad()
                        x = (torch.rand(5, 1, 192, 60, generator=torch.Generator().manual_seed(3)) < 0.1).float().to(dev).requires_grad_(need_dx)
                        y = m(x)
                        assert ("BarDisc" in type(y.grad_fn).__name__) == chained
                        (y * torch.linspace(-1, 2, y.numel(), device=dev).view_as(y)).sum().backward()
                        torch.cuda.synchronize()
                        sd = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
                        res[chained] = (y.detach().clone(), x.grad.clone() if need_dx else None, opt.grad.clone(), sd)
                    (y0, dx0, g0, sd0), (y1, dx1, g1, sd1) = res[False], res[True]
                    tag = ("bar discriminator", training, frozen, need_dx)
                    assert float((y0 - y1).abs().max()) <= 1e-6, tag
                    assert float((g0 - g1).abs().max()) <= 1e-5 * max(float(g0.abs().max()), 1e-30), tag
                    assert (float(g1.abs().max()) == 0.0) == frozen, tag
                    if need_dx:
                        assert float(dx0.abs().max()) > 0 and float((dx0 - dx1).abs().max()) <= 1e-5 * float(dx0.abs().max()), tag
                    for k in sd0:
                        assert float((sd0[k].double() - sd1[k].double()).abs().max()) <= 1e-6 * max(1.0, float(sd0[k].double().abs().max())), (tag, k)
                    assert int(sd1["chord.batch_norm1.num_batches_tracked"]) == (1 if training else 0)
                    assert int(sd1["basic.layers.2.bn1.num_batches_tracked"]) == 0
        for storage in ("f32", "bf16"):
            HF.set_compute_dtype(storage)
            torch.manual_seed(6)
            enc = Encoder([64, 128, 256, 512, 1024]).to(dev)
            with torch.no_grad():
                for prm in enc.parameters():
                    if prm.dim() > 1:
                        prm.mul_(0.05).add_(0.01 * torch.randn_like(prm))
            opt = FlatParams(list(enc.parameters()))
            x = (torch.rand(3, 1, 96, 60) < 0.1).float().to(dev)
            res = {}
            for chained in (False, True):
                HB.ENABLED = chained
                opt.zero_grad()
                z = enc(x)
                z.backward(torch.linspace(-1, 1, z.numel(), device=dev).view_as(z))
                torch.cuda.synchronize()
                res[chained] = (z.detach().clone(), opt.grad.clone())
            # (the NCHW stem convs may split K with fp32 atomics: forward values agree to summation order, not bit for bit;
            # in bf16 storage that noise can move a bf16 rounding somewhere along the trunk)
            ztol = (1e-5 if storage == "f32" else 2e-2) * float(res[False][0].abs().max())
            assert float((res[False][0] - res[True][0]).abs().max()) <= ztol, (storage, float((res[False][0] - res[True][0]).abs().max()))
            gs = float(res[False][1].abs().max())
            assert float((res[False][1] - res[True][1]).abs().max()) <= (1e-4 if storage == "f32" else 2e-2) * gs, (
                storage, float((res[False][1] - res[True][1]).abs().max()) / gs)
    finally:
        HB.ENABLED = True
        HF.FORK_MIN_BATCH = fork_min
        HF.set_compute_dtype("f32")


BF16_BLOCKS = ["residual64", "pooling64", "residual128", "residual512", "pooling512", "deconv_pp1024", "deconv_pp512", "deconv256", "deconv128"]


@pytest.mark.parametrize("tag", BF16_BLOCKS)
def test_channels_last_blocks_bf16_storage_against_rounding_oracle(tag):
    """every block type of the island in bf16 STORAGE (BASELINE.json configs 3-4) against the fp64 oracle under the
    island's rounding model (oracle.restate.ISLAND_ROUNDING: conv operands and results bf16, gradients bf16, weight
    gradients fp32).  The model explains the bf16 mode: the HIP block sits within 4e-3 of it forward (one final bf16
    store) and within a quarter of the distance between the rounding oracle and exact arithmetic (4 - 7 %) backward."""
    import graph.encodingBlock as EB
    import graph.decoder as DD
    from hipops import FlatParams
    from hipops import functional as HF
    from oracle import restate as R, weights as W
    from parity_util import RoundBf16, RoundBf16Forward
    mk, prefix, ofn, shape = {
        "residual64": (lambda: EB.ResidualModule(64, True), "encoder.layers.0.", R.residual_module, (3, 64, 48, 30)),
        "pooling64": (lambda: EB.PoolingModule(64, 128, True), "encoder.layers.1.", R.pooling_module, (3, 64, 48, 30)),
        "residual128": (lambda: EB.ResidualModule(128, True), "encoder.layers.2.", R.residual_module, (2, 128, 24, 15)),
        "residual512": (lambda: EB.ResidualModule(512, True), "encoder.layers.6.", R.residual_module, (3, 512, 6, 4)),
        "pooling512": (lambda: EB.PoolingModule(512, 1024, True), "encoder.layers.7.", R.pooling_module, (3, 512, 6, 4)),
        "deconv_pp1024": (lambda: DD.DeConvPitchPadding(1024, 512, True), "decoder.layers.0.", R.deconv_pitch_padding, (3, 1024, 6, 3)),
        "deconv_pp512": (lambda: DD.DeConvPitchPadding(512, 256, True), "decoder.layers.1.", R.deconv_pitch_padding, (3, 512, 12, 7)),
        "deconv256": (lambda: DD.DeConvModule(256, 128, True), "decoder.layers.2.", R.deconv_module, (2, 256, 24, 15)),
        "deconv128": (lambda: DD.DeConvModule(128, 64, True), "decoder.layers.3.", R.deconv_module, (2, 128, 48, 30))}[tag]
    gsd = W.make_state_dict(W.manifest_generator(), 0, "wc")
    l2 = lambda a, b: float((a.detach().double().cpu() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-300))
    mod = mk()
    sub = {k[len(prefix):]: v for k, v in gsd.items() if k.startswith(prefix)}
    mod.load_state_dict(sub)
    mod = mod.to(dev)
    opt = FlatParams(list(mod.parameters()))
    opt.zero_grad()
    x = torch.randn(shape).relu_().bfloat16().float()                # what the block receives is a stored bf16 tensor
    res = {}
    dy = None
    for name, rounding in (("round", (RoundBf16.apply, RoundBf16Forward.apply)), ("exact", None)):
        osd = {k: v.clone().double().requires_grad_(True) for k, v in sub.items()}
        xr = x.double().requires_grad_(True)
        R.ISLAND_ROUNDING = rounding
        try:
            yr = ofn(osd, "", xr)
            if dy is None:
                dy = torch.randn_like(yr).bfloat16().double()
            yr.backward(dy)
        finally:
            R.ISLAND_ROUNDING = None
        res[name] = (yr.detach(), xr.grad, {k: v.grad for k, v in osd.items()})
    HF.set_compute_dtype("bf16")
    try:
        xd = x.to(dev).requires_grad_(True)
        y = mod(HF.to_channels_last(xd))
        assert y.dtype == torch.bfloat16
        y.backward(dy.float().to(dev).to(y.dtype))
        torch.cuda.synchronize()
    finally:
        HF.set_compute_dtype("f32")
    f_hr, f_rx = l2(y.float(), res["round"][0]), l2(res["round"][0], res["exact"][0])
    d_hr, d_rx = l2(xd.grad, res["round"][1]), l2(res["round"][1], res["exact"][1])
    REPORT.append("bf16 block %-14s fwd |hip-round| %.2e |round-exact| %.2e   dx |hip-round| %.2e |round-exact| %.2e" % (tag, f_hr, f_rx, d_hr, d_rx))
    assert f_hr <= 4e-3, (f_hr, f_rx)
    assert d_hr <= max(1.2e-2, 0.25 * d_rx), (d_hr, d_rx)
    for n, p in mod.named_parameters():
        gr, gx = res["round"][2][n], res["exact"][2][n]
        if gr is None or gr.numel() < 256:
            continue
        e, m = l2(p.grad, gr), l2(gr, gx)
        REPORT.append("    d%-40s |hip-round| %.2e |round-exact| %.2e" % (n, e, m))
        assert e <= max(1.5e-2, 0.3 * m), (n, e, m)


@pytest.mark.parametrize("shape", [(3, 64, 96, 60), (2, 128, 48, 30), (2, 256, 24, 15), (3, 512, 12, 7)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_instance_norm_channels_last(shape, act):
    from hipops import functional as HF
    N, C, H, W_ = shape
    x = torch.randn(shape) * 3 + 1
    g = torch.randn(C); b = torch.randn(C)
    fn = (lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01))[act]
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = fn(F.instance_norm(xr, None, None, gr, br, True, 0.01, 1e-5))
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = cl(x).requires_grad_(True); gd = torch.nn.Parameter(g.to(dev)); bd = torch.nn.Parameter(b.to(dev))
    buf = cl(torch.zeros(N, 2 * C, H, W_))
    y = HF.instance_norm_cl(xd, gd, bd, 1e-5, act, 0.01, out=buf[:, C:])
    tag = "instnorm_cl %s act%d" % (shape, act)
    check(tag + " fwd", y, yr)
    assert buf[:, :C].abs().max().item() == 0
    y.backward(cl(dy.float()))
    check(tag + " dx", xd.grad, xr.grad); check(tag + " dgamma", gd.grad, gr.grad); check(tag + " dbeta", bd.grad, br.grad)


@pytest.mark.parametrize("g", [(3, 1024, 6, 3, 512, 4, 2, 1, (0, 1), True), (2, 256, 24, 15, 128, 3, 2, 1, (1, 1), True),
                               (2, 256, 24, 15, 128, 4, 2, 1, (0, 0), False), (2, 128, 48, 30, 64, 3, 2, 1, (1, 1), True)])
def test_conv_transpose_channels_last_autograd(g):
    from hipops import functional as HF
    N, Ci, h, w_, Co, k, s, p, op, bias = g
    x = torch.randn(N, Ci, h, w_).relu_()
    w = torch.randn(Ci, Co, k, k) * 0.1
    b = torch.randn(Co) if bias else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=op)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xd = cl(x).requires_grad_(True); wd = torch.nn.Parameter(cl(w)); bd = torch.nn.Parameter(b.to(dev)) if bias else None
    y = HF.conv_transpose2d_cl(xd, wd, bd, (s, s), (p, p), op)
    assert tuple(y.shape) == tuple(yr.shape)
    check("convT_cl fwd %s" % (g,), y, yr)
    y.backward(cl(dy.float()))
    check("convT_cl dx %s" % (g,), xd.grad, xr.grad)
    check("convT_cl dw %s" % (g,), wd.grad, wr.grad)
    if bias:
        check("convT_cl db %s" % (g,), bd.grad, br.grad)


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("g", [(3, 64, 20, 30, 64), (2, 128, 12, 15, 128), (2, 64, 9, 8, 192), (1, 32, 40, 17, 96), (5, 64, 3, 30, 64),
                               (2, 256, 12, 8, 256)], ids=lambda g: "x".join(map(str, g)))
def test_nhwc_x3_halo_form(g):
    """the halo form of the 3x3 stride-1 x3 convs (csrc/conv_nhwc_x3_halo.inc: one zero-padded activation patch per channel
    block, the nine taps as constant row shifts) forced through the raw C ABI (MGVAE_X3_FORCE=12 / 14), forward and data
    gradient, against fp64 at the family's fp32-grade bar (2e-5 of the largest entry) and against the implicit-GEMM form on
    the same operands: odd widths (a tile then starts mid-row and the last tile of a sample is ragged), several samples
    (tiles never straddle two samples), Cx != Cy, bias + ReLU, the activation-gradient mask, channel slices on both sides."""
    import os
    from hipops import _native as nat
    L = nat.lib()
    N, Cx, H, W_, Cy = g
    k, s, p = 3, 1, 1
    x = torch.randn(N, Cx, H, W_).relu_() * torch.logspace(-2, 2, Cx).view(1, Cx, 1, 1)
    w = (torch.randn(Cy, Cx, k, k) * 0.2 - 0.03) / torch.logspace(-2, 2, Cx).view(1, Cx, 1, 1)
    b = torch.randn(Cy)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), stride=s, padding=p)
    dy = torch.randn_like(yr).float()
    yr.backward(dy.double())
    xd, wd, dyd = cl(x), cl(w), cl(dy)
    nw = Cy * k * k * Cx
    wk3 = torch.empty(3 * nw, device=dev, dtype=torch.bfloat16); wt3 = torch.empty(3 * nw, device=dev, dtype=torch.bfloat16)
    assert L.mgvae_pack_conv_weights_x3(vp(wd), vp(wk3), vp(wt3), Cy, k * k, Cx, stream()) == 0
    d = nat.ConvDesc(N, Cx, H, W_, Cy, H, W_, k, k, s, s, p, p, Cx, 0, Cy, 0, 0, 0.0)
    # slices: x inside a wider tensor (offset 4), y into a wider tensor (offset 8), with bias + ReLU
    d2 = nat.ConvDesc(N, Cx, H, W_, Cy, H, W_, k, k, s, s, p, p, Cx + 8, 4, Cy + 12, 8, 1, 0.0)
    wide_x = cl(torch.randn(N, Cx + 8, H, W_)); wide_x[:, 4:4 + Cx] = xd
    mask_src = cl(torch.randn(N, Cx, H, W_))                      # dgrad epilogue mask: relu'(mask_src)
    m = nat.ActMask(mask_src.data_ptr(), Cx, 0, 1, 0.0)
    got = {}
    try:
        for tile in (12, 14, 15, 8):
            os.environ["MGVAE_X3_FORCE"] = "%d,1" % tile
            y = cl(torch.full((N, Cy, H, W_), 3.0))
            assert L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(xd), vp(wk3), vp(b.to(dev)), vp(y), None, None, 0, stream()) == 0
            dx = cl(torch.full((N, Cx, H, W_), 3.0))
            assert L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(dyd), vp(wt3), None, vp(dx), None, None, 0, stream()) == 0
            wide_y = cl(torch.full((N, Cy + 12, H, W_), 7.0))
            assert L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d2), vp(wide_x), vp(wk3), vp(b.to(dev)), vp(wide_y), None, None, 0, stream()) == 0
            dxm = cl(torch.zeros(N, Cx, H, W_))
            assert L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(dyd), vp(wt3), None, vp(dxm), ctypes.byref(m), None, 0, stream()) == 0
            torch.cuda.synchronize()
            got[tile] = (y.cpu(), dx.cpu(), wide_y.cpu(), dxm.cpu())
            assert _rel(y, yr) <= 2e-5, (tile, "fwd", _rel(y, yr))
            assert _rel(dx, xr.grad) <= 2e-5, (tile, "dx", _rel(dx, xr.grad))
            assert _rel(wide_y[:, 8:8 + Cy], F.relu(yr)) <= 2e-5, (tile, "sliced fwd")
            assert bool((wide_y[:, :8] == 7).all()) and bool((wide_y[:, 8 + Cy:] == 7).all()), "wrote outside its channel slice"
            assert _rel(dxm, xr.grad * (mask_src.cpu().double() > 0)) <= 2e-5, (tile, "masked dx")
            REPORT.append("x3 halo-form check %-22s tile %2d  fwd %.2e  dx %.2e" % (g, tile, _rel(y, yr), _rel(dx, xr.grad)))
    finally:
        os.environ.pop("MGVAE_X3_FORCE", None)
    # the weight gradient with the nine taps inside the workgroup (same file: K over the zero-padded positions, taps as row
    # shifts of the X operand's transposing reads); forced id 12 = that form wherever Cx and Cy are multiples of 64
    try:
        dws = {}
        for tile, split in ((12, 1), (12, 3), (8, 2)):
            os.environ["MGVAE_X3_FORCE"] = "%d,%d" % (tile, split)
            dw = cl(torch.zeros(Cy, Cx, k, k))
            assert L.mgvae_conv2d_nhwc_x3_bwd_weight(ctypes.byref(d), vp(xd), vp(dyd), vp(dw), stream()) == 0
            # and ACCUMULATING into a non-zero gradient, from a channel slice of a wider activation tensor
            dw2 = cl(torch.ones(Cy, Cx, k, k))
            dsl = nat.ConvDesc(N, Cx, H, W_, Cy, H, W_, k, k, s, s, p, p, Cx + 8, 4, Cy, 0, 0, 0.0)
            assert L.mgvae_conv2d_nhwc_x3_bwd_weight(ctypes.byref(dsl), vp(wide_x), vp(dyd), vp(dw2), stream()) == 0
            torch.cuda.synchronize()
            assert _rel(dw, wr.grad) <= 2e-5, (tile, split, "dw", _rel(dw, wr.grad))
            assert _rel(dw2 - 1.0, wr.grad) <= 2e-5, (tile, split, "dw accumulate / slice")
            dws[(tile, split)] = dw.cpu()
            REPORT.append("x3 weight gradient %-22s forced %2d,%d  dw %.2e" % (g, tile, split, _rel(dw, wr.grad)))
    finally:
        os.environ.pop("MGVAE_X3_FORCE", None)
    for tile in (12, 14, 15):      # against the implicit-GEMM form: same products, another summation order over K
        for a, bref in zip(got[tile], got[8]):
            assert float((a - bref).abs().max()) <= 1e-5 * float(bref.abs().max()), tile
    # where the form does not apply (stride 2; W + 2 > 32) a forced halo id falls back to an implicit-GEMM form
    os.environ["MGVAE_X3_FORCE"] = "12,1"
    try:
        dn = nat.ConvDesc(2, 64, 10, 40, 64, 10, 40, 3, 3, 1, 1, 1, 1, 64, 0, 64, 0, 0, 0.0)
        xn = torch.randn(2, 64, 10, 40); wn = torch.randn(64, 64, 3, 3) * 0.1
        wkn = torch.empty(3 * wn.numel(), device=dev, dtype=torch.bfloat16); wtn = torch.empty_like(wkn)
        assert L.mgvae_pack_conv_weights_x3(vp(cl(wn)), vp(wkn), vp(wtn), 64, 9, 64, stream()) == 0
        yn = cl(torch.zeros(2, 64, 10, 40))
        assert L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(dn), vp(cl(xn)), vp(wkn), None, vp(yn), None, None, 0, stream()) == 0
        assert _rel(yn, F.conv2d(xn.double(), wn.double(), None, 1, 1)) <= 2e-5
    finally:
        os.environ.pop("MGVAE_X3_FORCE", None)


@pytest.mark.parametrize("fam", ["x3", "bf16"])
def test_split_k_workspace_is_caller_owned(fam):
    """``size_t mgvae_<op>_workspace(desc, mode)`` + ``ws, ws_bytes`` in the entry points, no allocation
    inside the library.  On a deep, under-filled layer (512 -> 512, 3x3 on a 6x4 map: the geometry the deterministic split-K
    exists for) the query asks for a workspace; with every split forced in turn the result must be the same to fp32 rounding
    with a full, a half-sized (the split shrinks to fit), a null (plain launch) workspace, a filled-chip geometry asks for
    none, and nothing is written past ``ws_bytes``."""
    import os
    from hipops import _native as nat
    L = nat.lib()
    N, C, H, W_, k = 8, 512, 6, 4, 3
    d = nat.ConvDesc(N, C, H, W_, C, H, W_, k, k, 1, 1, 1, 1, C, 0, C, 0, 0, 0.0)
    q = getattr(L, "mgvae_conv2d_nhwc_%s_workspace" % fam)
    need = [int(q(ctypes.byref(d), m)) for m in (0, 1, 2)]
    rows = N * H * W_
    assert need[0] == need[1] == 4 * rows * C * 4 and need[2] == 0, need
    big = nat.ConvDesc(64, 64, 192, 30, 64, 192, 30, 3, 3, 1, 1, 1, 1, 64, 0, 64, 0, 0, 0.0)
    assert int(q(ctypes.byref(big), 0)) == 0 and int(q(ctypes.byref(big), 1)) == 0        # 5760 workgroups: never split
    torch.manual_seed(3)
    x = torch.randn(N, C, H, W_).relu_(); w = torch.randn(C, C, k, k) * 0.05
    yr = F.conv2d(x.double(), w.double(), None, 1, 1)
    dt = torch.float32 if fam == "x3" else torch.bfloat16
    xd = cl(x).to(dt); wd = cl(w)
    planes = 3 if fam == "x3" else 1
    wk = torch.empty(planes * w.numel(), device=dev, dtype=torch.bfloat16); wt = torch.empty_like(wk)
    pack = L.mgvae_pack_conv_weights_x3 if fam == "x3" else L.mgvae_pack_conv_weights_bf16
    assert pack(vp(wd), vp(wk), vp(wt), C, k * k, C, stream()) == 0
    fwd = getattr(L, "mgvae_conv2d_nhwc_%s_fwd" % fam)
    guard = 4096
    ws = torch.full((need[0] + guard,), 0x5A, device=dev, dtype=torch.uint8)
    tol = 2e-5 if fam == "x3" else 2e-2
    outs = []
    try:
        for split in (1, 2, 4):
            for nbytes in (need[0], need[0] // 2, 0):
                os.environ["MGVAE_X3_FORCE" if fam == "x3" else "MGVAE_BF16_FORCE"] = "0,%d" % split
                y = cl(torch.zeros(N, C, H, W_)).to(dt)
                ws.fill_(0x5A)
                assert fwd(ctypes.byref(d), vp(xd), vp(wk), None, vp(y), None, vp(ws) if nbytes else None, nbytes, stream()) == 0
                torch.cuda.synchronize()
                assert bool((ws[nbytes:] == 0x5A).all()), "the library wrote past ws_bytes (split %d, %d bytes)" % (split, nbytes)
                used = bool((ws[:max(nbytes, 1)] != 0x5A).any()) if nbytes else False
                assert used == (split > 1 and nbytes > 0), (split, nbytes, used)
                assert _rel(y, yr) <= tol, (split, nbytes, _rel(y, yr))
                outs.append(y.float().cpu())
    finally:
        os.environ.pop("MGVAE_X3_FORCE", None); os.environ.pop("MGVAE_BF16_FORCE", None)
    # deterministic: the same split and workspace size twice gives the same bits (no atomics in the forward-like products)
    assert all(float((o - outs[0]).abs().max()) <= tol * float(outs[0].abs().max()) for o in outs)


@pytest.mark.parametrize("g", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_nhwc_x3_conv_three_products_are_fp32_grade(g):
    """fp32-storage family on the bf16 matrix pipe (csrc/conv_nhwc_x3.inc): every fp32 operand enters as three bf16 parts,
    every product as six MFMAs.  Same C-ABI shapes and the same fp64 reference as the fp32-MFMA test above, but the bar is
    fp32 ROUNDING, not 1e-3: 2e-5 of the largest entry (measured 1e-7 - 4e-6), and never worse than three times what the
    fp32 matrix instruction itself achieves on the same operands (the two sum K in different orders).  The weight planes are checked to re-sum to the fp32 master exactly."""
    from hipops import _native as nat
    L = nat.lib()
    N, Cx, H, W_, Cy, k, s, p = g
    OH, OW = (H + 2 * p - k) // s + 1, (W_ + 2 * p - k) // s + 1
    x = torch.randn(N, Cx, H, W_).relu_() * torch.logspace(-3, 3, Cx).view(1, Cx, 1, 1)        # six decades of scale
    w = (torch.randn(Cy, Cx, k, k) * 0.2 - 0.05) / torch.logspace(-3, 3, Cx).view(1, Cx, 1, 1)
    b = torch.randn(Cy)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), stride=s, padding=p)
    dy = torch.randn_like(yr).float()
    yr.backward(dy.double())
    xd, wd, dyd = cl(x), cl(w), cl(dy)
    nw = Cy * k * k * Cx
    wk3 = torch.empty(3 * nw, device=dev, dtype=torch.bfloat16); wt3 = torch.empty(3 * nw, device=dev, dtype=torch.bfloat16)
    assert L.mgvae_pack_conv_weights_x3(vp(wd), vp(wk3), vp(wt3), Cy, k * k, Cx, stream()) == 0
    def planes_of(buf, rows, K):
        """the three planes as [3, rows, k, k, K]: classic [plane][row][tap][k], or -- both channel counts multiples of 32 -- the
        block-major [tap][k / 32][row][plane][32] of csrc/conv_nhwc_x3.inc::x3w_blocked"""
        if Cy % 32 == 0 and Cx % 32 == 0:
            return buf.view(k * k, K // 32, rows, 3, 32).permute(3, 2, 0, 1, 4).reshape(3, rows, k, k, K).float()
        return buf.view(3, rows, k, k, K).float()
    planes = planes_of(wk3, Cy, Cx)
    assert torch.equal((planes[0] + planes[1] + planes[2]).permute(0, 3, 1, 2).cpu(), w), "h + m + l must be the fp32 weight, exactly"
    planes = planes_of(wt3, Cx, Cy)
    assert torch.equal((planes[0] + planes[1] + planes[2]).permute(3, 0, 1, 2).cpu(), w)
    d = nat.ConvDesc(N, Cx, H, W_, Cy, OH, OW, k, k, s, s, p, p, Cx, 0, Cy, 0, 0, 0.0)
    got, ref = {}, {}
    y = cl(torch.zeros(N, Cy, OH, OW)); y0 = cl(torch.zeros(N, Cy, OH, OW))
    assert L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(xd), vp(wk3), vp(b.to(dev)), vp(y), None, None, 0, stream()) == 0
    assert L.mgvae_conv2d_nhwc_fwd(ctypes.byref(d), vp(xd), vp(wd), vp(b.to(dev)), vp(y0), None, stream()) == 0
    got["fwd"], ref["fwd"] = _rel(y, yr), _rel(y0, yr)
    dx = cl(torch.zeros(N, Cx, H, W_)); dx0 = cl(torch.zeros(N, Cx, H, W_))
    assert L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), vp(dyd), vp(wt3), None, vp(dx), None, None, 0, stream()) == 0
    assert L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d), vp(dyd), vp(wd), None, vp(dx0), None, stream()) == 0
    got["dx"], ref["dx"] = _rel(dx, xr.grad), _rel(dx0, xr.grad)
    dw = cl(torch.zeros(Cy, Cx, k, k)); dw0 = cl(torch.zeros(Cy, Cx, k, k))
    assert L.mgvae_conv2d_nhwc_x3_bwd_weight(ctypes.byref(d), vp(xd), vp(dyd), vp(dw), stream()) == 0
    assert L.mgvae_conv2d_nhwc_bwd_weight(ctypes.byref(d), vp(xd), vp(dyd), vp(dw0), stream()) == 0
    got["dw"], ref["dw"] = _rel(dw, wr.grad), _rel(dw0, wr.grad)
    for kname in got:
        REPORT.append("x3 %-4s %-34s rel=%.3e   fp32-MFMA kernel rel=%.3e" % (kname, g, got[kname], ref[kname]))
        assert got[kname] <= 2e-5, (kname, got[kname])
        assert got[kname] <= max(3 * ref[kname], 5e-6), (kname, got[kname], ref[kname])
    # transposed-conv forward with bias, activation and an epilogue mask, slices of wider tensors
    bt = torch.randn(Cx)
    tr = F.relu(F.conv_transpose2d(dy.double(), w.double(), bt.double(), stride=s, padding=p,
                                   output_padding=(H - ((OH - 1) * s - 2 * p + k), W_ - ((OW - 1) * s - 2 * p + k))))
    d2 = nat.ConvDesc(N, Cx, H, W_, Cy, OH, OW, k, k, s, s, p, p, Cx + 8, 4, Cy + 4, 4, 1, 0.0)
    wide_y = cl(torch.zeros(N, Cy + 4, OH, OW)); wide_y[:, 4:] = dyd
    wide_x = cl(torch.full((N, Cx + 8, H, W_), 7.0))
    assert L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d2), vp(wide_y), vp(wt3), vp(bt.to(dev)), vp(wide_x), None, None, 0, stream()) == 0
    check("x3 convT+bias+relu into a slice %s" % (g,), wide_x[:, 4:4 + Cx], tr, 2e-5)
    assert (wide_x[:, :4] == 7).all() and (wide_x[:, 4 + Cx:] == 7).all()


BF16_GEOMS = [  # N, Cx, H, W, Cy, k, s, p  (channel counts multiples of 64)
    (3, 64, 48, 30, 64, 3, 1, 1), (2, 128, 24, 15, 128, 3, 1, 1), (3, 512, 6, 4, 512, 3, 1, 1), (3, 64, 48, 30, 128, 3, 2, 1),
    (5, 128, 24, 15, 256, 3, 2, 1), (3, 2048, 6, 3, 1024, 1, 1, 0), (2, 128, 24, 15, 256, 4, 2, 1), (2, 64, 96, 60, 128, 3, 2, 1),
    (1, 64, 5, 4, 64, 3, 1, 1),
]


@pytest.mark.parametrize("g", BF16_GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_nhwc_bf16_conv_three_products(g):
    """bf16-storage channels-last family (BASELINE.json configs 3-4) through the C ABI: operands are bf16 tensors, the
    reference multiplies the SAME bf16 values in fp64.  Forward / data gradient store bf16 (one rounding of the fp32
    accumulator: 2^-9 relative per element -> 4e-3 of the largest entry), the weight gradient stays fp32 (1e-3)."""
    from hipops import _native as nat
    L = nat.lib()
    N, Cx, H, W_, Cy, k, s, p = g
    OH, OW = (H + 2 * p - k) // s + 1, (W_ + 2 * p - k) // s + 1
    bf = lambda t: t.bfloat16()
    x = bf(torch.randn(N, Cx, H, W_).relu_()); w = bf(torch.randn(Cy, Cx, k, k) * 0.2 - 0.05); b = torch.randn(Cy)
    dy = bf(torch.randn(N, Cy, OH, OW))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), stride=s, padding=p)
    yr.backward(dy.double())
    clb = lambda t: t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)      # bf16 channels-last
    xd, dyd = clb(x), clb(dy)
    w32 = cl(w.float())                                                                  # fp32 master, channels-last
    wk = torch.empty(Cy * k * k * Cx, device=dev, dtype=torch.bfloat16)
    wt = torch.empty(Cy * k * k * Cx, device=dev, dtype=torch.bfloat16)
    assert L.mgvae_pack_conv_weights_bf16(vp(w32), vp(wk), vp(wt), Cy, k * k, Cx, stream()) == 0
    assert torch.equal(wk.view(Cy, k, k, Cx).permute(0, 3, 1, 2).cpu(), w)
    assert torch.equal(wt.view(Cx, k, k, Cy).permute(3, 0, 1, 2).cpu(), w)
    d = nat.ConvDesc(N, Cx, H, W_, Cy, OH, OW, k, k, s, s, p, p, Cx, 0, Cy, 0, 0, 0.0)
    yd = clb(torch.zeros(N, Cy, OH, OW).bfloat16())
    assert L.mgvae_conv2d_nhwc_bf16_fwd(ctypes.byref(d), vp(xd), vp(wk), vp(b.to(dev)), vp(yd), None, None, 0, stream()) == 0
    check("nhwc bf16 fwd %s" % (g,), yd.float(), yr, 4e-3)
    dx = clb(torch.zeros(N, Cx, H, W_).bfloat16())
    assert L.mgvae_conv2d_nhwc_bf16_bwd_data(ctypes.byref(d), vp(dyd), vp(wt), None, vp(dx), None, None, 0, stream()) == 0
    check("nhwc bf16 dx %s" % (g,), dx.float(), xr.grad, 4e-3)
    dw = cl(torch.ones(Cy, Cx, k, k))
    assert L.mgvae_conv2d_nhwc_bf16_bwd_weight(ctypes.byref(d), vp(xd), vp(dyd), vp(dw), stream()) == 0
    check("nhwc bf16 dw %s" % (g,), dw, 1.0 + wr.grad)


@pytest.mark.parametrize("shape", [(3, 64, 48, 30), (2, 256, 12, 8), (2, 1024, 3, 2), (3, 32, 48, 30)])
@pytest.mark.parametrize("mode,act", [(1, 1), (2, 1)])
def test_norm_cbam_channels_last_bf16_storage(shape, mode, act):
    """the fused InstanceNorm / CBAM op with bf16 STORAGE of the big tensors (statistics, gates, arithmetic fp32): against the
    fp64 oracle evaluated on the SAME bf16 input values.  The output and dx are rounded to bf16 once (2^-9 per element);
    inside the backward the partial du is stored in bf16 as well, so dx carries two roundings: 1e-2 of the largest entry."""
    from hipops import functional as HF
    from oracle import restate as R
    N, C, H, W_ = shape
    bf = lambda t: t.bfloat16()
    x = bf(torch.randn(shape) * 2 + 0.5); res = bf(torch.randn(shape)); dy = bf(torch.randn(shape))
    g = torch.randn(C); b = torch.randn(C)
    sd = {"channel_attention.conv1.weight": torch.randn(C // 16, C, 1, 1) * 0.2,
          "channel_attention.conv2.weight": torch.randn(C, C // 16, 1, 1) * 0.2,
          "spatial_attention.conv.weight": torch.randn(1, 2, 3, 3) * 0.3}
    sdr = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ur = F.instance_norm(xr, None, None, gr, br, True, 0.01, 1e-5)
    o = R.cbam(sdr, "", ur)
    yr = F.relu(ur + o) if mode == 1 else F.relu(rr + o)
    yr.backward(dy.double())
    clb = lambda t: t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xd = clb(x).requires_grad_(True); rd = clb(res).requires_grad_(True)
    gd = torch.nn.Parameter(g.to(dev)); bd = torch.nn.Parameter(b.to(dev))
    ps = {k: torch.nn.Parameter(v.to(dev)) for k, v in sd.items()}
    y = HF.norm_cbam_cl(xd, gd, bd, ps["channel_attention.conv1.weight"], ps["channel_attention.conv2.weight"],
                        ps["spatial_attention.conv.weight"], 1e-5, mode, rd if mode == 2 else None, act, 0.01)
    assert y.dtype == torch.bfloat16
    tag = "norm_cbam_cl bf16 %s mode%d" % (shape, mode)
    check(tag + " fwd", y.float(), yr, 4e-3)
    # the reference's y differs from the stored bf16 y by one rounding; ReLU masks are read from the stored y
    y.backward(clb(dy))
    assert xd.grad.dtype == torch.bfloat16
    check(tag + " dx", xd.grad.float(), xr.grad, 1e-2)
    check(tag + " dgamma", gd.grad, gr.grad, 4e-3); check(tag + " dbeta", bd.grad, br.grad, 4e-3)
    if mode == 2:
        check(tag + " dres", rd.grad.float(), rr.grad, 4e-3)
    for k in sd:
        check(tag + " d" + k, ps[k].grad, sdr[k].grad, 1e-2)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(3, 96, 60, (4, 1), (2, 1), (1, 0)), (3, 96, 60, (1, 4), (1, 2), (0, 1)), (2, 384, 60, (4, 1), (2, 1), (1, 0)),
                                  (5, 17, 9, (1, 4), (1, 2), (0, 1))])
def test_one_channel_conv_writes_channels_last(geom, storage):
    """the encoder stems' first conv (graph/encodingBlock.py:11-14,42-45: one input channel -> 32, LeakyReLU) written
    channels-last by csrc/thin_nhwc.hip: forward against torch fp64; the weight gradient with the activation gradient applied
    inside the kernel (mask) and with an already-masked gradient (the deferred form the stems use); accumulation into dw"""
    from hipops import _native as nat
    from hipops import functional as HF
    L = nat.lib()
    N, H, W, k, s, p = geom
    torch.manual_seed(3)
    x = torch.randn(N, 1, H, W)
    w = torch.randn(32, 1, *k) * 0.5
    dy = torch.randn(N, 32, (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1)
    xr, wr = x.double(), w.double().requires_grad_(True)
    yr = F.leaky_relu(F.conv2d(xr, wr, None, s, p), 0.01)
    yr.backward(dy.double())
    dt = torch.bfloat16 if storage == "bf16" else torch.float32
    tol = 8e-3 if storage == "bf16" else 2e-5
    wd = torch.nn.Parameter(w.to(dev))
    y = HF.conv2d_c1_cl(x.to(dev), wd, s, p, HF.ACT_LEAKY, 0.01, dtype=dt)
    assert y.dtype == dt and HF.cl_pitch(y) == 32
    check("conv_c1 %s %s fwd" % (geom, storage), y.float(), yr.detach(), tol)
    # (a) the kernel applies LeakyReLU'(y) itself; in bf16 storage dy is rounded once more
    y.backward(cl(dy).to(dt))
    check("conv_c1 %s %s dw (mask in the kernel)" % (geom, storage), wd.grad, wr.grad, tol)
    # (b) deferred: the consumer has applied the mask; a second backward ACCUMULATES
    OH, OW = dy.shape[2:]
    d = nat.ConvDesc(N, 1, H, W, 32, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], 1, 0, 32, 0, HF.ACT_LEAKY, 0.01)
    g = cl(dy * torch.where(yr.detach() > 0, 1.0, 0.01).float()).to(dt)
    dw = wd.grad.clone()
    assert L.mgvae_conv2d_c1_nhwc_bwd_weight(ctypes.byref(d), vp(x.to(dev).contiguous()), vp(g), None, vp(dw), 1 if storage == "bf16" else 0, stream()) == 0
    check("conv_c1 %s %s dw (deferred, accumulated)" % (geom, storage), dw, 2 * wr.grad, tol)
    # refusals: more than one input channel, too many taps, a sigmoid
    bad = nat.ConvDesc(N, 2, H, W, 32, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], 2, 0, 32, 0, 0, 0.0)
    assert L.mgvae_conv2d_c1_nhwc_fwd(ctypes.byref(bad), vp(g), vp(dw), vp(g), 0, stream()) == -1


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 64, 96, 60), (2, 32, 17, 9), (2, 128, 6, 5)])
def test_conv_to_one_channel_reads_channels_last(shape, storage):
    """the decoder's fit2 + Sigmoid (graph/decoder.py:186,217) on the channels-last map: forward, dx, dw against torch fp64 on
    the same (rounded) input values; a frozen weight still gives dx"""
    from hipops import functional as HF
    N, C, H, W = shape
    torch.manual_seed(4)
    dt = torch.bfloat16 if storage == "bf16" else torch.float32
    x = (torch.randn(shape) * 0.7).to(dt).float()
    w = torch.randn(1, C, 1, 1) * 0.3
    dy = torch.randn(N, 1, H, W)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = torch.sigmoid(F.conv2d(xr, wr))
    yr.backward(dy.double())
    xd = cl(x).to(dt).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(dev))
    y = HF.conv2d_to1_cl(xd, wd, HF.ACT_SIGMOID)
    assert tuple(y.shape) == (N, 1, H, W) and y.dtype == torch.float32 and y.is_contiguous()
    check("conv_to1 %s %s fwd" % (shape, storage), y, yr.detach(), 2e-5)
    y.backward(dy.to(dev))
    check("conv_to1 %s %s dx" % (shape, storage), xd.grad.float(), xr.grad, 8e-3 if storage == "bf16" else 2e-5)
    check("conv_to1 %s %s dw" % (shape, storage), wd.grad, wr.grad, 2e-5)
    wd2 = torch.nn.Parameter(w.to(dev), requires_grad=False)
    xd2 = cl(x).to(dt).requires_grad_(True)
    HF.conv2d_to1_cl(xd2, wd2, HF.ACT_SIGMOID).backward(dy.to(dev))
    assert torch.equal(xd2.grad, xd.grad) and wd2.grad is None


def test_storage_cast_round_trip():
    from hipops import functional as HF
    x = cl(torch.randn(3, 64, 12, 10)).requires_grad_(True)
    y = HF.cast_cl(x, torch.bfloat16)
    assert y.dtype == torch.bfloat16 and HF.cl_pitch(y) == 64 and torch.equal(y, x.detach().bfloat16())
    g = cl(torch.randn(3, 64, 12, 10)).bfloat16()
    y.backward(g)
    assert x.grad.dtype == torch.float32 and torch.equal(x.grad, g.float())
    assert HF.cast_cl(x, torch.float32) is x


@pytest.mark.parametrize("g", [(3, 48, 12, 10, 80, 3, 1, 1), (2, 16, 9, 7, 48, 3, 2, 1), (2, 80, 6, 5, 16, 1, 1, 0), (130, 48, 4, 4, 144, 3, 1, 1),
                               (2, 64, 12, 10, 160, 3, 1, 1), (3, 128, 9, 7, 64, 4, 2, 1)])
def test_nhwc_x3_ring_form(g):
    """the LDS-DMA ring form of the x3 forward product (csrc/conv_nhwc_x3_ring.inc, an experiment switched on by MGVAE_X3_RING;
    DESIGN.md section 3.12): against torch fp64 at the family's tolerance, with bias and ReLU, ragged tiles in both directions,
    padding taps (zero fill of out-of-range lanes), 2 / 3 / 6 stages; both weight-plane layouts (classic: a channel count that
    is not a multiple of 32; block-major otherwise)."""
    import os
    from hipops import _native as nat
    L = nat.lib()
    N, Cx, H, W_, Cy, k, s, p = g
    OH, OW = (H + 2 * p - k) // s + 1, (W_ + 2 * p - k) // s + 1
    torch.manual_seed(11)
    x = torch.randn(N, Cx, H, W_); w = torch.randn(Cy, Cx, k, k) * 0.2; b = torch.randn(Cy)
    yr = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p))
    xd, wd = cl(x), cl(w)
    nw = Cy * k * k * Cx
    wk3 = torch.empty(3 * nw, device=dev, dtype=torch.bfloat16); wt3 = torch.empty(3 * nw, device=dev, dtype=torch.bfloat16)
    assert L.mgvae_pack_conv_weights_x3(vp(wd), vp(wk3), vp(wt3), Cy, k * k, Cx, stream()) == 0
    d = nat.ConvDesc(N, Cx, H, W_, Cy, OH, OW, k, k, s, s, p, p, Cx, 0, Cy, 0, 1, 0.0)
    try:
        for stages in (2, 3, 6):
            os.environ["MGVAE_X3_RING"] = str(stages)
            y = cl(torch.full((N, Cy, OH, OW), 7.0))
            assert L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), vp(xd), vp(wk3), vp(b.to(dev)), vp(y), None, None, 0, stream()) == 0
            check("x3 ring form %s, %d stages" % (g, stages), y, yr, 2e-5)
    finally:
        os.environ.pop("MGVAE_X3_RING", None)
