"""Whole blocks of the channels-last island as ONE autograd node and ONE foreign call per direction.

The reference's block boundaries -- ResidualModule / PoolingModule (graph/encodingBlock.py:87-100,118-126), DeConvModule /
DeConvPitchPadding (graph/decoder.py:91-109,135-154), and the decoder's fit1 + InstanceNorm + CBAM (graph/decoder.py:213-215)
-- are where this build cuts its launch chains (hipops/chain.py, csrc/chain.hip).  The forward of a block is the same list
of entry points, in the same order, that the per-op autograd functions of hipops/functional.py issue; so is the backward,
with three differences that autograd used to supply from outside: the two gradient contributions of a tensor with two
consumers inside a block (the residual input; the input of a decoder block's twin transposed convs) are summed by
mgvae_add_inplace_typed instead of an ATen add, the weight gradients are forked onto the side stream by mgvae_stream_fork
instead of torch stream contexts, and intermediate gradients never become autograd tensors.

``MGVAE_CHAIN=0`` switches the island back to one node per op (tests run both; tools/ab_env.sh measures the difference).
"""
import ctypes
import os

import torch

from . import _native as nat
from . import functional as HF
from .chain import Chain

ENABLED = os.environ.get("MGVAE_CHAIN", "1") != "0"
_chains = {}


def usable(x):
    """chains cover the channels-last island in its three engines; anything else keeps the per-op path"""
    if not ENABLED or not HF.DEFER_ACT_GRAD or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    return HF.cl_pitch(x) is not None


def _engine(x, *channels):
    if x.dtype == torch.bfloat16:
        return "bf16"
    if HF.FP32_ENGINE == "x3" and all(c % 16 == 0 for c in channels):
        return "x3"
    return "f32"


def _weights(eng, w):
    """(forward operand, data-gradient operand) of conv weight ``w`` for the engine"""
    if eng == "x3":
        return HF._x3_weights(w)
    if eng == "bf16":
        return HF._bf16_weights(w)
    return w, w


def _desc(N, Cx, H, W, Cy, OH, OW, k, s, p, xct, yct, act=HF.ACT_NONE, slope=0.0):
    return nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], xct, 0, yct, 0, act, slope)


def _ws_bytes(eng, d, mode):
    if eng == "f32":
        return 0
    fn = nat.lib().mgvae_conv2d_nhwc_x3_workspace if eng == "x3" else nat.lib().mgvae_conv2d_nhwc_bf16_workspace
    return int(fn(ctypes.byref(d), mode))


class _Emit:
    """appends the conv entry points of one engine to a chain; tracks the largest split-K workspace any of them may use"""

    def __init__(self, ch, eng, ws, st):
        self.ch, self.eng, self.ws, self.st, self.ws_bytes = ch, eng, ws, st, 0

    def _ws(self, d, mode):
        n = _ws_bytes(self.eng, d, mode)
        self.ws_bytes = max(self.ws_bytes, n)
        return n

    def fwd(self, d, x, wk, bias, y, mask=None):
        """Y = conv(X) -- also the data gradient of a transposed conv"""
        ref = self.ch.struct(d)
        if self.eng == "f32":
            self.ch.call("mgvae_conv2d_nhwc_fwd", ref, x, wk, bias, y, mask, self.st)
        else:
            n = self._ws(d, 0)
            self.ch.call("mgvae_conv2d_nhwc_%s_fwd" % self.eng, ref, x, wk, bias, y, mask, self.ws if n else None, n, self.st)

    def bwd_data(self, d, y, wt, bias, x, mask=None):
        """X = conv_transpose(Y) -- the data gradient of a conv, and a transposed conv's forward"""
        ref = self.ch.struct(d)
        if self.eng == "f32":
            self.ch.call("mgvae_conv2d_nhwc_bwd_data", ref, y, wt, bias, x, mask, self.st)
        else:
            n = self._ws(d, 1)
            self.ch.call("mgvae_conv2d_nhwc_%s_bwd_data" % self.eng, ref, y, wt, bias, x, mask, self.ws if n else None, n, self.st)

    def bwd_weight(self, d, x, y, dw, stream):
        name = {"f32": "mgvae_conv2d_nhwc_bwd_weight", "x3": "mgvae_conv2d_nhwc_x3_bwd_weight",
                "bf16": "mgvae_conv2d_nhwc_bf16_bwd_weight"}[self.eng]
        self.ch.call(name, self.ch.struct(d), x, y, dw, stream)

    def mask(self, src, ctot, act, slope=0.0):
        return self.ch.struct(nat.ActMask(0, ctot, 0, act, slope), (("src", src),))


def _new(n, c, h, w, like):
    return HF.new_channels_last(n, c, h, w, like.device, like.dtype)


def _f32(n, dev):
    return torch.empty((n,), device=dev, dtype=torch.float32)


def _ptr(t):
    return t.data_ptr() if t is not None else 0


def _main():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _grad(p):
    return HF.grad_slot(p) if (p is not None and p.requires_grad) else None


def _side_for(n, trainable, need_dx, touched):
    """the stream this backward's weight gradients run on: the side stream paired with the current one (and the
    end-of-backward join armed, the tensors it reads marked for the allocator) when the batch is large enough for forking to
    pay (hipops/functional.py: FORK_MIN_BATCH), else the current stream itself"""
    cur_raw = _main()
    if not (HF.FORK_WGRAD and n >= HF.FORK_MIN_BATCH and need_dx and trainable):
        if HF._trunk_streams or HF._used_sides:
            HF._ensure_join_callback()
        return cur_raw, cur_raw
    cur = HF.cur_stream()
    HF._wgrad_rr[0] += 1
    slot = 2 + HF._wgrad_rr[0] % HF.WGRAD_STREAMS
    side = HF.side_stream_of(cur, slot)
    for t in touched:
        if t is not None:
            t.record_stream(side)
    HF._used_sides[id(side)] = (side, slot, None if HF._shared_sides() else cur.cuda_stream)
    HF._ensure_join_callback()
    return cur_raw, side.cuda_stream


def _gate_wgrad(ch, f1, f2, s_save, s_scr, s_d1, s_d2, N, C, H, W, s_st, s_side):
    """the CBAM gate MLP's weight gradients as a launch of their own on the weight-gradient stream: mgvae_norm_cbam_nhwc_bwd was
    given NULL for them, so the data-gradient chain does not wait for a kernel only the optimizer needs (it was 0.25 ms per
    step alone on the chip -- tools/trace_timeline.py)"""
    if f1 or f2:
        ch.call("mgvae_stream_fork", s_st, s_side)
        ch.call("mgvae_norm_cbam_nhwc_bwd_mlp_wgrad", s_save, s_scr, s_d1 if f1 else None, s_d2 if f2 else None, N, C, H, W, s_side)


def _dense_cl(dy, like, pitch):
    """the incoming gradient as a channels-last tensor of ``like``'s storage type whose pixel pitch is ``pitch``"""
    dy = HF._as_cl(dy, like)
    if HF.cl_pitch(dy) != pitch:
        dy = dy.contiguous(memory_format=HF.CL)
        if HF.cl_pitch(dy) != pitch:
            raise RuntimeError("block backward: gradient pitch %s does not match the output's %s" % (HF.cl_pitch(dy), pitch))
    return dy


# ============================================================================================== residual block
class _ResidualFn(torch.autograd.Function):
    """relu(x + CBAM(IN(conv2(relu(conv1(x)))))) -- graph/encodingBlock.py:87-100"""

    @staticmethod
    def forward(ctx, x, w1, w2, gamma, beta, ca1, ca2, sa, eps):
        N, C, H, W = x.shape
        xct = HF._need_cl(x, "residual block")
        HF._cl_weight(w1, "residual block"); HF._cl_weight(w2, "residual block")
        eng = _engine(x, C)
        st = HF._store(x)
        key = ("res-f", N, C, H, W, xct, eng, float(eps))
        e = _chains.get(key)
        if e is None:
            ch = Chain()
            s_x, s_wk1, s_wk2, s_g, s_b, s_c1, s_c2, s_sa, s_t1, s_t2, s_y, s_save, s_ws, s_st = ch.slots(14)
            em = _Emit(ch, eng, s_ws, s_st)
            k, o, p = (3, 3), (1, 1), (1, 1)
            em.fwd(_desc(N, C, H, W, C, H, W, k, o, p, xct, C, HF.ACT_RELU, 0.01), s_x, s_wk1, None, s_t1)
            em.fwd(_desc(N, C, H, W, C, H, W, k, o, p, C, C), s_t1, s_wk2, None, s_t2)
            ch.call("mgvae_norm_cbam_nhwc_fwd", s_t2, s_g, s_b, s_x, xct, 0, s_c1, s_c2, s_sa, s_y, s_save, N, C, H, W, C, 0,
                    eps, 2, HF.ACT_RELU, 0.01, st, s_st)
            e = _chains[key] = (ch.finalize(), em.ws_bytes, int(nat.lib().mgvae_norm_cbam_nhwc_save_floats(N, C, H, W)))
        ch, wsn, nsave = e
        wk1, _ = _weights(eng, w1)
        wk2, _ = _weights(eng, w2)
        t1, t2, y = _new(N, C, H, W, x), _new(N, C, H, W, x), _new(N, C, H, W, x)
        save = _f32(nsave, x.device)
        ws = torch.empty(wsn, device=x.device, dtype=torch.uint8) if wsn else None
        ch.run([x.data_ptr(), wk1.data_ptr(), wk2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ca1.data_ptr(), ca2.data_ptr(),
                sa.data_ptr(), t1.data_ptr(), t2.data_ptr(), y.data_ptr(), save.data_ptr(), _ptr(ws), _main()])
        ctx.save_for_backward(x, w1, w2, gamma, beta, ca1, ca2, sa, t1, t2, y, save)
        ctx.cfg = (eng, xct, float(eps))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, gamma, beta, ca1, ca2, sa, t1, t2, y, save = ctx.saved_tensors
        eng, xct, eps = ctx.cfg
        N, C, H, W = x.shape
        st = HF._store(x)
        need_dx = ctx.needs_input_grad[0]
        flags = tuple(bool(p.requires_grad) for p in (w1, w2, gamma, beta, ca1, ca2, sa))
        key = ("res-b", N, C, H, W, xct, eng, need_dx, flags)
        e = _chains.get(key)
        if e is None:
            ch = Chain()
            (s_x, s_t1, s_t2, s_y, s_dy, s_wt1, s_wt2, s_g, s_b, s_c1, s_c2, s_sa, s_save, s_dt2, s_dres, s_dt1, s_dx, s_dg, s_db,
             s_dc1, s_dc2, s_dsa, s_dw1, s_dw2, s_scr, s_ws, s_st, s_side) = ch.slots(28)
            em = _Emit(ch, eng, s_ws, s_st)
            k, o, p = (3, 3), (1, 1), (1, 1)
            ch.call("mgvae_norm_cbam_nhwc_bwd", s_t2, s_g, s_b, s_y, s_dy, s_c1, s_c2, s_sa, s_save, s_dt2, s_dres, s_dg, s_db,
                    None, None, s_dsa, s_scr, N, C, H, W, C, 0, 2, HF.ACT_RELU, 0.01, st, s_st)
            _gate_wgrad(ch, flags[4], flags[5], s_save, s_scr, s_dc1, s_dc2, N, C, H, W, s_st, s_side)
            if flags[1]:
                ch.call("mgvae_stream_fork", s_st, s_side)
                em.bwd_weight(_desc(N, C, H, W, C, H, W, k, o, p, C, C), s_t1, s_dt2, s_dw2, s_side)
            # conv2's data gradient applies relu'(t1) while storing (conv1 skipped its own activation-gradient pass)
            em.bwd_data(_desc(N, C, H, W, C, H, W, k, o, p, C, C), s_dt2, s_wt2, None, s_dt1, em.mask(s_t1, C, HF.ACT_RELU))
            if flags[0]:
                ch.call("mgvae_stream_fork", s_st, s_side)
                em.bwd_weight(_desc(N, C, H, W, C, H, W, k, o, p, xct, C), s_x, s_dt1, s_dw1, s_side)
            if need_dx:
                em.bwd_data(_desc(N, C, H, W, C, H, W, k, o, p, C, C), s_dt1, s_wt1, None, s_dx)
                ch.call("mgvae_add_inplace_typed", s_dx, s_dres, N * C * H * W, st, s_st)
            e = _chains[key] = (ch.finalize(), em.ws_bytes, int(nat.lib().mgvae_norm_cbam_nhwc_scratch_floats(N, C, H, W)))
        ch, wsn, nscr = e
        dy = _dense_cl(dy, x, C)
        _, wt1 = _weights(eng, w1)
        _, wt2 = _weights(eng, w2)
        dt2, dres, dt1 = _new(N, C, H, W, x), _new(N, C, H, W, x), _new(N, C, H, W, x)
        dx = _new(N, C, H, W, x) if need_dx else None
        scr = _f32(nscr, x.device)
        ws = torch.empty(wsn, device=x.device, dtype=torch.uint8) if wsn else None
        main, side = _side_for(N, flags[0] or flags[1] or flags[4] or flags[5], need_dx, (x, t1, dt2, dt1, save, scr))
        ch.run([x.data_ptr(), t1.data_ptr(), t2.data_ptr(), y.data_ptr(), dy.data_ptr(), wt1.data_ptr(), wt2.data_ptr(),
                gamma.data_ptr(), beta.data_ptr(), ca1.data_ptr(), ca2.data_ptr(), sa.data_ptr(), save.data_ptr(), dt2.data_ptr(),
                dres.data_ptr(), dt1.data_ptr(), _ptr(dx), _ptr(_grad(gamma)), _ptr(_grad(beta)), _ptr(_grad(ca1)), _ptr(_grad(ca2)),
                _ptr(_grad(sa)), _ptr(_grad(w1)), _ptr(_grad(w2)), scr.data_ptr(), _ptr(ws), main, side])
        return (dx,) + (None,) * 8


def residual_block(x, conv1, conv2, bn, cbam):
    for c in (conv1, conv2):
        if (tuple(c.kernel_size), tuple(c.stride), tuple(c.padding)) != ((3, 3), (1, 1), (1, 1)) or c.bias is not None:
            raise RuntimeError("residual_block: 3x3 stride-1 pad-1 convs without bias (graph/encodingBlock.py:74-77)")
    ca, sa = cbam.channel_attention, cbam.spatial_attention
    return _ResidualFn.apply(x, conv1.weight, conv2.weight, bn.weight, bn.bias, ca.conv1.weight, ca.conv2.weight, sa.conv.weight, bn.eps)


# ============================================================================== conv -> InstanceNorm -> +CBAM -> ReLU
class _ConvNormCbamFn(torch.autograd.Function):
    """relu(u + CBAM(u)), u = IN(conv(x)): PoolingModule (3x3 s2, graph/encodingBlock.py:118-126) and the decoder's
    fit1 stage (1x1, graph/decoder.py:213-215)"""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, ca1, ca2, sa, eps, stride, pad):
        N, Cx, H, W = x.shape
        Cy, _, KH, KW = w.shape
        xct = HF._need_cl(x, "conv block")
        HF._cl_weight(w, "conv block")
        OH = (H + 2 * pad[0] - KH) // stride[0] + 1
        OW = (W + 2 * pad[1] - KW) // stride[1] + 1
        eng = _engine(x, Cx, Cy)
        st = HF._store(x)
        key = ("cnc-f", N, Cx, H, W, Cy, KH, KW, stride, pad, xct, eng, float(eps))
        e = _chains.get(key)
        if e is None:
            ch = Chain()
            s_x, s_wk, s_g, s_b, s_c1, s_c2, s_sa, s_t, s_y, s_save, s_ws, s_st = ch.slots(12)
            em = _Emit(ch, eng, s_ws, s_st)
            em.fwd(_desc(N, Cx, H, W, Cy, OH, OW, (KH, KW), stride, pad, xct, Cy), s_x, s_wk, None, s_t)
            ch.call("mgvae_norm_cbam_nhwc_fwd", s_t, s_g, s_b, None, 0, 0, s_c1, s_c2, s_sa, s_y, s_save, N, Cy, OH, OW, Cy, 0,
                    eps, 1, HF.ACT_RELU, 0.01, st, s_st)
            e = _chains[key] = (ch.finalize(), em.ws_bytes, int(nat.lib().mgvae_norm_cbam_nhwc_save_floats(N, Cy, OH, OW)))
        ch, wsn, nsave = e
        wk, _ = _weights(eng, w)
        t, y = _new(N, Cy, OH, OW, x), _new(N, Cy, OH, OW, x)
        save = _f32(nsave, x.device)
        ws = torch.empty(wsn, device=x.device, dtype=torch.uint8) if wsn else None
        ch.run([x.data_ptr(), wk.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ca1.data_ptr(), ca2.data_ptr(), sa.data_ptr(),
                t.data_ptr(), y.data_ptr(), save.data_ptr(), _ptr(ws), _main()])
        ctx.save_for_backward(x, w, gamma, beta, ca1, ca2, sa, t, y, save)
        ctx.cfg = (eng, xct, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, gamma, beta, ca1, ca2, sa, t, y, save = ctx.saved_tensors
        eng, xct, stride, pad = ctx.cfg
        N, Cx, H, W = x.shape
        Cy, _, KH, KW = w.shape
        _, _, OH, OW = y.shape
        st = HF._store(x)
        need_dx = ctx.needs_input_grad[0]
        flags = tuple(bool(p.requires_grad) for p in (w, gamma, beta, ca1, ca2, sa))
        key = ("cnc-b", N, Cx, H, W, Cy, KH, KW, stride, pad, xct, eng, need_dx, flags)
        e = _chains.get(key)
        if e is None:
            ch = Chain()
            (s_x, s_t, s_y, s_dy, s_wt, s_g, s_b, s_c1, s_c2, s_sa, s_save, s_dt, s_dx, s_dg, s_db, s_dc1, s_dc2, s_dsa, s_dw,
             s_scr, s_ws, s_st, s_side) = ch.slots(23)
            em = _Emit(ch, eng, s_ws, s_st)
            ch.call("mgvae_norm_cbam_nhwc_bwd", s_t, s_g, s_b, s_y, s_dy, s_c1, s_c2, s_sa, s_save, s_dt, None, s_dg, s_db, None,
                    None, s_dsa, s_scr, N, Cy, OH, OW, Cy, 0, 1, HF.ACT_RELU, 0.01, st, s_st)
            _gate_wgrad(ch, flags[3], flags[4], s_save, s_scr, s_dc1, s_dc2, N, Cy, OH, OW, s_st, s_side)
            if flags[0]:
                ch.call("mgvae_stream_fork", s_st, s_side)
                em.bwd_weight(_desc(N, Cx, H, W, Cy, OH, OW, (KH, KW), stride, pad, xct, Cy), s_x, s_dt, s_dw, s_side)
            if need_dx:
                em.bwd_data(_desc(N, Cx, H, W, Cy, OH, OW, (KH, KW), stride, pad, Cx, Cy), s_dt, s_wt, None, s_dx)
            e = _chains[key] = (ch.finalize(), em.ws_bytes, int(nat.lib().mgvae_norm_cbam_nhwc_scratch_floats(N, Cy, OH, OW)))
        ch, wsn, nscr = e
        dy = _dense_cl(dy, x, Cy)
        _, wt = _weights(eng, w)
        dt = _new(N, Cy, OH, OW, x)
        dx = _new(N, Cx, H, W, x) if need_dx else None
        scr = _f32(nscr, x.device)
        ws = torch.empty(wsn, device=x.device, dtype=torch.uint8) if wsn else None
        main, side = _side_for(N, flags[0] or flags[3] or flags[4], need_dx, (x, dt, save, scr))
        ch.run([x.data_ptr(), t.data_ptr(), y.data_ptr(), dy.data_ptr(), wt.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                ca1.data_ptr(), ca2.data_ptr(), sa.data_ptr(), save.data_ptr(), dt.data_ptr(), _ptr(dx), _ptr(_grad(gamma)),
                _ptr(_grad(beta)), _ptr(_grad(ca1)), _ptr(_grad(ca2)), _ptr(_grad(sa)), _ptr(_grad(w)), scr.data_ptr(), _ptr(ws),
                main, side])
        return (dx,) + (None,) * 9


def conv_norm_cbam_block(x, conv, bn, cbam):
    if conv.bias is not None:
        raise RuntimeError("conv_norm_cbam_block: the island's convs in front of an InstanceNorm carry no bias")
    ca, sa = cbam.channel_attention, cbam.spatial_attention
    return _ConvNormCbamFn.apply(x, conv.weight, bn.weight, bn.bias, ca.conv1.weight, ca.conv2.weight, sa.conv.weight, bn.eps,
                                 tuple(conv.stride), tuple(conv.padding))


# ================================================================================================ decoder blocks
class _DeConvFn(torch.autograd.Function):
    """DeConvModule (graph/decoder.py:91-109) and DeConvPitchPadding (:135-154; ``pp``): two transposed convs of the same
    input -> InstanceNorm (+CBAM on branch a when ``pp``) -> ReLU, side by side in one buffer -> 1x1 conv -> InstanceNorm ->
    +CBAM -> ReLU.  ``pp`` is literal about reference defect D5: ``bn2`` (ga == gb) normalises both branches."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, ga, bta, gb, btb, a_c1, a_c2, a_sa, w3, g3, b3, c1, c2, sa, eps, pp, geo_a, geo_b):
        N, Ci, h, w = x.shape
        xct = HF._need_cl(x, "decoder block")
        Co = wa.shape[1]
        for t in (wa, wb, w3):
            HF._cl_weight(t, "decoder block")
        (ka, sa_, pa, opa), (kb, sb_, pb, opb) = geo_a, geo_b
        OH = (h - 1) * sa_[0] - 2 * pa[0] + ka[0] + opa[0]
        OW = (w - 1) * sa_[1] - 2 * pa[1] + ka[1] + opa[1]
        if (OH, OW) != ((h - 1) * sb_[0] - 2 * pb[0] + kb[0] + opb[0], (w - 1) * sb_[1] - 2 * pb[1] + kb[1] + opb[1]):
            raise RuntimeError("decoder block: the two transposed convs disagree about the output size")
        eng = _engine(x, Ci, Co)
        st = HF._store(x)
        esz = x.element_size()
        L = nat.lib()
        key = ("dec-f", N, Ci, h, w, Co, geo_a, geo_b, xct, eng, bool(pp), ba is not None, bb is not None, float(eps))
        e = _chains.get(key)
        if e is None:
            ch = Chain()
            (s_x, s_wta, s_ba, s_wtb, s_bb, s_ga, s_bta, s_gb, s_btb, s_ac1, s_ac2, s_asa, s_wk3, s_g3, s_b3, s_c1, s_c2, s_sa, s_ta,
             s_tb, s_cat, s_sta, s_stb, s_t3, s_y, s_save, s_ws, s_st) = ch.slots(28)
            em = _Emit(ch, eng, s_ws, s_st)
            # transposed conv = the stride-phase data-gradient kernel: image side X = the output (Cx = Co), feature side Y = x
            em.bwd_data(_desc(N, Co, OH, OW, Ci, h, w, ka, sa_, pa, Co, xct), s_x, s_wta, s_ba if ba is not None else None, s_ta)
            if pp:
                ch.call("mgvae_norm_cbam_nhwc_fwd", s_ta, s_ga, s_bta, None, 0, 0, s_ac1, s_ac2, s_asa, s_cat, s_sta, N, Co, OH, OW,
                        2 * Co, 0, eps, 1, HF.ACT_RELU, 0.01, st, s_st)
            else:
                ch.call("mgvae_instance_norm_nhwc_fwd", s_ta, s_ga, s_bta, s_cat, s_sta, N, Co, OH, OW, 2 * Co, 0, eps, HF.ACT_RELU,
                        0.01, st, s_st)
            em.bwd_data(_desc(N, Co, OH, OW, Ci, h, w, kb, sb_, pb, Co, xct), s_x, s_wtb, s_bb if bb is not None else None, s_tb)
            ch.call("mgvae_instance_norm_nhwc_fwd", s_tb, s_gb, s_btb, s_cat + Co * esz, s_stb, N, Co, OH, OW, 2 * Co, 0, eps,
                    HF.ACT_RELU, 0.01, st, s_st)
            em.fwd(_desc(N, 2 * Co, OH, OW, Co, OH, OW, (1, 1), (1, 1), (0, 0), 2 * Co, Co), s_cat, s_wk3, None, s_t3)
            ch.call("mgvae_norm_cbam_nhwc_fwd", s_t3, s_g3, s_b3, None, 0, 0, s_c1, s_c2, s_sa, s_y, s_save, N, Co, OH, OW, Co, 0,
                    eps, 1, HF.ACT_RELU, 0.01, st, s_st)
            n_nc = int(L.mgvae_norm_cbam_nhwc_save_floats(N, Co, OH, OW))
            n_in = int(L.mgvae_instance_norm_nhwc_stats_floats(N, Co, OH, OW))
            e = _chains[key] = (ch.finalize(), em.ws_bytes, n_nc, n_in)
        ch, wsn, n_nc, n_in = e
        _, wta = _weights(eng, wa)
        _, wtb = _weights(eng, wb)
        wk3, _ = _weights(eng, w3)
        ta, tb = _new(N, Co, OH, OW, x), _new(N, Co, OH, OW, x)
        cat = _new(N, 2 * Co, OH, OW, x)
        t3, y = _new(N, Co, OH, OW, x), _new(N, Co, OH, OW, x)
        sta = _f32(n_nc if pp else n_in, x.device)
        stb = _f32(n_in, x.device)
        save = _f32(n_nc, x.device)
        ws = torch.empty(wsn, device=x.device, dtype=torch.uint8) if wsn else None
        ch.run([x.data_ptr(), wta.data_ptr(), _ptr(ba), wtb.data_ptr(), _ptr(bb), ga.data_ptr(), bta.data_ptr(), gb.data_ptr(),
                btb.data_ptr(), _ptr(a_c1), _ptr(a_c2), _ptr(a_sa), wk3.data_ptr(), g3.data_ptr(), b3.data_ptr(), c1.data_ptr(),
                c2.data_ptr(), sa.data_ptr(), ta.data_ptr(), tb.data_ptr(), cat.data_ptr(), sta.data_ptr(), stb.data_ptr(),
                t3.data_ptr(), y.data_ptr(), save.data_ptr(), _ptr(ws), _main()])
        ctx.save_for_backward(x, wa, ba, wb, bb, ga, bta, gb, btb, a_c1, a_c2, a_sa, w3, g3, b3, c1, c2, sa, ta, tb, cat, sta, stb,
                              t3, y, save)
        ctx.cfg = (eng, xct, bool(pp), geo_a, geo_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, wa, ba, wb, bb, ga, bta, gb, btb, a_c1, a_c2, a_sa, w3, g3, b3, c1, c2, sa, ta, tb, cat, sta, stb, t3, y,
         save) = ctx.saved_tensors
        eng, xct, pp, geo_a, geo_b = ctx.cfg
        N, Ci, h, w = x.shape
        _, Co, OH, OW = y.shape
        (ka, sa_, pa, _), (kb, sb_, pb, _) = geo_a, geo_b
        st = HF._store(x)
        esz = x.element_size()
        L = nat.lib()
        need_dx = ctx.needs_input_grad[0]
        params = (wa, ba, wb, bb, ga, bta, gb, btb, a_c1, a_c2, a_sa, w3, g3, b3, c1, c2, sa)
        flags = tuple(bool(p is not None and p.requires_grad) for p in params)
        key = ("dec-b", N, Ci, h, w, Co, geo_a, geo_b, xct, eng, pp, need_dx, flags)
        e = _chains.get(key)
        if e is None:
            ch = Chain()
            (s_x, s_ta, s_tb, s_cat, s_sta, s_stb, s_t3, s_y, s_save, s_dy, s_wka, s_wkb, s_wt3, s_ga, s_bta, s_gb, s_ac1, s_ac2, s_asa,
             s_g3, s_b3, s_c1, s_c2, s_sa, s_dt3, s_dcat, s_dta, s_dtb, s_dxa, s_dxb, s_dwa, s_dba, s_dwb, s_dbb, s_dga, s_dbta, s_dgb,
             s_dbtb, s_dac1, s_dac2, s_dasa, s_dw3, s_dg3, s_db3, s_dc1, s_dc2, s_dsa, s_scr3, s_scra, s_scrb, s_ws, s_st,
             s_side) = ch.slots(53)
            em = _Emit(ch, eng, s_ws, s_st)
            ch.call("mgvae_norm_cbam_nhwc_bwd", s_t3, s_g3, s_b3, s_y, s_dy, s_c1, s_c2, s_sa, s_save, s_dt3, None, s_dg3, s_db3,
                    None, None, s_dsa, s_scr3, N, Co, OH, OW, Co, 0, 1, HF.ACT_RELU, 0.01, st, s_st)
            _gate_wgrad(ch, flags[14], flags[15], s_save, s_scr3, s_dc1, s_dc2, N, Co, OH, OW, s_st, s_side)
            d3 = lambda xc: _desc(N, 2 * Co, OH, OW, Co, OH, OW, (1, 1), (1, 1), (0, 0), xc, Co)
            if flags[11]:
                ch.call("mgvae_stream_fork", s_st, s_side)
                em.bwd_weight(d3(2 * Co), s_cat, s_dt3, s_dw3, s_side)
            em.bwd_data(d3(2 * Co), s_dt3, s_wt3, None, s_dcat)
            ch.call("mgvae_instance_norm_nhwc_bwd", s_tb, s_gb, s_stb, s_cat + Co * esz, s_dcat + Co * esz, s_dtb, s_dgb, s_dbtb,
                    s_scrb, N, Co, OH, OW, 2 * Co, 0, HF.ACT_RELU, 0.01, st, s_st)
            if pp:
                ch.call("mgvae_norm_cbam_nhwc_bwd", s_ta, s_ga, s_bta, s_cat, s_dcat, s_ac1, s_ac2, s_asa, s_sta, s_dta, None, s_dga,
                        s_dbta, None, None, s_dasa, s_scra, N, Co, OH, OW, 2 * Co, 0, 1, HF.ACT_RELU, 0.01, st, s_st)
                _gate_wgrad(ch, flags[8], flags[9], s_sta, s_scra, s_dac1, s_dac2, N, Co, OH, OW, s_st, s_side)
            else:
                ch.call("mgvae_instance_norm_nhwc_bwd", s_ta, s_ga, s_sta, s_cat, s_dcat, s_dta, s_dga, s_dbta, s_scra, N, Co, OH, OW,
                        2 * Co, 0, HF.ACT_RELU, 0.01, st, s_st)
            if flags[0] or flags[1] or flags[2] or flags[3]:
                ch.call("mgvae_stream_fork", s_st, s_side)
            # weight gradient of a transposed conv: the roles of the two tensors are swapped (image side = its output)
            if flags[0]:
                em.bwd_weight(_desc(N, Co, OH, OW, Ci, h, w, ka, sa_, pa, Co, xct), s_dta, s_x, s_dwa, s_side)
            if flags[1]:
                ch.call("mgvae_channel_sum_nhwc_accum", s_dta, N * OH * OW, Co, Co, 0, s_dba, st, s_side)
            if flags[2]:
                em.bwd_weight(_desc(N, Co, OH, OW, Ci, h, w, kb, sb_, pb, Co, xct), s_dtb, s_x, s_dwb, s_side)
            if flags[3]:
                ch.call("mgvae_channel_sum_nhwc_accum", s_dtb, N * OH * OW, Co, Co, 0, s_dbb, st, s_side)
            if need_dx:
                # d/dx of a transposed conv is the forward-conv kernel
                em.fwd(_desc(N, Co, OH, OW, Ci, h, w, ka, sa_, pa, Co, Ci), s_dta, s_wka, None, s_dxa)
                em.fwd(_desc(N, Co, OH, OW, Ci, h, w, kb, sb_, pb, Co, Ci), s_dtb, s_wkb, None, s_dxb)
                ch.call("mgvae_add_inplace_typed", s_dxa, s_dxb, N * Ci * h * w, st, s_st)
            n3 = int(L.mgvae_norm_cbam_nhwc_scratch_floats(N, Co, OH, OW))
            e = _chains[key] = (ch.finalize(), em.ws_bytes, n3)
        ch, wsn, n3 = e
        dy = _dense_cl(dy, x, Co)
        wka, _ = _weights(eng, wa)
        wkb, _ = _weights(eng, wb)
        _, wt3 = _weights(eng, w3)
        dt3 = _new(N, Co, OH, OW, x)
        dcat = _new(N, 2 * Co, OH, OW, x)
        dta, dtb = _new(N, Co, OH, OW, x), _new(N, Co, OH, OW, x)
        dxa = _new(N, Ci, h, w, x) if need_dx else None
        dxb = _new(N, Ci, h, w, x) if need_dx else None
        scr3 = _f32(n3, x.device)
        scra = _f32(n3 if pp else 2 * N * Co, x.device)
        scrb = _f32(2 * N * Co, x.device)
        ws = torch.empty(wsn, device=x.device, dtype=torch.uint8) if wsn else None
        main, side = _side_for(N, any(flags[:4]) or flags[11] or flags[8] or flags[9] or flags[14] or flags[15], need_dx,
                               (x, cat, dt3, dta, dtb, save, scr3, sta, scra))
        g = _grad
        ch.run([x.data_ptr(), ta.data_ptr(), tb.data_ptr(), cat.data_ptr(), sta.data_ptr(), stb.data_ptr(), t3.data_ptr(), y.data_ptr(),
                save.data_ptr(), dy.data_ptr(), wka.data_ptr(), wkb.data_ptr(), wt3.data_ptr(), ga.data_ptr(), bta.data_ptr(),
                gb.data_ptr(), _ptr(a_c1), _ptr(a_c2), _ptr(a_sa), g3.data_ptr(), b3.data_ptr(), c1.data_ptr(), c2.data_ptr(),
                sa.data_ptr(), dt3.data_ptr(), dcat.data_ptr(), dta.data_ptr(), dtb.data_ptr(), _ptr(dxa), _ptr(dxb), _ptr(g(wa)),
                _ptr(g(ba)), _ptr(g(wb)), _ptr(g(bb)), _ptr(g(ga)), _ptr(g(bta)), _ptr(g(gb)), _ptr(g(btb)), _ptr(g(a_c1)),
                _ptr(g(a_c2)), _ptr(g(a_sa)), _ptr(g(w3)), _ptr(g(g3)), _ptr(g(b3)), _ptr(g(c1)), _ptr(g(c2)), _ptr(g(sa)),
                scr3.data_ptr(), scra.data_ptr(), scrb.data_ptr(), _ptr(ws), main, side])
        return (dxa,) + (None,) * 21


def _geo(m):
    return (tuple(m.kernel_size), tuple(m.stride), tuple(m.padding), tuple(m.output_padding))


def deconv_block(x, mod, pp):
    """``mod``: graph.decoder.DeConvModule (pp False) / DeConvPitchPadding (pp True)"""
    if pp:
        na, a = mod.bn2, mod.cbam1          # D5: bn2 on both branches, bn1 never used
        a_c = (a.channel_attention.conv1.weight, a.channel_attention.conv2.weight, a.spatial_attention.conv.weight)
        c = mod.cbam2
    else:
        na, a_c, c = mod.bn1, (None, None, None), mod.cbam
    nb = mod.bn2
    return _DeConvFn.apply(x, mod.deConv1.weight, mod.deConv1.bias, mod.deConv2.weight, mod.deConv2.bias, na.weight, na.bias,
                           nb.weight, nb.bias, a_c[0], a_c[1], a_c[2], mod.conv.weight, mod.bn3.weight, mod.bn3.bias,
                           c.channel_attention.conv1.weight, c.channel_attention.conv2.weight, c.spatial_attention.conv.weight,
                           mod.bn3.eps, bool(pp), _geo(mod.deConv1), _geo(mod.deConv2))


# ======================================================================================= NCHW ends: MLPs and stems
def _pitch2d(x):
    """row pitch (floats) of a 2-D fp32 tensor [B, K] that is dense or a column slice; else a dense copy"""
    if x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
        x = x.contiguous()
    return x, (x.stride(0) if x.shape[0] > 1 else x.shape[1])


class _MlpFn(torch.autograd.Function):
    """a stack of nn.Linear (+ bias) (+ ReLU / sigmoid): the latent discriminators (graph/z_discriminator.py:28-29,53-54)
    and the feature discriminator (graph/bar_discriminator_with_feature.py:17-25).  Every Linear is the 1x1 implicit GEMM /
    skinny GEMM of mgvae_conv2d_fwd with the activation in its epilogue, as in hipops.functional.linear."""

    @staticmethod
    def forward(ctx, x, acts, *wb):
        HF._need_cuda(x, "mlp")
        x, xct = _pitch2d(x)
        B, K = x.shape
        nl = len(acts)
        ws, bs = wb[:nl], wb[nl:]
        dims = [K] + [w.shape[0] for w in ws]
        key = ("mlp-f", B, xct, tuple(dims), acts, tuple(b is not None for b in bs))
        ch = _chains.get(key)
        if ch is None:
            ch = Chain()
            s_in = ch.slot()
            s_w, s_b, s_h = ch.slots(nl), ch.slots(nl), ch.slots(nl)
            s_st = ch.slot()
            cur, ct = s_in, xct
            for i in range(nl):
                d = _desc(B, dims[i], 1, 1, dims[i + 1], 1, 1, (1, 1), (1, 1), (0, 0), ct, dims[i + 1], acts[i], 0.01)
                ch.call("mgvae_conv2d_fwd", ch.struct(d), cur, s_w[i], s_b[i] if bs[i] is not None else None, s_h[i], s_st)
                cur, ct = s_h[i], dims[i + 1]
            ch = _chains[key] = ch.finalize()
        hs = [torch.empty((B, dims[i + 1]), device=x.device, dtype=torch.float32) for i in range(nl)]
        ch.run([x.data_ptr()] + [w.data_ptr() for w in ws] + [_ptr(b) for b in bs] + [h.data_ptr() for h in hs] + [_main()])
        ctx.save_for_backward(x, *ws, *bs, *hs)
        ctx.cfg = (acts, xct, nl)
        return hs[-1]

    @staticmethod
    def backward(ctx, dy):
        acts, xct, nl = ctx.cfg
        sv = ctx.saved_tensors
        x, ws, bs, hs = sv[0], sv[1:1 + nl], sv[1 + nl:1 + 2 * nl], sv[1 + 2 * nl:]
        B, K = x.shape
        dims = [K] + [w.shape[0] for w in ws]
        need_dx = ctx.needs_input_grad[0]
        wflags = tuple(bool(w.requires_grad) for w in ws)
        bflags = tuple(bool(b is not None and b.requires_grad) for b in bs)
        key = ("mlp-b", B, xct, tuple(dims), acts, need_dx, wflags, bflags)
        ch = _chains.get(key)
        if ch is None:
            ch = Chain()
            s_in, s_dy, s_dx = ch.slots(3)
            s_w, s_h, s_dpre, s_dh, s_dw, s_db = ch.slots(nl), ch.slots(nl), ch.slots(nl), ch.slots(nl), ch.slots(nl), ch.slots(nl)
            s_st, s_side = ch.slots(2)
            g = s_dy                                    # gradient of layer i's (activated) output
            for i in range(nl - 1, -1, -1):
                ci, co = dims[i], dims[i + 1]
                if acts[i] != HF.ACT_NONE:
                    ch.call("mgvae_act_bwd", s_h[i], g, s_dpre[i], B, co, 1, co, 0, co, 0, co, 0, acts[i], 0.01, s_st)
                    g = s_dpre[i]
                src, sct = (s_in, xct) if i == 0 else (s_h[i - 1], ci)
                if wflags[i] or bflags[i]:
                    ch.call("mgvae_stream_fork", s_st, s_side)
                if wflags[i]:
                    ch.call("mgvae_conv2d_bwd_weight", ch.struct(_desc(B, ci, 1, 1, co, 1, 1, (1, 1), (1, 1), (0, 0), sct, co)), src, g,
                            s_dw[i], s_side)
                if bflags[i]:
                    ch.call("mgvae_channel_sum_accum", g, B, co, 1, co, 0, s_db[i], s_side)
                if i > 0 or need_dx:
                    dst = s_dx if i == 0 else s_dh[i - 1]
                    ch.call("mgvae_conv2d_bwd_data", ch.struct(_desc(B, ci, 1, 1, co, 1, 1, (1, 1), (1, 1), (0, 0), ci, co)), g, s_w[i],
                            None, dst, s_st)
                    g = dst
            ch = _chains[key] = ch.finalize()
        dy = dy.contiguous()
        dev = x.device
        dpre = [torch.empty((B, dims[i + 1]), device=dev, dtype=torch.float32) if acts[i] != HF.ACT_NONE else None for i in range(nl)]
        dh = [torch.empty((B, dims[i + 1]), device=dev, dtype=torch.float32) for i in range(nl - 1)] + [None]
        dx = torch.empty((B, K), device=dev, dtype=torch.float32) if need_dx else None
        main, side = _side_for(B, any(wflags) or any(bflags), need_dx, (x,) + tuple(hs) + tuple(t for t in dpre if t is not None) + (dy,))
        ch.run([x.data_ptr(), dy.data_ptr(), _ptr(dx)] + [w.data_ptr() for w in ws] + [h.data_ptr() for h in hs] + [_ptr(t) for t in dpre]
               + [_ptr(t) for t in dh] + [_ptr(_grad(w)) for w in ws] + [_ptr(_grad(b)) for b in bs] + [main, side])
        return (dx, None) + (None,) * (2 * nl)


def mlp(x, layers):
    """``layers``: [(weight [Cout, Cin], bias or None, activation)]; x [B, Cin] -> [B, Cout of the last layer]"""
    ws = tuple(l[0] for l in layers)
    bs = tuple(l[1] for l in layers)
    return _MlpFn.apply(x, tuple(int(l[2]) for l in layers), *ws, *bs)


def mlp_usable(x):
    return ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and HF.get_nchw_operand_dtype() == "f32" and not HF.USE_DIRECT


class _TrunkEntryFn(torch.autograd.Function):
    """Both stems of an encoder trunk (graph/encodingBlock.py:25-36,56-67: thin conv -> LeakyReLU -> thin conv -> InstanceNorm ->
    +CBAM -> LeakyReLU, each writing its half of the concat of graph/encoder.py:27-29) as one node, inside the island:
    [N,1,H,W] fp32 -> channels-last [N,64,H/2,W/2] of the island's storage type.  The one-channel conv writes channels-last
    itself (mgvae_conv2d_c1_nhwc_fwd), the 32 -> 32 conv is the island's matrix kernel, InstanceNorm / CBAM the channels-last
    ones; the stems compute in fp32 storage (a bf16 island begins with one cast of the concat)."""

    @staticmethod
    def forward(ctx, x, eps, geo, dtype, *prm):
        HF._need_cuda(x, "trunk entry")
        x = x.contiguous()
        N, _, H, W = x.shape
        OH, OW = H // 2, W // 2
        eng = "x3" if HF.FP32_ENGINE == "x3" else "f32"
        key = ("entry-f", N, H, W, geo, dtype, eng, float(eps))
        e = _chains.get(key)
        L = nat.lib()
        if e is None:
            ch = Chain()
            s_x, s_cat, s_ycl, s_ws, s_st = ch.slots(5)
            em = _Emit(ch, eng, s_ws, s_st)
            stems = []
            for si in range(2):
                (k1, s1, p1), (k2, s2, p2) = geo[si]
                H1, W1 = (H + 2 * p1[0] - k1[0]) // s1[0] + 1, (W + 2 * p1[1] - k1[1]) // s1[1] + 1
                if ((H1 + 2 * p2[0] - k2[0]) // s2[0] + 1, (W1 + 2 * p2[1] - k2[1]) // s2[1] + 1) != (OH, OW):
                    raise RuntimeError("trunk entry: a stem does not halve the map")
                s_w1, s_wk2, s_g, s_b, s_c1, s_c2, s_sa, s_t1, s_t2, s_save = ch.slots(10)
                stems.append((H1, W1))
                ch.call("mgvae_conv2d_c1_nhwc_fwd", ch.struct(_desc(N, 1, H, W, 32, H1, W1, k1, s1, p1, 1, 32, HF.ACT_LEAKY, 0.01)), s_x, s_w1,
                        s_t1, HF.STORE_F32, s_st)
                em.fwd(_desc(N, 32, H1, W1, 32, OH, OW, k2, s2, p2, 32, 32), s_t1, s_wk2, None, s_t2)
                ch.call("mgvae_norm_cbam_nhwc_fwd", s_t2, s_g, s_b, None, 0, 0, s_c1, s_c2, s_sa, s_cat, s_save, N, 32, OH, OW, 64, 32 * si,
                        eps, 1, HF.ACT_LEAKY, 0.01, HF.STORE_F32, s_st)
            if dtype == torch.bfloat16:
                ch.call("mgvae_cast_storage", s_cat, HF.STORE_F32, s_ycl, HF.STORE_BF16, N * 64 * OH * OW, s_st)
            e = _chains[key] = (ch.finalize(), stems, em.ws_bytes, int(L.mgvae_norm_cbam_nhwc_save_floats(N, 32, OH, OW)))
        ch, stems, wsn, nsave = e
        dev = x.device
        cat = HF.new_channels_last(N, 64, OH, OW, dev, torch.float32)
        ycl = HF.new_channels_last(N, 64, OH, OW, dev, dtype) if dtype == torch.bfloat16 else cat
        ws = torch.empty(wsn, device=dev, dtype=torch.uint8) if wsn else None
        addr = [x.data_ptr(), cat.data_ptr(), ycl.data_ptr(), _ptr(ws), _main()]
        keep = []
        for si in range(2):
            w1, w2, g, b, c1, c2, sa = prm[7 * si:7 * si + 7]
            HF._cl_weight(w2, "trunk entry")
            H1, W1 = stems[si]
            wk2, _ = _weights(eng, w2)
            t1, t2, save = HF.new_channels_last(N, 32, H1, W1, dev, torch.float32), HF.new_channels_last(N, 32, OH, OW, dev, torch.float32), _f32(nsave, dev)
            keep += [t1, t2, save]
            addr += [w1.data_ptr(), wk2.data_ptr(), g.data_ptr(), b.data_ptr(), c1.data_ptr(), c2.data_ptr(), sa.data_ptr(), t1.data_ptr(),
                     t2.data_ptr(), save.data_ptr()]
        ch.run(addr)              # slot order: the five shared slots first, then 10 per stem
        ctx.save_for_backward(x, cat, *prm, *keep)
        ctx.cfg = (float(eps), geo, stems, dtype, eng)
        return ycl

    @staticmethod
    def backward(ctx, dy):
        eps, geo, stems, dtype, eng = ctx.cfg
        sv = ctx.saved_tensors
        x, cat, prm, keep = sv[0], sv[1], sv[2:16], sv[16:]
        N, _, H, W = x.shape
        OH, OW = H // 2, W // 2
        flags = tuple(bool(p.requires_grad) for p in prm)
        key = ("entry-b", N, H, W, geo, dtype, eng, flags)
        e = _chains.get(key)
        L = nat.lib()
        if e is None:
            ch = Chain()
            s_x, s_cat, s_dy, s_dcat, s_ws, s_st, s_side = ch.slots(7)
            em = _Emit(ch, eng, s_ws, s_st)
            if dtype == torch.bfloat16:
                ch.call("mgvae_cast_storage", s_dy, HF.STORE_BF16, s_dcat, HF.STORE_F32, N * 64 * OH * OW, s_st)
            for si in range(2):
                (k1, s1, p1), (k2, s2, p2) = geo[si]
                H1, W1 = stems[si]
                fl = flags[7 * si:7 * si + 7]
                (s_wt2, s_g, s_b, s_c1, s_c2, s_sa, s_t1, s_t2, s_save, s_dt2, s_dt1, s_scr, s_dw1, s_dw2, s_dg, s_db, s_dc1, s_dc2,
                 s_dsa) = ch.slots(19)
                ch.call("mgvae_norm_cbam_nhwc_bwd", s_t2, s_g, s_b, s_cat, s_dcat, s_c1, s_c2, s_sa, s_save, s_dt2, None, s_dg, s_db, None,
                        None, s_dsa, s_scr, N, 32, OH, OW, 64, 32 * si, 1, HF.ACT_LEAKY, 0.01, HF.STORE_F32, s_st)
                _gate_wgrad(ch, fl[4], fl[5], s_save, s_scr, s_dc1, s_dc2, N, 32, OH, OW, s_st, s_side)
                d2 = _desc(N, 32, H1, W1, 32, OH, OW, k2, s2, p2, 32, 32)
                if fl[1]:
                    ch.call("mgvae_stream_fork", s_st, s_side)
                    em.bwd_weight(d2, s_t1, s_dt2, s_dw2, s_side)
                if fl[0]:
                    # the data gradient of the second conv applies the first conv's LeakyReLU' while it stores; it only feeds dw1
                    em.bwd_data(d2, s_dt2, s_wt2, None, s_dt1, em.mask(s_t1, 32, HF.ACT_LEAKY, 0.01))
                    ch.call("mgvae_stream_fork", s_st, s_side)
                    ch.call("mgvae_conv2d_c1_nhwc_bwd_weight", ch.struct(_desc(N, 1, H, W, 32, H1, W1, k1, s1, p1, 1, 32, HF.ACT_LEAKY, 0.01)), s_x,
                            s_dt1, None, s_dw1, HF.STORE_F32, s_side)
            e = _chains[key] = (ch.finalize(), em.ws_bytes, int(L.mgvae_norm_cbam_nhwc_scratch_floats(N, 32, OH, OW)))
        ch, wsn, nscr = e
        dev = x.device
        if dy.dtype != dtype:
            dy = dy.to(dtype)
        if HF.cl_pitch(dy) != 64:
            dy = dy.contiguous(memory_format=HF.CL)
        dcat = HF.new_channels_last(N, 64, OH, OW, dev, torch.float32) if dtype == torch.bfloat16 else dy
        ws = torch.empty(wsn, device=dev, dtype=torch.uint8) if wsn else None
        touched = [x, dcat]
        addr_stems, tmp = [], []
        for si in range(2):
            w1, w2, g, b, c1, c2, sa = prm[7 * si:7 * si + 7]
            t1, t2, save = keep[3 * si:3 * si + 3]
            H1, W1 = stems[si]
            _, wt2 = _weights(eng, w2)
            dt2, dt1, scr = HF.new_channels_last(N, 32, OH, OW, dev, torch.float32), HF.new_channels_last(N, 32, H1, W1, dev, torch.float32), _f32(nscr, dev)
            tmp += [dt2, dt1, scr]
            touched += [t1, dt2, dt1, save, scr]
            addr_stems += [wt2.data_ptr(), g.data_ptr(), b.data_ptr(), c1.data_ptr(), c2.data_ptr(), sa.data_ptr(), t1.data_ptr(), t2.data_ptr(),
                           save.data_ptr(), dt2.data_ptr(), dt1.data_ptr(), scr.data_ptr(), _ptr(_grad(w1)), _ptr(_grad(w2)), _ptr(_grad(g)),
                           _ptr(_grad(b)), _ptr(_grad(c1)), _ptr(_grad(c2)), _ptr(_grad(sa))]
        main, side = _side_for(N, any(flags), True, touched)
        ch.run([x.data_ptr(), cat.data_ptr(), dy.data_ptr(), dcat.data_ptr(), _ptr(ws), main, side] + addr_stems)
        return (None,) * (4 + 14)


def trunk_entry(x, trunk):
    """``trunk``: graph.encoder._ConvTrunk; its two stems in concat order (pitch_time -> channels 0..31, time_pitch -> 32..63)"""
    prm, geo = [], []
    for stem in (trunk.pitch_time, trunk.time_pitch):
        a, b = getattr(stem, stem.first), getattr(stem, stem.second)
        ca, sa = stem.cbam.channel_attention, stem.cbam.spatial_attention
        prm += [a.weight, b.weight, stem.bn.weight, stem.bn.bias, ca.conv1.weight, ca.conv2.weight, sa.conv.weight]
        geo.append(((tuple(a.kernel_size), tuple(a.stride), tuple(a.padding)), (tuple(b.kernel_size), tuple(b.stride), tuple(b.padding))))
    return _TrunkEntryFn.apply(x, trunk.pitch_time.bn.eps, tuple(geo), HF.island_dtype(), *prm)


def entry_usable(x):
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 1 and not x.requires_grad
            and HF.get_nchw_operand_dtype() == "f32" and not HF.USE_DIRECT)


# ============================================================================ decoder: head + stems + layout change
class _DecoderFrontFn(torch.autograd.Function):
    """graph/decoder.py:192-211 as one node: the two Linear + ReLU + Dropout branches over (phrase feature, position embedding)
    and (z, pre_z), their concat, the two stems (a non-overlapping transposed conv of the 1x1 map = a plain GEMM, a second
    transposed conv, InstanceNorm, +CBAM, ReLU) side by side, and the layout change into the channels-last island:
    (z, pre_z, phrase_feature [B,1152], position [B]) -> [B, 2048, 6, 3] channels-last of the island's storage type.
    ``mode``: 0 no dropout (eval), 1 Philox dropout (seed / offsets passed in), 2 the two given masks (tests)."""

    @staticmethod
    def forward(ctx, z, pre_z, pf, position, mode, p_drop, seed, offs, dtype, eps, geo, m0, m1, table, Wp, bp, Wb, bb, *stems):
        HF._need_cuda(z, "decoder front")
        z, zpt = _pitch2d(z)
        pre_z, pzpt = _pitch2d(pre_z)
        pf, pfpt = _pitch2d(pf)
        position = position.to(device=z.device, dtype=torch.int64).contiguous()
        B = z.shape[0]
        st = HF.STORE_BF16 if dtype == torch.bfloat16 else HF.STORE_F32
        L = nat.lib()
        tw = bool(HF.USE_TRANSPOSED_W)
        key = ("dfront-f", B, zpt, pzpt, pfpt, mode, geo, st, float(eps), float(p_drop), tw)
        e = _chains.get(key)
        NC, P = B * 1024, 18
        if e is None:
            ch = Chain()
            (s_z, s_pz, s_pf, s_pos, s_seed, s_o0, s_o1, s_m0, s_m1, s_tab, s_Wp, s_bp, s_Wb, s_bb, s_pbuf, s_bbuf, s_xbuf, s_hp, s_hb, s_dp,
             s_db, s_mk0, s_mk1, s_cat, s_ocl, s_st) = ch.slots(26)
            n = B * 1152
            lin = _desc(B, 2304, 1, 1, 1152, 1, 1, (1, 1), (1, 1), (0, 0), 2304, 1152, HF.ACT_RELU, 0.01)
            ch.call("mgvae_copy2d", s_pbuf, 2304, s_pf, pfpt, 1152, B, s_st)
            ch.call("mgvae_embedding_fwd", s_pos, s_tab, s_pbuf + 4 * 1152, B, 1152, 332, 2304, s_st)
            ch.call("mgvae_conv2d_fwd", ch.struct(lin), s_pbuf, s_Wp, s_bp, s_hp, s_st)
            if mode == 1:
                ch.call("mgvae_dropout_fwd", s_hp, s_dp, s_mk0, n, p_drop, s_seed, s_o0, s_st)
            elif mode == 2:
                ch.call("mgvae_mul", s_hp, s_m0, s_dp, n, s_st)
            ch.call("mgvae_copy2d", s_bbuf, 2304, s_z, zpt, 1152, B, s_st)
            ch.call("mgvae_copy2d", s_bbuf + 4 * 1152, 2304, s_pz, pzpt, 1152, B, s_st)
            ch.call("mgvae_conv2d_fwd", ch.struct(lin), s_bbuf, s_Wb, s_bb, s_hb, s_st)
            if mode == 1:
                ch.call("mgvae_dropout_fwd", s_hb, s_db, s_mk1, n, p_drop, s_seed, s_o1, s_st)
            elif mode == 2:
                ch.call("mgvae_mul", s_hb, s_m1, s_db, n, s_st)
            ch.call("mgvae_copy2d", s_xbuf, 2304, s_db if mode else s_hb, 1152, 1152, B, s_st)
            ch.call("mgvae_copy2d", s_xbuf + 4 * 1152, 2304, s_dp if mode else s_hp, 1152, 1152, B, s_st)
            shapes = []
            for si in range(2):
                (k1, k2) = geo[si]
                s_w1, s_w2, s_g, s_b, s_c1, s_c2, s_sa, s_t1, s_t2, s_wt, s_u, s_stats, s_save = ch.slots(13)
                co1 = 1024 * k1[0] * k1[1]
                h1, w1_ = k1
                if (h1 * k2[0], w1_ * k2[1]) != (6, 3):
                    raise RuntimeError("decoder stem does not produce a 6 x 3 map")
                shapes.append((co1, h1, w1_))
                # 1x1 map through a non-overlapping transposed conv = the GEMM y[n, (co,kh,kw)] = x[n, ci] . w[ci, (co,kh,kw)]
                ch.call("mgvae_conv2d_bwd_data", ch.struct(_desc(B, co1, 1, 1, 2304, 1, 1, (1, 1), (1, 1), (0, 0), co1, 2304, HF.ACT_RELU, 0.01)),
                        s_xbuf, s_w1, None, s_t1, s_st)
                d2 = _desc(B, 1024, 6, 3, 1024, h1, w1_, k2, k2, (0, 0), 1024, 1024, HF.ACT_NONE, 0.01)
                if tw and k2[0] * k2[1] > 1:
                    ch.call("mgvae_weight_transpose", s_w2, s_wt, 1024, 1024, k2[0] * k2[1], s_st)
                    ch.call("mgvae_conv2d_bwd_data_tw", ch.struct(d2), s_t1, s_wt, None, s_t2, s_st)
                else:
                    ch.call("mgvae_conv2d_bwd_data", ch.struct(d2), s_t1, s_w2, None, s_t2, s_st)
                ch.call("mgvae_instance_norm_fwd", s_t2, s_g, s_b, s_u, s_stats, B, 1024, P, 1024, 0, eps, HF.ACT_NONE, 0.0, s_save + 4 * NC,
                        s_save + 8 * NC, s_save + 12 * NC, s_st)
                ch.call("mgvae_cbam_fwd", s_u, None, s_c1, s_c2, s_sa, s_cat + 4 * si * 1024 * P, s_save, B, 1024, 6, 3, 2048, 0, 1, HF.ACT_RELU,
                        0.01, 3 | 4, s_st)
            ch.call("mgvae_layout_nchw_to_nhwc", s_cat, s_ocl, B, 2048, P, 2048, 0, 2048, 0, st, s_st)
            e = _chains[key] = (ch.finalize(), shapes, int(L.mgvae_cbam_save_floats(B, 1024, 6, 3)))
        ch, shapes, nsave = e
        dev = z.device
        f = lambda *s_: torch.empty(s_, device=dev, dtype=torch.float32)
        pbuf, bbuf, xbuf, hp, hb = f(B, 2304), f(B, 2304), f(B, 2304), f(B, 1152), f(B, 1152)
        dp = f(B, 1152) if mode else None
        db = f(B, 1152) if mode else None
        mk0 = f(B, 1152) if mode == 1 else None
        mk1 = f(B, 1152) if mode == 1 else None
        cat = f(B, 2048, 6, 3)
        ocl = HF.new_channels_last(B, 2048, 6, 3, dev, dtype)
        addr = [z.data_ptr(), pre_z.data_ptr(), pf.data_ptr(), position.data_ptr(), int(seed), int(offs[0]), int(offs[1]), _ptr(m0), _ptr(m1),
                table.data_ptr(), Wp.data_ptr(), bp.data_ptr(), Wb.data_ptr(), bb.data_ptr(), pbuf.data_ptr(), bbuf.data_ptr(), xbuf.data_ptr(),
                hp.data_ptr(), hb.data_ptr(), _ptr(dp), _ptr(db), _ptr(mk0), _ptr(mk1), cat.data_ptr(), ocl.data_ptr(), _main()]
        keep = []
        for si in range(2):
            w1, w2, g, b, c1, c2, sa = stems[7 * si:7 * si + 7]
            co1, h1, w1_ = shapes[si]
            t1, t2, wt, u, stats, save = f(B, co1), f(B, 1024, 6, 3), f(w2.numel()), f(B, 1024, 6, 3), f(2 * NC), f(nsave)
            keep += [t1, t2, u, stats, save]
            addr += [w1.data_ptr(), w2.data_ptr(), g.data_ptr(), b.data_ptr(), c1.data_ptr(), c2.data_ptr(), sa.data_ptr(), t1.data_ptr(),
                     t2.data_ptr(), wt.data_ptr(), u.data_ptr(), stats.data_ptr(), save.data_ptr()]
        ch.run(addr)
        masks = (mk0, mk1) if mode == 1 else ((m0, m1) if mode == 2 else (None, None))
        ctx.save_for_backward(position, table, Wp, bp, Wb, bb, pbuf, bbuf, xbuf, hp, hb, masks[0], masks[1], cat, *stems, *keep)
        ctx.cfg = (mode, st, geo, shapes, dtype, tw)
        return ocl

    @staticmethod
    def backward(ctx, dy):
        mode, st, geo, shapes, dtype, tw = ctx.cfg
        sv = ctx.saved_tensors
        position, table, Wp, bp, Wb, bb, pbuf, bbuf, xbuf, hp, hb, mk0, mk1, cat = sv[:14]
        stems, keep = sv[14:28], sv[28:]
        B = xbuf.shape[0]
        NC, P, NP = B * 1024, 18, B * 18
        L = nat.lib()
        flags = tuple(bool(p.requires_grad) for p in (table, Wp, bp, Wb, bb) + tuple(stems))
        need = tuple(bool(v) for v in ctx.needs_input_grad[:3])
        key = ("dfront-b", B, mode, geo, st, flags, tw)
        e = _chains.get(key)
        if e is None:
            ch = Chain()
            (s_pos, s_Wp, s_Wb, s_pbuf, s_bbuf, s_xbuf, s_hp, s_hb, s_mk0, s_mk1, s_cat, s_dy, s_dcat, s_dx0, s_dx1, s_gb, s_gb2, s_dpre_b, s_dbbuf,
             s_gp, s_gp2, s_dpre_p, s_dpbuf, s_dtab, s_dWp, s_dbp, s_dWb, s_dbb, s_st, s_side) = ch.slots(30)
            ch.call("mgvae_layout_nhwc_to_nchw", s_dy, s_dcat, B, 2048, P, 2048, 0, 2048, 0, st, s_st)
            dxs = (s_dx0, s_dx1)
            for si in range(2):
                (k1, k2) = geo[si]
                co1, h1, w1_ = shapes[si]
                fl = flags[5 + 7 * si:5 + 7 * si + 7]
                (s_w1, s_w2, s_g, s_b, s_c1, s_c2, s_sa, s_t1, s_t2, s_u, s_stats, s_save, s_du, s_dt2, s_dt1, s_scr, s_dw1, s_dw2, s_dg, s_db_,
                 s_dc1, s_dc2, s_dsa) = ch.slots(23)
                off = 4 * si * 1024 * P
                ch.call("mgvae_cbam_bwd", s_u, s_cat + off, s_dcat + off, s_c1, s_c2, s_sa, s_save, s_du, None, s_dc1, s_dc2, s_dsa, s_scr, B, 1024,
                        6, 3, 2048, 0, 1, HF.ACT_RELU, 0.01, 3 | 4, s_st)
                ch.call("mgvae_instance_norm_bwd", s_t2, s_g, s_b, s_stats, s_du, s_dt2, s_dg, s_db_, B, 1024, P, 1024, 0, HF.ACT_NONE, 0.0,
                        s_scr + 4 * (3 * NP + NC), s_scr + 4 * (3 * NP + 2 * NC), s_save + 12 * NC, s_st)
                # second transposed conv: weight gradient with the roles swapped, d/dx = the forward-conv kernel, which also applies
                # relu'(t1) (the first transposed conv skipped its own activation-gradient pass)
                if fl[1]:
                    ch.call("mgvae_stream_fork", s_st, s_side)
                    ch.call("mgvae_conv2d_bwd_weight", ch.struct(_desc(B, 1024, 6, 3, 1024, h1, w1_, k2, k2, (0, 0), 1024, 1024)), s_dt2, s_t1,
                            s_dw2, s_side)
                m = ch.struct(nat.ActMask(0, 1024, 0, HF.ACT_RELU, 0.0), (("src", s_t1),))
                ch.call("mgvae_conv2d_fwd_masked", ch.struct(_desc(B, 1024, 6, 3, 1024, h1, w1_, k2, k2, (0, 0), 1024, 1024)), s_dt2, s_w2, None,
                        s_dt1, m, s_st)
                if fl[0]:
                    ch.call("mgvae_stream_fork", s_st, s_side)
                    ch.call("mgvae_conv2d_bwd_weight", ch.struct(_desc(B, co1, 1, 1, 2304, 1, 1, (1, 1), (1, 1), (0, 0), co1, 2304)), s_dt1, s_xbuf,
                            s_dw1, s_side)
                ch.call("mgvae_conv2d_fwd", ch.struct(_desc(B, co1, 1, 1, 2304, 1, 1, (1, 1), (1, 1), (0, 0), co1, 2304)), s_dt1, s_w1, None,
                        dxs[si], s_st)
            ch.call("mgvae_add_inplace", s_dx0, s_dx1, B * 2304, s_st)
            n = B * 1152
            lin_w = _desc(B, 2304, 1, 1, 1152, 1, 1, (1, 1), (1, 1), (0, 0), 2304, 1152)
            for (col, s_g1, s_g2, s_mk, s_h, s_dpre, s_in, s_W, s_dW, s_dbias, s_dbuf, fw, fb) in (
                    (0, s_gb, s_gb2, s_mk1, s_hb, s_dpre_b, s_bbuf, s_Wb, s_dWb, s_dbb, s_dbbuf, flags[3], flags[4]),
                    (1152, s_gp, s_gp2, s_mk0, s_hp, s_dpre_p, s_pbuf, s_Wp, s_dWp, s_dbp, s_dpbuf, flags[1], flags[2])):
                ch.call("mgvae_copy2d", s_g1, 1152, s_dx0 + 4 * col, 2304, 1152, B, s_st)
                g = s_g1
                if mode:
                    ch.call("mgvae_mul", s_g1, s_mk, s_g2, n, s_st)
                    g = s_g2
                ch.call("mgvae_act_bwd", s_h, g, s_dpre, B, 1152, 1, 1152, 0, 1152, 0, 1152, 0, HF.ACT_RELU, 0.01, s_st)
                if fw or fb:
                    ch.call("mgvae_stream_fork", s_st, s_side)
                if fw:
                    ch.call("mgvae_conv2d_bwd_weight", ch.struct(lin_w), s_in, s_dpre, s_dW, s_side)
                if fb:
                    ch.call("mgvae_channel_sum_accum", s_dpre, B, 1152, 1, 1152, 0, s_dbias, s_side)
                ch.call("mgvae_conv2d_bwd_data", ch.struct(lin_w), s_dpre, s_W, None, s_dbuf, s_st)
            if flags[0]:
                ch.call("mgvae_embedding_bwd", s_pos, s_dpbuf + 4 * 1152, s_dtab, B, 1152, 332, 2304, s_st)
            e = _chains[key] = (ch.finalize(), int(L.mgvae_cbam_bwd_scratch_floats(B, 1024, 6, 3)))
        ch, nscr = e
        dev = xbuf.device
        if dy.dtype != dtype:
            dy = dy.to(dtype)
        if HF.cl_pitch(dy) != 2048:
            dy = dy.contiguous(memory_format=HF.CL)
        f = lambda *s_: torch.empty(s_, device=dev, dtype=torch.float32)
        dcat, dx0, dx1 = f(B, 2048, 6, 3), f(B, 2304), f(B, 2304)
        gb, gb2, dpre_b, dbbuf = f(B, 1152), (f(B, 1152) if mode else None), f(B, 1152), f(B, 2304)
        gp, gp2, dpre_p, dpbuf = f(B, 1152), (f(B, 1152) if mode else None), f(B, 1152), f(B, 2304)
        touched = [xbuf, bbuf, pbuf, dpre_b, dpre_p, dcat]
        addr_st = []
        tmp = []
        for si in range(2):
            w1, w2, g, b, c1, c2, sa = stems[7 * si:7 * si + 7]
            t1, t2, u, stats, save = keep[5 * si:5 * si + 5]
            co1, h1, w1_ = shapes[si]
            du, dt2, dt1, scr = f(B, 1024, 6, 3), f(B, 1024, 6, 3), f(B, co1), f(nscr)
            tmp += [du, dt2, dt1, scr]
            touched += [t1, dt2, dt1, save, scr]
            addr_st += [w1.data_ptr(), w2.data_ptr(), g.data_ptr(), b.data_ptr(), c1.data_ptr(), c2.data_ptr(), sa.data_ptr(), t1.data_ptr(),
                        t2.data_ptr(), u.data_ptr(), stats.data_ptr(), save.data_ptr(), du.data_ptr(), dt2.data_ptr(), dt1.data_ptr(), scr.data_ptr(),
                        _ptr(_grad(w1)), _ptr(_grad(w2)), _ptr(_grad(g)), _ptr(_grad(b)), _ptr(_grad(c1)), _ptr(_grad(c2)), _ptr(_grad(sa))]
        main, side = _side_for(B, any(flags), True, touched)
        ch.run([position.data_ptr(), Wp.data_ptr(), Wb.data_ptr(), pbuf.data_ptr(), bbuf.data_ptr(), xbuf.data_ptr(), hp.data_ptr(), hb.data_ptr(),
                _ptr(mk0), _ptr(mk1), cat.data_ptr(), dy.data_ptr(), dcat.data_ptr(), dx0.data_ptr(), dx1.data_ptr(), gb.data_ptr(), _ptr(gb2),
                dpre_b.data_ptr(), dbbuf.data_ptr(), gp.data_ptr(), _ptr(gp2), dpre_p.data_ptr(), dpbuf.data_ptr(), _ptr(_grad(table)),
                _ptr(_grad(Wp)), _ptr(_grad(bp)), _ptr(_grad(Wb)), _ptr(_grad(bb)), main, side] + addr_st)
        dz = dbbuf[:, :1152] if need[0] else None
        dpz = dbbuf[:, 1152:] if need[1] else None
        dpf = dpbuf[:, :1152] if need[2] else None
        return (dz, dpz, dpf) + (None,) * (15 + len(stems))


def decoder_front(dec, z, pre_z, phrase_feature, position):
    """``dec``: graph.decoder.Decoder"""
    masks = dec._drop_masks
    if masks is not None:
        mode, m0, m1, seed, offs = 2, masks[0].contiguous(), masks[1].contiguous(), 0, (0, 0)
    elif dec.training and dec.dropout_p > 0.0:
        mode, m0, m1, seed = 1, None, None, HF._rng_state["seed"]
        offs = (HF._next_offset(), HF._next_offset())          # phrase branch first, then the bar branch (graph/decoder.py:196,201)
    else:
        mode, m0, m1, seed, offs = 0, None, None, 0, (0, 0)
    stems, geo = [], []
    for stem in (dec.pitch, dec.time):                          # concat order: pitch -> channels 0..1023, time -> 1024..2047
        a, b = getattr(stem, stem.first), getattr(stem, stem.second)
        ca, sa = stem.cbam.channel_attention, stem.cbam.spatial_attention
        stems += [a.weight, b.weight, stem.bn.weight, stem.bn.bias, ca.conv1.weight, ca.conv2.weight, sa.conv.weight]
        geo.append((tuple(a.kernel_size), tuple(b.kernel_size)))
    return _DecoderFrontFn.apply(z, pre_z, phrase_feature, position, mode, float(dec.dropout_p), seed, offs, HF.island_dtype(),
                                 dec.pitch.bn.eps, tuple(geo), m0, m1, dec.position_embedding.weight, dec.phrase_linear.weight,
                                 dec.phrase_linear.bias, dec.bar_linear.weight, dec.bar_linear.bias, *stems)


def front_usable(z):
    return ENABLED and z.is_cuda and z.dtype == torch.float32 and HF.get_nchw_operand_dtype() == "f32" and not HF.USE_DIRECT and HF.DEFER_ACT_GRAD
