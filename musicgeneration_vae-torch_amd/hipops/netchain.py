"""A whole small NCHW network as ONE autograd node and one launch chain per direction.

hipops/blocks.py states the island's blocks call by call.  The bar-pair discriminator (reference
graph/bar_discriminator.py:7-217: three towers of thin convs and BatchNorms, ~45 ops per pass, three passes forward and up
to three backward per GAN iteration, every kernel a few microseconds) is too long for that style, so this module emits its
chains from a description of the network instead: ``NetBuilder`` offers the handful of NCHW fp32 ops the tower uses -- conv
(+ fused activation), BatchNorm2d (+ activation), pitch-axis group sum, whole-map average, a copy into a column slice,
Linear -- and each op appends its forward call to the forward chain and a closure that will append its backward calls;
the backward chain is those closures run in reverse.  Temporaries live in ONE arena per direction (static offsets inside
one allocation per run), so a run patches a few dozen addresses, not hundreds.  The calls are the entry points of
include/mgvae.h that hipops/functional.py issues for the same ops, in the same order forward; backward the order within
the node is the reverse of forward (autograd's order for a chain), sums of several consumers' input gradients use
mgvae_add_inplace.
"""
import torch

from . import _native as nat
from . import functional as HF
from .chain import Chain

_ALIGN = 256


class T:
    """a tensor inside the chains: ``where`` in {"x", "fa" (forward arena), "ba" (backward arena), "y"}, byte offset,
    logical NCHW shape, channel pitch ``ctot`` (> C: a channel slice of a wider buffer)"""
    __slots__ = ("where", "off", "N", "C", "H", "W", "ctot", "needs_grad", "grad")

    def __init__(self, where, off, N, C, H, W, ctot=None, needs_grad=True):
        self.where, self.off, self.N, self.C, self.H, self.W = where, off, N, C, H, W
        self.ctot = C if ctot is None else ctot
        self.needs_grad = needs_grad
        self.grad = None            # T of the gradient (backward arena), set while the backward chain is being emitted

    @property
    def numel(self):
        return self.N * self.C * self.H * self.W


class NetBuilder:
    def __init__(self, N, param_names, trainable, need_dx):
        self.N = N
        self.f, self.b = Chain(), Chain()
        self.names = list(param_names)
        self.trainable = dict(zip(param_names, trainable))
        self.need_dx = need_dx
        # forward slots: x, arena, y, stream, then one per parameter / buffer
        self.fs = {"x": self.f.slot(), "fa": self.f.slot(), "y": self.f.slot(), "st": self.f.slot()}
        for n in self.names:
            self.fs["p:" + n] = self.f.slot()
        # backward slots: x, forward arena, backward arena, y, dy, dx, streams, parameters, gradients
        self.bs = {k: self.b.slot() for k in ("x", "fa", "ba", "y", "dy", "dx", "st", "side")}
        for n in self.names:
            self.bs["p:" + n] = self.b.slot()
            self.bs["g:" + n] = self.b.slot()
        self.fa_bytes = 0
        self.ba_bytes = 0
        self.bwd = []

    # ------------------------------------------------------------------ addressing
    def _alloc(self, which, nbytes):
        off = getattr(self, which + "_bytes")
        setattr(self, which + "_bytes", (off + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN)
        return off

    def new(self, C, H, W, where="fa", needs_grad=True):
        return T(where, self._alloc(where, 4 * self.N * C * H * W), self.N, C, H, W, needs_grad=needs_grad)

    def fp(self, t):
        return self.fs[t.where] + t.off

    def bp(self, t):
        return self.bs[t.where] + t.off

    def _acc(self, t, g):
        """``g`` is one consumer's contribution to the gradient of ``t``"""
        if t.grad is None:
            t.grad = g
        else:
            if t.grad.ctot != t.grad.C or g.ctot != g.C:
                raise RuntimeError("gradient accumulation needs dense tensors")
            self.b.call("mgvae_add_inplace", self.bp(t.grad), self.bp(g), t.numel, self.bs["st"])

    @staticmethod
    def _desc(N, Cx, H, W, Cy, OH, OW, k, s, p, xct, yct, act=HF.ACT_NONE):
        return nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], xct, 0, yct, 0, act, 0.01)

    # ------------------------------------------------------------------ ops
    def conv(self, x, wname, Cy, k, s=(1, 1), p=(0, 0), act=HF.ACT_NONE, out=None):
        N, Cx, H, W = x.N, x.C, x.H, x.W
        OH, OW = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
        y = out if out is not None else self.new(Cy, OH, OW)
        if (y.C, y.H, y.W) != (Cy, OH, OW):
            raise RuntimeError("conv output buffer has the wrong shape")
        self.f.call("mgvae_conv2d_fwd", self.f.struct(self._desc(N, Cx, H, W, Cy, OH, OW, k, s, p, x.ctot, y.ctot, act)), self.fp(x),
                    self.fs["p:" + wname], None, self.fp(y), self.fs["st"])

        def backward():
            g = y.grad
            if g is None:
                return
            b = self.b
            if act != HF.ACT_NONE:
                d = self.new(Cy, OH, OW, "ba")
                b.call("mgvae_act_bwd", self.bp(y), self.bp(g), self.bp(d), N, Cy, OH * OW, y.ctot, 0, g.ctot, 0, Cy, 0, act, 0.01, self.bs["st"])
                g = d
            if self.trainable[wname]:
                b.call("mgvae_stream_fork", self.bs["st"], self.bs["side"])
                b.call("mgvae_conv2d_bwd_weight", b.struct(self._desc(N, Cx, H, W, Cy, OH, OW, k, s, p, x.ctot, g.ctot)), self.bp(x), self.bp(g),
                       self.bs["g:" + wname], self.bs["side"])
            if x.needs_grad:
                dx = self.new(Cx, H, W, "ba")
                d2 = self._desc(N, Cx, H, W, Cy, OH, OW, k, s, p, Cx, g.ctot)
                if k[0] * k[1] > 1 and HF.USE_TRANSPOSED_W:
                    wt = self._alloc("ba", 4 * Cy * Cx * k[0] * k[1])
                    b.call("mgvae_weight_transpose", self.bs["p:" + wname], self.bs["ba"] + wt, Cy, Cx, k[0] * k[1], self.bs["st"])
                    b.call("mgvae_conv2d_bwd_data_tw", b.struct(d2), self.bp(g), self.bs["ba"] + wt, None, self.bp(dx), self.bs["st"])
                else:
                    b.call("mgvae_conv2d_bwd_data", b.struct(d2), self.bp(g), self.bs["p:" + wname], None, self.bp(dx), self.bs["st"])
                self._acc(x, dx)
        self.bwd.append(backward)
        return y

    def batch_norm(self, x, prefix, training, momentum, eps, act=HF.ACT_NONE):
        if x.ctot != x.C:
            raise RuntimeError("batch_norm needs a dense input")
        N, C, H, W = x.N, x.C, x.H, x.W
        y = self.new(C, H, W)
        stats = self._alloc("fa", 4 * 2 * C)
        nm = lambda s_: "p:" + prefix + s_
        self.f.call("mgvae_batch_norm_fwd", self.fp(x), self.fs[nm("weight")], self.fs[nm("bias")], self.fs[nm("running_mean")],
                    self.fs[nm("running_var")], self.fp(y), self.fs["fa"] + stats, N, C, H * W, 1 if training else 0, momentum, eps, act, 0.01,
                    self.fs["st"])

        def backward():
            g = y.grad
            if g is None:
                return
            if g.ctot != g.C:
                raise RuntimeError("batch_norm backward needs a dense gradient")
            dx = self.new(C, H, W, "ba")
            gw = self.bs["g:" + prefix + "weight"] if self.trainable[prefix + "weight"] else None
            gb = self.bs["g:" + prefix + "bias"] if self.trainable[prefix + "bias"] else None
            self.b.call("mgvae_batch_norm_bwd", self.bp(x), self.bs[nm("weight")], self.bs[nm("bias")], self.bs["fa"] + stats, self.bp(g),
                        self.bp(dx), gw, gb, N, C, H * W, 1 if training else 0, act, 0.01, self.bs["st"])
            self._acc(x, dx)
        self.bwd.append(backward)
        return y

    def group_sum(self, x, gsize):
        """sum over groups of ``gsize`` adjacent pitches (last axis)"""
        if x.ctot != x.C or x.W % gsize:
            raise RuntimeError("group_sum needs a dense input whose width is a multiple of the group size")
        groups = x.W // gsize
        y = self.new(x.C, x.H, groups, needs_grad=x.needs_grad)
        rows = x.N * x.C * x.H
        self.f.call("mgvae_group_sum_fwd", self.fp(x), self.fp(y), rows, groups, gsize, self.fs["st"])

        def backward():
            if y.grad is None or not x.needs_grad:
                return
            dx = self.new(x.C, x.H, x.W, "ba")
            self.b.call("mgvae_group_sum_bwd", self.bp(y.grad), self.bp(dx), rows, groups, gsize, self.bs["st"])
            self._acc(x, dx)
        self.bwd.append(backward)
        return y

    def pool_into(self, x, feat, col):
        """whole-map average [N, C, H, W] -> columns [col, col + C) of the 2-D buffer ``feat`` [N, feat.C] (feat is an
        NCHW T with H = W = 1)"""
        if x.ctot != x.C:
            raise RuntimeError("pool needs a dense input")
        N, C, L = x.N, x.C, x.H * x.W
        m = self.new(C, 1, 1)
        self.f.call("mgvae_rowmean_fwd", self.fp(x), self.fp(m), N * C, L, self.fs["st"])
        self.f.call("mgvae_copy2d", self.fp(feat) + 4 * col, feat.C, self.fp(m), C, C, N, self.fs["st"])

        def backward():
            if feat.grad is None:
                return
            dm = self.new(C, 1, 1, "ba")
            self.b.call("mgvae_copy2d", self.bp(dm), C, self.bp(feat.grad) + 4 * col, feat.C, C, N, self.bs["st"])
            dx = self.new(C, x.H, x.W, "ba")
            self.b.call("mgvae_rowmean_bwd", self.bp(dm), self.bp(dx), N * C, L, self.bs["st"])
            self._acc(x, dx)
        self.bwd.append(backward)

    def linear_out(self, feat, wname, act):
        """the final Linear (no bias) + activation into the node's output ``y`` [N, 1]"""
        N, K = feat.N, feat.C
        y = T("y", 0, N, 1, 1, 1)
        self.f.call("mgvae_conv2d_fwd", self.f.struct(self._desc(N, K, 1, 1, 1, 1, 1, (1, 1), (1, 1), (0, 0), K, 1, act)), self.fp(feat),
                    self.fs["p:" + wname], None, self.fs["y"], self.fs["st"])

        def backward():
            b = self.b
            d = self.new(1, 1, 1, "ba")
            b.call("mgvae_act_bwd", self.bs["y"], self.bs["dy"], self.bp(d), N, 1, 1, 1, 0, 1, 0, 1, 0, act, 0.01, self.bs["st"])
            if self.trainable[wname]:
                b.call("mgvae_stream_fork", self.bs["st"], self.bs["side"])
                b.call("mgvae_conv2d_bwd_weight", b.struct(self._desc(N, K, 1, 1, 1, 1, 1, (1, 1), (1, 1), (0, 0), K, 1)), self.bp(feat), self.bp(d),
                       self.bs["g:" + wname], self.bs["side"])
            df = self.new(K, 1, 1, "ba")
            b.call("mgvae_conv2d_bwd_data", b.struct(self._desc(N, K, 1, 1, 1, 1, 1, (1, 1), (1, 1), (0, 0), K, 1)), self.bp(d),
                   self.bs["p:" + wname], None, self.bp(df), self.bs["st"])
            feat.grad = df
        self.bwd.append(backward)
        return y

    # ------------------------------------------------------------------ finishing
    def finish(self, x):
        for fn in reversed(self.bwd):
            fn()
        if self.need_dx:
            if x.grad is None:
                raise RuntimeError("the input's gradient was asked for but nothing produced it")
            self.b.call("mgvae_copy2d", self.bs["dx"], x.numel // x.N, self.bp(x.grad), x.numel // x.N, x.numel // x.N, x.N, self.bs["st"])
        self.f.finalize()
        self.b.finalize()
        return self


# ============================================================================================== the bar discriminator
def _build_bar_disc(N, names, trainable, need_dx, training, bn_cfg):
    """graph/bar_discriminator.py:200-217 with its three towers (:31-58, :85-100, :165-183), op for op as
    musicgeneration_vae-torch_amd/graph/bar_discriminator.py issues them"""
    Rl = HF.ACT_RELU
    nb = NetBuilder(N, names, trainable, need_dx)
    x = T("x", 0, N, 1, 192, 60, needs_grad=need_dx)
    feat = nb.new(192, 1, 1)
    bn = lambda t, prefix, act=Rl: nb.batch_norm(t, prefix, training, bn_cfg[prefix][0], bn_cfg[prefix][1], act)
    # chord tower
    o = nb.group_sum(x, 5)
    o = bn(nb.conv(o, "chord.chord_conv1.weight", 8, (3, 1), (2, 1), (1, 0)), "chord.batch_norm1.")
    o = bn(nb.conv(o, "chord.chord_conv2.weight", 16, (3, 1), (2, 1), (1, 0)), "chord.batch_norm2.")
    o = bn(nb.conv(o, "chord.chord_fit.weight", 16, (1, 1)), "chord.batch_norm3.")
    o = bn(nb.conv(o, "chord.chord_conv3.weight", 32, (3, 3), (2, 2), (1, 1)), "chord.batch_norm4.")
    o = bn(nb.conv(o, "chord.chord_conv4.weight", 64, (3, 3), (2, 2), (1, 1)), "chord.batch_norm5.")
    nb.pool_into(o, feat, 0)
    # on/off tower
    o = nb.group_sum(x, 60)
    o = nb.conv(o, "onoff.onoff_conv1.weight", 8, (3, 3), (2, 1), (1, 1), Rl)
    o = nb.conv(o, "onoff.onoff_conv2.weight", 8, (3, 3), (2, 1), (1, 1), Rl)
    o = bn(o, "onoff.batch_norm2.", HF.ACT_NONE)
    o = nb.conv(o, "onoff.onoff_conv3.weight", 16, (3, 3), (2, 1), (1, 1), Rl)
    o = nb.conv(o, "onoff.onoff_conv4.weight", 32, (3, 3), (2, 1), (1, 1), Rl)
    o = nb.conv(o, "onoff.onoff_fit.weight", 32, (1, 1), act=Rl)
    o = nb.conv(o, "onoff.onoff_conv5.weight", 64, (3, 3), (2, 1), (1, 1), Rl)
    nb.pool_into(o, feat, 64)
    # basic tower
    cat = nb.new(16, 96, 30)
    half = 4 * 8 * 96 * 30
    lo = T("fa", cat.off, N, 8, 96, 30, ctot=16)
    hi = T("fa", cat.off + half, N, 8, 96, 30, ctot=16)
    p1 = nb.conv(x, "basic.pitch1.weight", 8, (1, 4), (1, 2), (0, 1), Rl)
    nb.conv(p1, "basic.pitch2.weight", 8, (4, 1), (2, 1), (1, 0), Rl, out=lo)
    t1 = nb.conv(x, "basic.time1.weight", 8, (4, 1), (2, 1), (1, 0), Rl)
    nb.conv(t1, "basic.time2.weight", 8, (1, 4), (1, 2), (0, 1), Rl, out=hi)

    def split_cat_grad():
        if cat.grad is not None:
            g = cat.grad
            lo.grad = T(g.where, g.off, N, 8, 96, 30, ctot=16)
            hi.grad = T(g.where, g.off + half, N, 8, 96, 30, ctot=16)
    o = nb.conv(cat, "basic.fit.weight", 8, (1, 1))
    # (closures run in reverse: this one runs AFTER fit's backward produced cat.grad and BEFORE pitch2 / time2's)
    nb.bwd.insert(len(nb.bwd) - 1, split_cat_grad)
    o = bn(o, "basic.bn.")
    for i, (cin, cout, basic) in enumerate(((8, 16, False), (16, 32, False), (32, 64, True))):
        q = "basic.layers.%d." % i
        if not basic:
            o = bn(nb.conv(o, q + "conv1.weight", cin, (3, 3), (1, 1), (1, 1)), q + "bn1.")
        o = bn(nb.conv(o, q + "conv2.weight", cout, (3, 3), (2, 2), (1, 1)), q + "bn2.")
    nb.pool_into(o, feat, 128)
    nb.linear_out(feat, "linear.weight", HF.ACT_SIGMOID)
    return nb.finish(x)


_nets = {}


class _BarDiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, training, names, bn_cfg, *tensors):
        HF._need_cuda(x, "bar discriminator")
        x = x.reshape(-1, 1, 192, 60).contiguous()
        N = x.shape[0]
        trainable = tuple(bool(t.requires_grad) for t in tensors)
        need_dx = bool(x.requires_grad)
        key = (N, training, trainable, need_dx, bool(HF.USE_TRANSPOSED_W))
        nb = _nets.get(key)
        if nb is None:
            nb = _nets[key] = _build_bar_disc(N, names, trainable, need_dx, training, dict(bn_cfg))
        fa = torch.empty(max(nb.fa_bytes, 16), device=x.device, dtype=torch.uint8)
        y = torch.empty((N, 1), device=x.device, dtype=torch.float32)
        main = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
        nb.f.run([x.data_ptr(), fa.data_ptr(), y.data_ptr(), main] + [t.data_ptr() for t in tensors])
        ctx.save_for_backward(x, fa, y, *tensors)
        ctx.nb = nb
        return y

    @staticmethod
    def backward(ctx, dy):
        from .blocks import _grad, _ptr, _side_for
        sv = ctx.saved_tensors
        x, fa, y, tensors = sv[0], sv[1], sv[2], sv[3:]
        nb = ctx.nb
        dy = dy.contiguous()
        ba = torch.empty(max(nb.ba_bytes, 16), device=x.device, dtype=torch.uint8)
        dx = torch.empty_like(x) if nb.need_dx else None
        main, side = _side_for(x.shape[0], any(nb.trainable.values()), True, (x, fa, ba))
        addr = [x.data_ptr(), fa.data_ptr(), ba.data_ptr(), y.data_ptr(), dy.data_ptr(), _ptr(dx), main, side]
        for t in tensors:
            addr += [t.data_ptr(), _ptr(_grad(t)) if t.requires_grad else 0]
        nb.b.run(addr)
        return (dx, None, None, None) + (None,) * len(tensors)


def _collect(module):
    """(names, tensors, BatchNorm configuration, counters of the BatchNorms the forward uses), cached on the module (the
    module drops the cache whenever its tensors are re-created: graph.bar_discriminator.BarDiscriminator._apply)"""
    c = module.__dict__.get("_mg_chain_cache")
    if c is None:
        names, tensors = [], []
        for n, p in module.named_parameters():
            names.append(n); tensors.append(p)
        bn_cfg, counters = [], []
        for mn, m in module.named_modules():
            if type(m).__name__ == "BatchNorm2d":
                for bname in ("running_mean", "running_var"):
                    names.append(mn + "." + bname); tensors.append(getattr(m, bname))
                bn_cfg.append((mn + ".", (float(m.momentum), float(m.eps))))
                if not mn.endswith("layers.2.bn1"):          # constructed by the reference, never used (ConvModule isBasic)
                    counters.append(m.num_batches_tracked)
        c = module.__dict__["_mg_chain_cache"] = (tuple(names), tensors, tuple(bn_cfg), counters)
    return c


def bar_discriminator(module, x):
    """``module``: graph.bar_discriminator.BarDiscriminator"""
    names, tensors, bn_cfg, counters = _collect(module)
    training = bool(module.training)
    y = _BarDiscFn.apply(x, training, names, bn_cfg, *tensors)
    if training:
        torch._foreach_add_(counters, 1)          # torch's BatchNorm bookkeeping, one launch for all ten counters
    return y


def usable(module, x):
    from . import blocks as HB
    if not (HB.ENABLED and x.is_cuda and x.dtype == torch.float32 and HF.get_nchw_operand_dtype() == "f32" and not HF.USE_DIRECT):
        return False
    # (one mode for the whole network: the agents only ever switch it as a whole; checked on its three towers)
    return module.chord.training == module.training and module.onoff.training == module.training and \
        module.basic.training == module.training and module.basic.bn.training == module.training
