"""torch.autograd bindings of the HIP kernels (C ABI: include/mgvae.h).

PyTorch only supplies device memory, the current HIP stream and the autograd tape;
every forward and backward below is one or more launches from libmgvae_hip.so.

Conventions
  * fp32, NCHW.  A tensor may be a *channel slice* ``buf[:, a:b]`` of a larger buffer
    (that is how the reference's ``torch.cat(dim=1)`` is realised without a copy): ops
    read the channel pitch from the strides.
  * Weight-like gradients (conv / linear weights and biases, InstanceNorm affine, CBAM
    weights, the embedding table) are ACCUMULATED straight into ``param.grad`` by the
    backward kernels and ``None`` is returned to autograd for them -- the flat gradient
    buffer (hipops.flat.FlatParams) is therefore written exactly once, by HIP code.
"""
import ctypes

import numpy as np
import torch

from . import _native as nat

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
_vp = ctypes.c_void_p


def _p(t):
    return _vp(t.data_ptr()) if t is not None else None


def _s():
    """the current HIP stream as a void* (raw C entry points: torch.cuda.current_stream() costs ~8 us of Python per call,
    161 calls in a forward pass)"""
    return _vp(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


_stream_objs = {}


def cur_stream():
    """``torch.cuda.current_stream()`` without its ~9 us of Python device look-ups (123 calls and 1.1 ms per step at 32 bars,
    tools/host_profile.py): the Stream object of the current raw stream, built once per (device, stream)."""
    dev = torch._C._cuda_getDevice()
    raw = torch._C._cuda_getCurrentRawStream(dev)
    st = _stream_objs.get((dev, raw))
    if st is None:
        st = _stream_objs[(dev, raw)] = torch.cuda.current_stream()
    return st


def cur_raw_stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: expected a ROCm device tensor; the MI355X hot path has no CPU fallback" % what)
    if t.dtype != torch.float32:
        raise RuntimeError("%s: expected float32, got %s" % (what, t.dtype))


def _pitch(t):
    """channel pitch (ctot) of an NCHW tensor that is dense or a channel slice; else None."""
    n, c, h, w = t.shape
    hw = h * w
    st = t.stride()
    if w > 1 and st[3] != 1:
        return None
    if h > 1 and st[2] != w:
        return None
    if c > 1 and st[1] != hw:
        return None
    if n > 1:
        if st[0] % hw or st[0] // hw < c:
            return None
        return st[0] // hw
    return c


def _sliceable(t, what="tensor"):
    """return (tensor, ctot); makes a dense copy only if the layout is not a channel slice"""
    ct = _pitch(t)
    if ct is None:
        t = t.contiguous()
        ct = t.shape[1]
    return t, ct


def grad_slot(p):
    """the tensor the backward kernels accumulate into for parameter ``p``"""
    g = p.grad
    if g is None:
        g = getattr(p, "_mg_grad", None)
        if g is None:
            g = torch.zeros_like(p)
        else:
            g.zero_()
        p.grad = g
    return g


def _desc(N, Cx, H, W, Cy, OH, OW, k, s, p, x_ctot, y_ctot, act, slope):
    return nat.ConvDesc(N, Cx, H, W, Cy, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], x_ctot, 0, y_ctot, 0, act, slope)


# One process drives one GPU: torch's autograd engine would still hand every backward node of a CUDA tensor to a worker thread
# of its own, and that hand-over (a condition variable per node, the GIL passed back and forth between the caller and the
# worker that runs our Python backward functions) is what made the small-batch steps bimodal -- 7.8 or 8.9-9.5 ms at 32 bars
# bf16 from one run to the next, 15.5-16.0 against 15.1 ms for the GAN iteration at 16 bars (tools/r3_run28.sh).  Backward runs
# on the calling thread instead (MGVAE_AUTOGRAD_THREAD=1 restores torch's default).
if __import__("os").environ.get("MGVAE_AUTOGRAD_THREAD", "0") == "0":
    torch.autograd.set_multithreading_enabled(False)

USE_TRANSPOSED_W = __import__("os").environ.get("MGVAE_TW", "1") != "0"   # data-gradient / transposed-conv kernels read a [Cy][KK][Cx] copy of the weight
import os as _os
# direct (halo-tile, packed-weight, register-prefetched) conv kernels vs the implicit GEMM: MGVAE_DIRECT=0 (default)
# implicit GEMM only, 1 = direct wherever supported, auto = time both the first time a geometry is seen and keep the
# faster.  Back to back the direct kernel wins the large-map 3x3 layers (100-107 vs 88-98 TFLOP/s) and loses the rest;
# in the step "auto" moves 11 launches and the step time does not change (27.5 ms either way), so it stays opt-in.
USE_DIRECT = {"0": False, "1": True, "auto": "auto"}.get(_os.environ.get("MGVAE_DIRECT", "0"), False)
_impl_choice = {}


def _direct_candidate(d):
    """"auto" only considers the direct kernel where it can win: 3x3 / 4x4-class windows on maps of >= 128 pixels,
    fp32 operands (True tries it wherever the geometry is supported)"""
    if USE_DIRECT is True:
        return True
    return d.KH * d.KW >= 9 and d.OH * d.OW >= 128 and d.H * d.W >= 128 and get_nchw_operand_dtype() == "f32"


def _direct_wins(mode, d, run_igemm, run_direct):
    """time both implementations once per (mode, geometry) and remember the faster"""
    key = (mode, d.N, d.Cx, d.H, d.W, d.Cy, d.KH, d.KW, d.SH, d.SW, d.PH, d.PW, d.x_ctot, d.y_ctot)
    c = _impl_choice.get(key)
    if c is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        times = []
        for fn in (run_igemm, run_direct):
            fn()                                   # warm-up (also lets the implicit GEMM autotune itself)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(); fn()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        c = times[1] < 0.97 * times[0]
        _impl_choice[key] = c
    return c


DEFER_ACT_GRAD = _os.environ.get("MGVAE_DEFER_ACT", "1") != "0"   # fold act' into the consumer's data gradient


def _mask(t, act, slope):
    """MgvaeActMask over tensor ``t`` (the activated tensor whose act'(.) multiplies the conv's output)"""
    t, ct = _sliceable(t)
    return nat.ActMask(t.data_ptr(), ct, 0, act, slope), t


_compute_dtype = ["f32"]


def set_compute_dtype(name):
    """"f32" (default) or "bf16" (BASELINE.json configs 3-4).  bf16 = the channels-last island -- the encoder trunks and
    the decoder's blocks, 97 % of the step's FLOPs -- keeps activations and their gradients as bf16 tensors and multiplies
    bf16 copies of the fp32 master weights on the bf16 matrix pipe (fp32 accumulation, statistics, gates, weight gradients
    and Adam); everything outside the island (C = 1 / C = 32 stems, the latent linears, the output convs, the
    discriminators) stays fp32.  One rounding model, independent of which kernel variant the tuner picks: it is the one
    oracle.restate.ISLAND_ROUNDING restates and tests/test_hip_parity.py checks the bf16 step against."""
    name = {"fp32": "f32", "float32": "f32", "bfloat16": "bf16"}.get(str(name).lower(), str(name).lower())
    if name not in ("f32", "bf16"):
        raise ValueError("compute dtype must be f32 or bf16, got %r" % (name,))
    _compute_dtype[0] = name


def get_compute_dtype():
    return _compute_dtype[0]


def set_nchw_operand_dtype(name):
    """operand precision of the NCHW tiled conv kernels alone (mgvae_set_compute_dtype): "f32", "bf16" (operands rounded to
    bf16 while staged, fp32 accumulation) or "f32_bf16x3".  Not used by any model path; the kernels are kept and tested
    (tests/test_hip_parity.py::test_conv2d_bf16_compute)."""
    code = {"f32": 0, "bf16": 1, "f32_bf16x3": 2}[str(name).lower()]
    nat.check(nat.lib().mgvae_set_compute_dtype(code), "set_compute_dtype")


def get_nchw_operand_dtype():
    return {0: "f32", 1: "bf16", 2: "f32_bf16x3"}[nat.lib().mgvae_get_compute_dtype()]


def _conv_fwd(d, x, w, b, y, mask=None):
    """Y = conv(X, Wt): direct kernel when the geometry is supported, else implicit GEMM; ``mask`` = (tensor, act,
    slope): multiply Y by act'(tensor) in the epilogue"""
    L = nat.lib()
    if mask is not None:
        m, keep = _mask(*mask)
        nat.check(L.mgvae_conv2d_fwd_masked(ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), ctypes.byref(m), _s()), "conv2d_fwd_masked")
        return
    def igemm():
        nat.check(L.mgvae_conv2d_fwd(ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), _s()), "conv2d_fwd")

    n = L.mgvae_conv_pack_floats(ctypes.byref(d), 0) if (USE_DIRECT and _direct_candidate(d)) else 0
    if not n:
        return igemm()

    def direct():
        wp = torch.empty((n,), device=w.device, dtype=torch.float32)
        nat.check(L.mgvae_conv_pack(ctypes.byref(d), 0, _p(w), _p(wp), _s()), "conv_pack")
        nat.check(L.mgvae_conv2d_fwd_packed(ctypes.byref(d), _p(x), _p(wp), _p(b), _p(y), _s()), "conv2d_fwd_packed")

    if USE_DIRECT is True or _direct_wins(0, d, igemm, direct):
        direct()
    else:
        igemm()


def _conv_bwd_data(d, y, w, b, x, mask=None):
    """X = conv_transpose(Y, Wt) (+bias): direct per-phase kernels when supported; ``mask`` as in _conv_fwd (over X)"""
    L = nat.lib()
    if mask is not None:
        m, keep = _mask(*mask)
        tw = d.KH * d.KW > 1 and USE_TRANSPOSED_W
        if tw:
            wt = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
            nat.check(L.mgvae_weight_transpose(_p(w), _p(wt), d.Cy, d.Cx, d.KH * d.KW, _s()), "weight_transpose")
            w = wt
        nat.check(L.mgvae_conv2d_bwd_data_masked(ctypes.byref(d), _p(y), _p(w), 1 if tw else 0, _p(b), _p(x), ctypes.byref(m), _s()),
                  "conv2d_bwd_data_masked")
        return
    def igemm():
        if d.KH * d.KW > 1 and USE_TRANSPOSED_W:
            wt = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
            nat.check(L.mgvae_weight_transpose(_p(w), _p(wt), d.Cy, d.Cx, d.KH * d.KW, _s()), "weight_transpose")
            nat.check(L.mgvae_conv2d_bwd_data_tw(ctypes.byref(d), _p(y), _p(wt), _p(b), _p(x), _s()), "conv2d_bwd_data_tw")
        else:
            nat.check(L.mgvae_conv2d_bwd_data(ctypes.byref(d), _p(y), _p(w), _p(b), _p(x), _s()), "conv2d_bwd_data")

    n = L.mgvae_conv_pack_floats(ctypes.byref(d), 1) if (USE_DIRECT and _direct_candidate(d)) else 0
    if not n:
        return igemm()

    def direct():
        wp = torch.empty((n,), device=w.device, dtype=torch.float32)
        nat.check(L.mgvae_conv_pack(ctypes.byref(d), 1, _p(w), _p(wp), _s()), "conv_pack")
        nat.check(L.mgvae_conv2d_bwd_data_packed(ctypes.byref(d), _p(y), _p(wp), _p(b), _p(x), _s()), "conv2d_bwd_data_packed")

    if USE_DIRECT is True or _direct_wins(1, d, igemm, direct):
        direct()
    else:
        igemm()


# ---- side streams: independent work of one backward node (the weight gradient vs the data gradient of a conv) and
# independent branches of the model run concurrently; under-filled launches (small maps, tiny GEMMs) then overlap.
SERIAL = _os.environ.get("MGVAE_SERIAL", "0") != "0"       # one stream only: per-kernel profiles without co-running kernels
FORK_WGRAD = _os.environ.get("MGVAE_FORK_WGRAD", "1") != "0" and not SERIAL
_side_streams = {}      # (device index, id of the stream forked from) -> side stream
_used_sides = {}        # streams with work of the running backward pass, to be joined by _join_sides
# forking costs the host ~30 us per conv backward (events, allocator bookkeeping): it pays when the step is bound by the
# GPU (batch 64: 30.5 -> 27.4 ms) and costs when the host's launch rate is the bound (the ~2000-launch GAN iteration at
# 16 bars per GPU: 26.0 -> 29.1 ms), so small batches stay on one stream
FORK_MIN_BATCH = int(_os.environ.get("MGVAE_FORK_MIN_BATCH", "32"))
WGRAD_STREAMS = int(_os.environ.get("MGVAE_WGRAD_STREAMS", "1"))   # side streams the weight gradients rotate over
_wgrad_rr = [0]
_join_queued = [-1]      # id of the autograd graph task that already has the join callback queued


def _shared_sides():
    """under torch.distributed ONE weight-gradient stream serves every forking stream: main + phrase trunk + that
    stream are then the only compute streams beside RCCL's own (a fifth busy stream shares a hardware
    queue with another one and the step falls off a cliff, 23.9 -> 36 ms).  MGVAE_SHARED_WGRAD=0/1 overrides."""
    v = _os.environ.get("MGVAE_SHARED_WGRAD")
    if v is not None:
        return v != "0"
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def side_stream_of(cur, slot=0):
    key = (cur.device.index, 0 if (slot >= 2 and _shared_sides()) else cur.cuda_stream, slot)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=cur.device)
        _side_streams[key] = st
    return st


_trunk_streams = []      # long-lived side streams that carry whole branches of the model (Model.encode_phrase)


def register_trunk_stream(st):
    """a stream on which a module runs a whole branch: autograd replays that branch's backward there, and its
    weight-like gradients are written by our kernels (no AccumulateGrad node, so the engine does not sync it back):
    the end-of-backward join and the gradient reducer wait for it explicitly"""
    if all(st is not t for t in _trunk_streams):
        _trunk_streams.append(st)


def _join_sides(slot=None, parent=None):
    """make the caller's stream wait for every side stream used since the last join (end of a backward pass);
    ``slot``: only the side streams of that slot; ``parent``: only the side streams forked from that stream (and no
    trunk stream): what an early gradient bucket launched from the step's main stream has to wait for"""
    cur = cur_stream()
    if slot is None and parent is None:
        for st in _trunk_streams:
            if st.device == cur.device and st.cuda_stream != cur.cuda_stream:
                cur.wait_stream(st)
    for key in list(_used_sides):
        st, sl, par = _used_sides[key]
        if (slot is None or sl == slot) and (parent is None or par is None or par == parent.cuda_stream):
            cur.wait_stream(st)
            del _used_sides[key]
    if slot is None and parent is None:
        _join_queued[0] = -1


def join_side_streams(slot=None, parent=None):
    _join_sides(slot, parent)


def live_streams():
    """every stream this module has created (side streams of forked work, registered trunk streams): the data-parallel
    step asserts their number at start-up (more concurrent streams than hardware queues is a cliff)"""
    seen = {}
    for st in list(_side_streams.values()) + list(_trunk_streams):
        seen[st.cuda_stream] = st
    return list(seen.values())


def _ensure_join_callback():
    """queue the end-of-backward join once per backward pass (called from every conv backward)"""
    task = torch._C._current_graph_task_id()
    if task != -1 and _join_queued[0] != task:
        torch.autograd.Variable._execution_engine.queue_callback(_join_sides)
        _join_queued[0] = task


class _forked:
    """``with _forked(tensors...)``: run the body on the side stream paired with the current stream, after everything
    already enqueued on the current stream; the listed tensors are marked as in use there (caching allocator)."""

    def __init__(self, *tensors, slot=0):
        self.tensors, self.slot = tensors, slot

    def __enter__(self):
        cur = cur_stream()
        side = side_stream_of(cur, self.slot)
        side.wait_stream(cur)
        for t in self.tensors:
            if t is not None:
                t.record_stream(side)
        _used_sides[id(side)] = (side, self.slot, None if (self.slot >= 2 and _shared_sides()) else cur.cuda_stream)
        _ensure_join_callback()      # once per backward pass (keyed by task: survives an aborted pass)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return side

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)


FORK_BRANCHES = _os.environ.get("MGVAE_FORK_BRANCHES", "0") != "0" and not SERIAL   # measured: -0.5 % (decoder branches, z-losses) -> off


class forked_branch:
    """``with forked_branch(x, buf):`` in a module's forward: run one of two independent branches on the side stream
    (inputs / output buffers listed so the allocator knows); call ``join_side_streams()`` before the results meet.
    A no-op context when disabled or on CPU tensors."""

    def __init__(self, *tensors, slot=0):
        self.on = FORK_BRANCHES and all(t is None or t.is_cuda for t in tensors)
        self.f = _forked(*tensors, slot=slot) if self.on else None

    def __enter__(self):
        return self.f.__enter__() if self.on else None

    def __exit__(self, *a):
        return self.f.__exit__(*a) if self.on else False


def _act_bwd(y, dy, act, slope):
    """dx = dy * act'(y) -> dense tensor"""
    y, yct = _sliceable(y)
    dy, dct = _sliceable(dy)
    n, c, h, w = y.shape
    dx = torch.empty((n, c, h, w), device=y.device, dtype=torch.float32)
    nat.check(nat.lib().mgvae_act_bwd(_p(y), _p(dy), _p(dx), n, c, h * w, yct, 0, dct, 0, c, 0, act, slope, _s()), "act_bwd")
    return dx


# =============================================================================== conv
class _ConvFn(torch.autograd.Function):
    """nn.Conv2d forward/backward (reference: graph/encodingBlock.py:12-15,74-77,107-108;
    graph/decoder.py:79,122,172,175) and, with H=W=1, nn.Linear."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, slope, out, in_act=None, defer_act_grad=False):
        _need_cuda(x, "conv2d")
        x, xct = _sliceable(x)
        N, Cx, H, W = x.shape
        Cy = w.shape[0]
        KH, KW = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
        OH = (H + 2 * pad[0] - KH) // stride[0] + 1
        OW = (W + 2 * pad[1] - KW) // stride[1] + 1
        y = out if out is not None else torch.empty((N, Cy, OH, OW), device=x.device, dtype=torch.float32)
        yct = _pitch(y)
        d = _desc(N, Cx, H, W, Cy, OH, OW, (KH, KW), stride, pad, xct, yct, act, slope)
        _conv_fwd(d, x, w, b, y)
        ctx.geom = (N, Cx, H, W, Cy, OH, OW, (KH, KW), stride, pad, xct, act, slope)
        # in_act = (act, slope): x is the ACTIVATED output of the producing layer, which skipped its own dy*act'(y)
        # pass (defer_act_grad there); this conv's data gradient applies act'(x) while storing.  The two flags must
        # be used as a pair, and only when this conv is the sole consumer of x.
        if not DEFER_ACT_GRAD:
            in_act, defer_act_grad = None, False
        ctx.in_act, ctx.defer = in_act, bool(defer_act_grad)
        ctx.save_for_backward(x, w, y if (act != ACT_NONE and not defer_act_grad) else None)
        ctx.b = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N, Cx, H, W, Cy, OH, OW, k, s, p, xct, act, slope = ctx.geom
        L = nat.lib()
        if _trunk_streams or _used_sides:
            _ensure_join_callback()
        if act != ACT_NONE and not ctx.defer:
            dy = _act_bwd(y, dy, act, slope)
        dy, dct = _sliceable(dy)
        d = _desc(N, Cx, H, W, Cy, OH, OW, k, s, p, xct, dct, ACT_NONE, 0.0)
        b = ctx.b

        def weight_grads():
            if w.requires_grad:
                nat.check(L.mgvae_conv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dy), _p(grad_slot(w)), _s()), "conv2d_bwd_weight")
            if b is not None and b.requires_grad:
                nat.check(L.mgvae_channel_sum_accum(_p(dy), N, Cy, OH * OW, dct, 0, _p(grad_slot(b)), _s()), "bias_grad")

        if FORK_WGRAD and N >= FORK_MIN_BATCH and ctx.needs_input_grad[0] and (w.requires_grad or (b is not None and b.requires_grad)):
            _wgrad_rr[0] += 1
            with _forked(x, dy, slot=2 + _wgrad_rr[0] % WGRAD_STREAMS):   # the weight gradient runs beside the data gradient below
                weight_grads()
        else:
            weight_grads()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, Cx, H, W), device=dy.device, dtype=torch.float32)
            d2 = _desc(N, Cx, H, W, Cy, OH, OW, k, s, p, Cx, dct, ACT_NONE, 0.0)
            _conv_bwd_data(d2, dy, w, None, dx, mask=(x,) + tuple(ctx.in_act) if ctx.in_act else None)
        return dx, None, None, None, None, None, None, None, None, None


def conv2d(x, w, b=None, stride=(1, 1), pad=(0, 0), act=ACT_NONE, slope=0.01, out=None, in_act=None, defer_act_grad=False):
    return _ConvFn.apply(x, w, b, stride, pad, act, slope, out, in_act, defer_act_grad)


def linear(x, w, b=None, act=ACT_NONE, slope=0.01, out=None):
    """x [B, K] (dense or a column slice of a wider [B, ctot] buffer) -> [B, Cout]"""
    B = x.shape[0]
    x4 = x.unsqueeze(-1).unsqueeze(-1)
    o4 = out.unsqueeze(-1).unsqueeze(-1) if out is not None else None
    y = _ConvFn.apply(x4, w, b, (1, 1), (0, 0), act, slope, o4)
    return y.view(B, -1) if out is None else y.squeeze(-1).squeeze(-1)


class _ConvTFn(torch.autograd.Function):
    """nn.ConvTranspose2d (graph/decoder.py:12-15,43-46,73-77,116-120): forward is the
    stride-phase data-gradient kernel, d/dx is the forward-conv kernel."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, opad, act, slope, out, in_act=None, defer_act_grad=False):
        _need_cuda(x, "conv_transpose2d")
        x, xct = _sliceable(x)
        N, Ci, h, wd = x.shape
        _, Co, KH, KW = w.shape
        OH = (h - 1) * stride[0] - 2 * pad[0] + KH + opad[0]
        OW = (wd - 1) * stride[1] - 2 * pad[1] + KW + opad[1]
        y = out if out is not None else torch.empty((N, Co, OH, OW), device=x.device, dtype=torch.float32)
        yct = _pitch(y)
        k = (KH, KW)
        if (h == 1 and wd == 1 and tuple(stride) == k and tuple(pad) == (0, 0) and tuple(opad) == (0, 0) and b is None
                and yct == Co):
            # a 1x1 map through a non-overlapping transposed conv (the decoder stems, graph/decoder.py:43-46) is the
            # plain GEMM y[n, (co,kh,kw)] = x[n, ci] . w[ci, (co,kh,kw)]: one launch with the weight read in place,
            # instead of KH*KW one-tap phases over a transposed copy of the 14 M-element weight
            Co, OH, OW, k, stride, yct = Co * KH * KW, 1, 1, (1, 1), (1, 1), Co * KH * KW
        # conv geometry: X = y (image side, Cx = Co), Y = x (feature side, Cy = Ci)
        d = _desc(N, Co, OH, OW, Ci, h, wd, k, stride, pad, yct, xct, act, slope)
        _conv_bwd_data(d, x, w, b, y)
        ctx.geom = (N, Co, OH, OW, Ci, h, wd, k, stride, pad, xct, act, slope)
        if not DEFER_ACT_GRAD:
            in_act, defer_act_grad = None, False
        ctx.in_act, ctx.defer = in_act, bool(defer_act_grad)      # see _ConvFn.forward
        ctx.save_for_backward(x, w, y if (act != ACT_NONE and not defer_act_grad) else None)
        ctx.b = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N, Co, OH, OW, Ci, h, wd, k, s, p, xct, act, slope = ctx.geom
        L = nat.lib()
        if act != ACT_NONE and not ctx.defer:
            dy = _act_bwd(y, dy, act, slope)
        dy, dct = _sliceable(dy)
        if Co != dy.shape[1]:      # flattened 1x1-input case (see forward): dy [N, co, KH, KW] -> [N, co*KH*KW, 1, 1]
            if dct != dy.shape[1]:
                dy = dy.contiguous()
            dy, dct = dy.view(N, Co, 1, 1), Co
        b = ctx.b

        def weight_grads():
            if w.requires_grad:
                d = _desc(N, Co, OH, OW, Ci, h, wd, k, s, p, dct, xct, ACT_NONE, 0.0)
                nat.check(L.mgvae_conv2d_bwd_weight(ctypes.byref(d), _p(dy), _p(x), _p(grad_slot(w)), _s()), "convT_bwd_weight")
            if b is not None and b.requires_grad:
                nat.check(L.mgvae_channel_sum_accum(_p(dy), N, Co, OH * OW, dct, 0, _p(grad_slot(b)), _s()), "bias_grad")

        if FORK_WGRAD and N >= FORK_MIN_BATCH and ctx.needs_input_grad[0] and (w.requires_grad or (b is not None and b.requires_grad)):
            _wgrad_rr[0] += 1
            with _forked(x, dy, slot=2 + _wgrad_rr[0] % WGRAD_STREAMS):
                weight_grads()
        else:
            weight_grads()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, Ci, h, wd), device=dy.device, dtype=torch.float32)
            d2 = _desc(N, Co, OH, OW, Ci, h, wd, k, s, p, dct, Ci, ACT_NONE, 0.0)
            _conv_fwd(d2, dy, w, None, dx, mask=(x,) + tuple(ctx.in_act) if ctx.in_act else None)
        return dx, None, None, None, None, None, None, None, None, None, None


def conv_transpose2d(x, w, b=None, stride=(1, 1), pad=(0, 0), opad=(0, 0), act=ACT_NONE, slope=0.01, out=None, in_act=None,
                     defer_act_grad=False):
    return _ConvTFn.apply(x, w, b, stride, pad, opad, act, slope, out, in_act, defer_act_grad)


# ====================================================================== instance norm
class _InstNormFn(torch.autograd.Function):
    """nn.InstanceNorm2d(affine) + fused (Leaky)ReLU (graph/encodingBlock.py:17,48,79,110;
    graph/decoder.py:81-83,124-126,173)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, slope, out):
        _need_cuda(x, "instance_norm")
        x = x.contiguous()
        N, C, H, W = x.shape
        y = out if out is not None else torch.empty_like(x)
        yct = _pitch(y)
        stats = torch.empty((N * C * 2,), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_instance_norm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), N, C, H * W, yct, 0,
                                                    eps, act, slope, None, None, None, _s()), "instance_norm_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        act, slope = ctx.cfg
        N, C, H, W = x.shape
        dy, dct = _sliceable(dy)
        dx = torch.empty_like(x)
        dg = grad_slot(gamma) if gamma.requires_grad else None
        db = grad_slot(beta) if beta.requires_grad else None
        nat.check(nat.lib().mgvae_instance_norm_bwd(_p(x), _p(gamma), _p(beta), _p(stats), _p(dy), _p(dx), _p(dg), _p(db),
                                                    N, C, H * W, dct, 0, act, slope, None, None, None, _s()),
                  "instance_norm_bwd")
        return dx, None, None, None, None, None, None


def instance_norm(x, gamma, beta, eps=1e-5, act=ACT_NONE, slope=0.01, out=None):
    return _InstNormFn.apply(x, gamma, beta, eps, act, slope, out)


class _BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d (+ fused ReLU) with torch's training / eval semantics and running statistics
    (graph/bar_discriminator.py:19-23,69,113-114,153)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training, momentum, eps, act, slope):
        _need_cuda(x, "batch_norm")
        x = x.contiguous()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((2 * C,), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_batch_norm_fwd(_p(x), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(y), _p(stats), N, C, H * W,
                                                 1 if training else 0, momentum, eps, act, slope, _s()), "batch_norm_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (training, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        training, act, slope = ctx.cfg
        N, C, H, W = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = grad_slot(gamma) if gamma.requires_grad else None
        db = grad_slot(beta) if beta.requires_grad else None
        nat.check(nat.lib().mgvae_batch_norm_bwd(_p(x), _p(gamma), _p(beta), _p(stats), _p(dy), _p(dx), _p(dg), _p(db), N, C,
                                                 H * W, 1 if training else 0, act, slope, _s()), "batch_norm_bwd")
        return (dx,) + (None,) * 9


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, act=ACT_NONE, slope=0.01):
    return _BatchNormFn.apply(x, gamma, beta, running_mean, running_var, training, momentum, eps, act, slope)


class _GroupSumFn(torch.autograd.Function):
    """sum over groups of `gsize` adjacent entries of the last axis (W = groups * gsize)"""

    @staticmethod
    def forward(ctx, x, gsize):
        _need_cuda(x, "group_sum")
        x = x.contiguous()
        W = x.shape[-1]
        if W % gsize:
            raise RuntimeError("group_sum: last axis %d is not a multiple of %d" % (W, gsize))
        groups, rows = W // gsize, x.numel() // W
        out = torch.empty(tuple(x.shape[:-1]) + (groups,), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_group_sum_fwd(_p(x), _p(out), rows, groups, gsize, _s()), "group_sum_fwd")
        ctx.cfg = (tuple(x.shape), rows, groups, gsize)
        return out

    @staticmethod
    def backward(ctx, dout):
        shape, rows, groups, gsize = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty(shape, device=dout.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_group_sum_bwd(_p(dout), _p(dx), rows, groups, gsize, _s()), "group_sum_bwd")
        return dx, None


def group_sum(x, gsize):
    return _GroupSumFn.apply(x, gsize)


class _MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=2) (graph/refiner.py:16,23)"""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x, "maxpool2")
        x = x.contiguous()
        N, C, H, W = x.shape
        y = torch.empty((N, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, C, H // 2, W // 2), device=x.device, dtype=torch.int32)
        nat.check(nat.lib().mgvae_maxpool2_fwd(_p(x), _p(y), _p(idx), N * C, H, W, _s()), "maxpool2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_maxpool2_bwd(_p(dy), _p(idx), _p(dx), N * C, H, W, _s()), "maxpool2_bwd")
        return dx


def maxpool2(x):
    return _MaxPool2Fn.apply(x)


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _need_cuda(x, "activation")
        x = x.contiguous()
        y = torch.empty_like(x)
        nat.check(nat.lib().mgvae_act_fwd(_p(x), _p(y), x.numel(), act, slope, _s()), "act_fwd")
        ctx.save_for_backward(y)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, slope = ctx.cfg
        y4 = y.reshape(y.shape[0], -1, 1, 1)
        return _act_bwd(y4, dy.reshape(y4.shape), act, slope).reshape(y.shape), None, None


def activation(x, act, slope=0.01):
    return _ActFn.apply(x, act, slope)


class _AxpbyFn(torch.autograd.Function):
    """a * x + b * y"""

    @staticmethod
    def forward(ctx, x, y, a, b):
        _need_cuda(x, "axpby")
        x, y = x.contiguous(), y.contiguous()
        out = torch.empty_like(x)
        nat.check(nat.lib().mgvae_axpby(_p(x), _p(y), _p(out), a, b, x.numel(), _s()), "axpby")
        ctx.ab = (a, b)
        return out

    @staticmethod
    def backward(ctx, d):
        a, b = ctx.ab
        d = d.contiguous()
        z = torch.zeros_like(d)

        def scaled(c):
            if c == 1.0:
                return d
            o = torch.empty_like(d)
            nat.check(nat.lib().mgvae_axpby(_p(d), _p(z), _p(o), c, 0.0, d.numel(), _s()), "axpby_bwd")
            return o
        return scaled(a), scaled(b), None, None


def axpby(x, y, a=1.0, b=1.0):
    return _AxpbyFn.apply(x, y, a, b)


class _CatTimeFn(torch.autograd.Function):
    """torch.cat((a, b), dim=2) for single-channel rolls [B,1,Ha,W] + [B,1,Hb,W] (the 2-bar pairs
    the bar discriminator sees: agent/barGen_with_gan.py:487-490)"""

    @staticmethod
    def forward(ctx, a, b):
        _need_cuda(a, "cat_time")
        a, b = a.contiguous(), b.contiguous()
        B, C, Ha, W = a.shape
        Hb = b.shape[2]
        if C != 1 or b.shape[1] != 1:
            raise RuntimeError("cat_time handles single-channel piano rolls")
        out = torch.empty((B, 1, Ha + Hb, W), device=a.device, dtype=torch.float32)
        L = nat.lib()
        pitch = (Ha + Hb) * W
        nat.check(L.mgvae_copy2d(_p(out), pitch, _p(a), Ha * W, Ha * W, B, _s()), "cat_time")
        nat.check(L.mgvae_copy2d(_vp(out.data_ptr() + 4 * Ha * W), pitch, _p(b), Hb * W, Hb * W, B, _s()), "cat_time")
        ctx.ha = Ha
        return out

    @staticmethod
    def backward(ctx, d):
        return d[:, :, :ctx.ha], d[:, :, ctx.ha:]


def cat_time(a, b):
    return _CatTimeFn.apply(a, b)


# ================================================================================ CBAM
class _CbamFn(torch.autograd.Function):
    """graph/cbam.py CBAM.forward fused with the residual/activation that follows it."""

    @staticmethod
    def forward(ctx, u, res, w1, w2, wsp, mode, act, slope, out, parts):
        _need_cuda(u, "cbam")
        u = u.contiguous()
        N, C, H, W = u.shape
        if res is not None:
            res = res.contiguous()
        L = nat.lib()
        y = out if out is not None else torch.empty_like(u)
        yct = _pitch(y)
        save = torch.empty((L.mgvae_cbam_save_floats(N, C, H, W),), device=u.device, dtype=torch.float32)
        nat.check(L.mgvae_cbam_fwd(_p(u), _p(res), _p(w1), _p(w2), _p(wsp), _p(y), _p(save), N, C, H, W, yct, 0, mode, act,
                                   slope, parts, _s()), "cbam_fwd")
        ctx.save_for_backward(u, y, w1, w2, wsp, save)
        ctx.cfg = (mode, act, slope, parts)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, y, w1, w2, wsp, save = ctx.saved_tensors
        mode, act, slope, parts = ctx.cfg
        N, C, H, W = u.shape
        L = nat.lib()
        yct = _pitch(y)
        dy, dct = _sliceable(dy)
        if dct != yct:    # the kernels address y and dy with one pitch
            y, dy, yct = y.contiguous(), dy.contiguous(), C
        du = torch.empty_like(u)
        dres = torch.empty_like(u) if mode == 2 else None
        scratch = torch.empty((L.mgvae_cbam_bwd_scratch_floats(N, C, H, W),), device=u.device, dtype=torch.float32)
        dw1 = grad_slot(w1) if w1 is not None and w1.requires_grad else None
        dw2 = grad_slot(w2) if w2 is not None and w2.requires_grad else None
        dws = grad_slot(wsp) if wsp is not None and wsp.requires_grad else None
        nat.check(L.mgvae_cbam_bwd(_p(u), _p(y), _p(dy), _p(w1), _p(w2), _p(wsp), _p(save), _p(du), _p(dres), _p(dw1), _p(dw2),
                                   _p(dws), _p(scratch), N, C, H, W, yct, 0, mode, act, slope, parts, _s()), "cbam_bwd")
        return du, dres, None, None, None, None, None, None, None, None


def cbam(u, w1, w2, wsp, mode=0, res=None, act=ACT_NONE, slope=0.01, out=None, parts=3):
    """parts: 3 = full CBAM, 1 = channel attention only (wsp may be None), 2 = spatial attention only"""
    return _CbamFn.apply(u, res, w1, w2, wsp, mode, act, slope, out, parts)


class _NormCbamFn(torch.autograd.Function):
    """InstanceNorm2d -> CBAM -> (+residual) -> activation as ONE autograd node (graph/encodingBlock.py:48-55,
    110-117; graph/decoder.py:124-133,173-176).  The norm kernel also produces CBAM's channel pooling, and the
    norm backward absorbs the tail of the CBAM backward, so the activation map is streamed two times fewer."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, w1, w2, wsp, eps, mode, act, slope, out):
        _need_cuda(x, "norm_cbam")
        x = x.contiguous()
        N, C, H, W = x.shape
        NC = N * C
        if res is not None:
            res = res.contiguous()
        L = nat.lib()
        u = torch.empty_like(x)
        stats = torch.empty((NC * 2,), device=x.device, dtype=torch.float32)
        y = out if out is not None else torch.empty_like(x)
        yct = _pitch(y)
        save = torch.empty((L.mgvae_cbam_save_floats(N, C, H, W),), device=x.device, dtype=torch.float32)
        nat.check(L.mgvae_instance_norm_fwd(_p(x), _p(gamma), _p(beta), _p(u), _p(stats), N, C, H * W, C, 0, eps, ACT_NONE,
                                            0.0, _p(save[NC:]), _p(save[2 * NC:]), _p(save[3 * NC:]), _s()),
                  "instance_norm_fwd")
        nat.check(L.mgvae_cbam_fwd(_p(u), _p(res), _p(w1), _p(w2), _p(wsp), _p(y), _p(save), N, C, H, W, yct, 0, mode, act,
                                   slope, 3 | 4, _s()), "cbam_fwd")
        ctx.save_for_backward(x, gamma, beta, stats, u, y, w1, w2, wsp, save)
        ctx.cfg = (mode, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats, u, y, w1, w2, wsp, save = ctx.saved_tensors
        mode, act, slope = ctx.cfg
        N, C, H, W = x.shape
        NC, NP = N * C, N * H * W
        L = nat.lib()
        yct = _pitch(y)
        dy, dct = _sliceable(dy)
        if dct != yct:
            y, dy, yct = y.contiguous(), dy.contiguous(), C
        du = torch.empty_like(x)
        dres = torch.empty_like(x) if mode == 2 else None
        scratch = torch.empty((L.mgvae_cbam_bwd_scratch_floats(N, C, H, W),), device=x.device, dtype=torch.float32)
        dw1 = grad_slot(w1) if w1.requires_grad else None
        dw2 = grad_slot(w2) if w2.requires_grad else None
        dws = grad_slot(wsp) if wsp.requires_grad else None
        nat.check(L.mgvae_cbam_bwd(_p(u), _p(y), _p(dy), _p(w1), _p(w2), _p(wsp), _p(save), _p(du), _p(dres), _p(dw1), _p(dw2),
                                   _p(dws), _p(scratch), N, C, H, W, yct, 0, mode, act, slope, 3 | 4, _s()), "cbam_bwd")
        dx = torch.empty_like(x)
        dg = grad_slot(gamma) if gamma.requires_grad else None
        db = grad_slot(beta) if beta.requires_grad else None
        davg = scratch[3 * NP + NC:]
        nat.check(L.mgvae_instance_norm_bwd(_p(x), _p(gamma), _p(beta), _p(stats), _p(du), _p(dx), _p(dg), _p(db), N, C,
                                            H * W, C, 0, ACT_NONE, 0.0, _p(davg), _p(davg[NC:]), _p(save[3 * NC:]), _s()),
                  "instance_norm_bwd")
        return (dx, None, None, dres) + (None,) * 8


def norm_cbam(x, gamma, beta, w1, w2, wsp, eps=1e-5, mode=0, res=None, act=ACT_NONE, slope=0.01, out=None):
    return _NormCbamFn.apply(x, gamma, beta, res, w1, w2, wsp, eps, mode, act, slope, out)


# ============================================================================ plumbing
class _JoinFn(torch.autograd.Function):
    """torch.cat(dim=1) with zero copies: the producers already wrote their channel
    slices of ``buf``; backward hands each producer its slice of the gradient."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.sizes = [t.shape[1] for t in parts]
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, dy):
        outs, o = [], 0
        for c in ctx.sizes:
            outs.append(dy[:, o:o + c])
            o += c
        return (None,) + tuple(outs)


def join(buf, *parts):
    return _JoinFn.apply(buf, *parts)


class _RowMeanFn(torch.autograd.Function):
    """nn.AvgPool2d over the whole map (graph/encoder.py:20,35; graph/phrase_encoder.py:21,36)"""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x, "global_avg_pool")
        x = x.contiguous()
        N, C, H, W = x.shape
        out = torch.empty((N, C), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_rowmean_fwd(_p(x), _p(out), N * C, H * W, _s()), "rowmean_fwd")
        ctx.shape = (N, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        N, C, H, W = ctx.shape
        dout = dout.contiguous()
        dx = torch.empty((N, C, H, W), device=dout.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_rowmean_bwd(_p(dout), _p(dx), N * C, H * W, _s()), "rowmean_bwd")
        return dx


def global_avg_pool(x):
    return _RowMeanFn.apply(x)


class _EmbeddingFn(torch.autograd.Function):
    """nn.Embedding gather (graph/decoder.py:187,193), optionally into a column slice"""

    @staticmethod
    def forward(ctx, idx, table, out):
        if not table.is_cuda:
            raise RuntimeError("embedding: expected a ROCm device tensor; no CPU fallback")
        idx = idx.to(device=table.device, dtype=torch.int64).contiguous()
        B, (rows, D) = idx.numel(), table.shape
        y = out if out is not None else torch.empty((B, D), device=table.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_embedding_fwd(_p(idx), _p(table), _p(y), B, D, rows, y.stride(0), _s()), "embedding_fwd")
        ctx.save_for_backward(idx, table)
        return y

    @staticmethod
    def backward(ctx, dout):
        idx, table = ctx.saved_tensors
        if table.requires_grad:
            if dout.stride(1) != 1:
                dout = dout.contiguous()
            rows, D = table.shape
            nat.check(nat.lib().mgvae_embedding_bwd(_p(idx), _p(dout), _p(grad_slot(table)), idx.numel(), D, rows,
                                                    dout.stride(0), _s()), "embedding_bwd")
        return None, None, None


def embedding(idx, table, out=None):
    return _EmbeddingFn.apply(idx, table, out)


class _CopyIntoFn(torch.autograd.Function):
    """dst[:, :] <- src for 2-D [B, D] tensors where dst is a column slice of a wider
    buffer (a concat member that was produced elsewhere, e.g. a user-supplied latent)"""

    @staticmethod
    def forward(ctx, src, dst):
        _need_cuda(src, "copy_into")
        src = src.contiguous()
        B, D = src.shape
        nat.check(nat.lib().mgvae_copy2d(_p(dst), dst.stride(0), _p(src), D, D, B, _s()), "copy2d")
        return dst.view(dst.shape)

    @staticmethod
    def backward(ctx, d):
        return d, None


def copy_into(src, dst_view):
    return _CopyIntoFn.apply(src, dst_view)


_rng_state = {"seed": 0x1234ABCD, "offset": 0}


def manual_seed(seed, rank=0):
    """seed the Philox streams used by dropout / prior noise (independent per rank)"""
    _rng_state["seed"] = (int(seed) * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    _rng_state["offset"] = 0


def _next_offset():
    _rng_state["offset"] += 1
    return _rng_state["offset"]


class _DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) (graph/decoder.py:164,196,201) on the build's own Philox stream"""

    @staticmethod
    def forward(ctx, x, p, mask_in):
        _need_cuda(x, "dropout")
        x = x.contiguous()
        y = torch.empty_like(x)
        if mask_in is None:
            mask = torch.empty_like(x)
            nat.check(nat.lib().mgvae_dropout_fwd(_p(x), _p(y), _p(mask), x.numel(), p, _rng_state["seed"], _next_offset(),
                                                  _s()), "dropout_fwd")
        else:
            mask = mask_in.contiguous()
            nat.check(nat.lib().mgvae_mul(_p(x), _p(mask), _p(y), x.numel(), _s()), "dropout_mask")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        nat.check(nat.lib().mgvae_mul(_p(dy), _p(mask), _p(dx), dy.numel(), _s()), "dropout_bwd")
        return dx, None, None


def dropout(x, p=0.3, training=True, mask=None):
    if mask is None and (not training or p == 0.0):
        return x
    return _DropoutFn.apply(x, p, mask)


def randn(shape, sigma=1.0, device="cuda", out=None):
    """Gaussian prior noise N(0, sigma^2) generated on the device (agent/barGen2.py:243,250); ``out``: fill this
    dense fp32 tensor instead of allocating (static buffers of a captured graph)"""
    if out is None:
        out = torch.empty(shape, device=device, dtype=torch.float32)
    nat.check(nat.lib().mgvae_randn(_p(out), out.numel(), sigma, _rng_state["seed"], _next_offset(), _s()), "randn")
    return out


def unpack_bits(packed, shape):
    """bit-packed {0,1} rolls (uint8 device tensor, LSB first) -> fp32 tensor of ``shape`` (input pipeline)"""
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("unpack_bits: expected a uint8 ROCm device tensor")
    packed = packed.contiguous()
    out = torch.empty(shape, device=packed.device, dtype=torch.float32)
    if out.numel() > packed.numel() * 8:
        raise RuntimeError("unpack_bits: %d cells do not fit %d packed bytes" % (out.numel(), packed.numel()))
    nat.check(nat.lib().mgvae_unpack_bits(_p(packed), _p(out), out.numel(), _s()), "unpack_bits")
    return out


# ============================================================================== losses
_prior_cache = {}


def _prior(device, prior_np):
    key = (str(device), id(prior_np))
    t = _prior_cache.get(key)
    if t is None:
        t = torch.from_numpy(np.asarray(prior_np, dtype=np.float32)).to(device)
        _prior_cache[key] = t
    return t


class _BceFn(torch.autograd.Function):
    """nn.BCELoss (mean) as used by graph/loss/bar_loss.py Loss / DLoss"""

    @staticmethod
    def forward(ctx, x, targets, prior, tconst, mode, count_term):
        _need_cuda(x, "bce")
        x = x.contiguous()
        if targets is not None:
            targets = targets.contiguous()
        L = nat.lib()
        partial = torch.empty((L.mgvae_bce_partial_floats(),), device=x.device, dtype=torch.float32)
        out = torch.empty((1,), device=x.device, dtype=torch.float32)
        nat.check(L.mgvae_bce_fwd(_p(x), _p(targets), _p(prior), tconst, x.numel(), mode, count_term, _p(partial), _p(out),
                                  _s()), "bce_fwd")
        ctx.save_for_backward(x, targets, prior)
        ctx.cfg = (tconst, mode)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        x, targets, prior = ctx.saved_tensors
        tconst, mode = ctx.cfg
        g = g.contiguous()
        dx = torch.empty_like(x)
        nat.check(nat.lib().mgvae_bce_bwd(_p(x), _p(targets), _p(prior), tconst, x.numel(), mode, _p(g), _p(dx), _s()), "bce_bwd")
        return dx, None, None, None, None, None


def bce(x, targets):
    """F.binary_cross_entropy(x, targets) -- mean reduction, log clamp at -100"""
    return _BceFn.apply(x, targets, None, 0.0, 0, 0)


def bce_const(x, value):
    """BCE against an all-``value`` target (DLoss with valid/fake targets)"""
    return _BceFn.apply(x, None, None, float(value), 2, 0)


def bar_recon_loss(gen, labels, prior_scaled, is_pretraining):
    """graph/loss/bar_loss.py:23-33 in one pass: BCE (plain or label-smoothed) + 0.005 * #missed notes"""
    if is_pretraining:
        return _BceFn.apply(gen, labels, None, 0.0, 0, 1)
    return _BceFn.apply(gen, labels, _prior(gen.device, prior_scaled), 0.0, 1, 1)


class _ReparamKlFn(torch.autograd.Function):
    """old/graphs/models/bar_v1/encoder.py:60-63 + old/graphs/losses/loss.py:14-17"""

    @staticmethod
    def forward(ctx, mean, logvar, eps):
        _need_cuda(mean, "reparam_kl")
        mean, logvar, eps = mean.contiguous(), logvar.contiguous(), eps.contiguous()
        L = nat.lib()
        z = torch.empty_like(mean)
        partial = torch.empty((L.mgvae_bce_partial_floats(),), device=mean.device, dtype=torch.float32)
        kl = torch.empty((1,), device=mean.device, dtype=torch.float32)
        nat.check(L.mgvae_reparam_kl_fwd(_p(mean), _p(logvar), _p(eps), _p(z), _p(partial), _p(kl), mean.numel(), _s()),
                  "reparam_kl_fwd")
        ctx.save_for_backward(mean, logvar, eps)
        return z, kl.view(())

    @staticmethod
    def backward(ctx, dz, dkl):
        mean, logvar, eps = ctx.saved_tensors
        dz = dz.contiguous() if dz is not None else torch.zeros_like(mean)
        dkl = dkl.contiguous() if dkl is not None else torch.zeros((), device=mean.device)
        dm, dlv = torch.empty_like(mean), torch.empty_like(mean)
        nat.check(nat.lib().mgvae_reparam_kl_bwd(_p(mean), _p(logvar), _p(eps), _p(dz), _p(dkl), _p(dm), _p(dlv), mean.numel(),
                                                 _s()), "reparam_kl_bwd")
        return dm, dlv, None


def reparam_kl(mean, logvar, eps=None):
    """z = mean + eps * exp(0.5 logvar); kl = -0.5 sum(1 + logvar - mean^2 - exp(logvar))"""
    if eps is None:
        eps = randn(tuple(mean.shape), 1.0, mean.device)
    return _ReparamKlFn.apply(mean, logvar, eps)


# ====================================================================== channels-last (NHWC) family
# Activations stored [N, H, W, C] and conv weights stored [Cy, KH, KW, Cx] -- torch.channels_last for both, so logical
# shapes, state_dict entries and everything that indexes tensors logically (tests, checkpoints) are unchanged.  The GEMM's
# K axis (tap, channel) is then contiguous in memory: csrc/conv_nhwc.inc.  The encoder trunks and the decoder's blocks run
# in this layout (graph/encoder.py, graph/decoder.py); ``to_channels_last`` / ``to_nchw`` convert at the ends of the island.
#
# Storage type of the island: fp32, or -- with set_compute_dtype("bf16"), BASELINE.json configs 3-4 -- bf16: activations
# and their gradients are bf16 tensors, the convs run on the bf16 matrix pipe from bf16 copies of the fp32 master weights
# (csrc/conv_nhwc_bf16.inc), statistics / gates / accumulators / weight gradients / Adam stay fp32.  The ops below take the
# storage type from the tensors they are given.
CL = torch.channels_last
STORE_F32, STORE_BF16 = 0, 1


def island_dtype():
    """storage type of channels-last activations created at the entry of an island"""
    return torch.bfloat16 if get_compute_dtype() == "bf16" else torch.float32


def _store(t):
    return STORE_BF16 if t.dtype == torch.bfloat16 else STORE_F32


def cl_pitch(t):
    """channel pitch (elements between consecutive pixels) of an NCHW-shaped tensor whose memory is channels-last --
    dense, or a channel slice of a wider channels-last buffer; None for any other layout"""
    n, c, h, w = t.shape
    st = t.stride()
    if c > 1 and st[1] != 1:
        return None
    if w > 1:
        ct = st[3]
    elif h > 1:
        ct = st[2]
    elif n > 1:
        ct = st[0]
    else:
        ct = c
    if ct < c or (h > 1 and st[2] != w * ct) or (n > 1 and st[0] != h * w * ct):
        return None
    return ct


def new_channels_last(n, c, h, w, device, dtype=torch.float32):
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def _need_cl(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: expected a ROCm device tensor; the MI355X hot path has no CPU fallback" % what)
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("%s: expected float32 or bfloat16, got %s" % (what, t.dtype))
    ct = cl_pitch(t)
    if ct is None or (ct % (8 if t.dtype == torch.bfloat16 else 4)) or (t.data_ptr() % 16):
        raise RuntimeError("%s: expected a channels-last tensor (stride %s of shape %s is not)" % (what, t.stride(), tuple(t.shape)))
    return ct


def _same_storage(x, y, what):
    if x.dtype != y.dtype:
        raise RuntimeError("%s: input is stored as %s but the output buffer as %s" % (what, x.dtype, y.dtype))


def _cl_weight(w, what):
    co, ci, kh, kw = w.shape
    if w.stride() != (kh * kw * ci, 1, kw * ci, ci) and not (kh == 1 and kw == 1 and w.stride(0) == ci and w.stride(1) == 1):
        raise RuntimeError("%s: the channels-last kernels need the weight stored [Cy, KH, KW, Cx] (torch.channels_last): build "
                           "the layer with channels_last=True" % what)


def _cl_mask(t, act, slope):
    ct = _need_cl(t, "activation mask")
    return nat.ActMask(t.data_ptr(), ct, 0, act, slope), t


class _WeightCopies:
    """the matrix-pipe copies of a network's conv weights -- planes = 1: bf16 (wk [Cy, T, Cx], wt [Cx, T, Cy]) for the
    bf16-storage island; planes = 3: the three-way split (wk3: rows Cy, wt3: rows Cx; 3 * numel elements each, layout owned by the library -- include/mgvae.h) of the fp32 island
    (csrc/conv_nhwc_x3.inc) -- refreshed ONCE per optimizer step.  A weight inside a FlatParams buffer registers here the
    first time a conv uses it (and is packed alone that once); from the next step on all registered weights are repacked
    by one grouped launch when the first of them is asked for, and every other stream that asks waits for that launch's
    event.  (A parameter outside a flat buffer is repacked whenever torch's version counter moves.)"""

    def __init__(self, planes):
        self.planes = planes
        self.copies = {}          # id(w) -> (w, wk, wt)
        self.table = None         # device table of the grouped launch, rebuilt when the set of weights changed
        self.blocks = 0
        self.version = None       # owner.weights_version the copies belong to
        self.event = None
        self.waited = set()

    def _alloc(self, w):
        n = self.planes * w.numel()
        return (torch.empty(n, device=w.device, dtype=torch.bfloat16), torch.empty(n, device=w.device, dtype=torch.bfloat16))

    def _pack_one(self, w, wk, wt):
        cy, cx, kh, kw = w.shape
        fn = nat.lib().mgvae_pack_conv_weights_x3 if self.planes == 3 else nat.lib().mgvae_pack_conv_weights_bf16
        nat.check(fn(_p(w), _p(wk), _p(wt), cy, kh * kw, cx, _s()), "pack_conv_weights")

    def _build_table(self, device):
        rec = np.zeros(len(self.copies), dtype=np.dtype([("w", "<u8"), ("wk", "<u8"), ("wt", "<u8"), ("Cy", "<i4"), ("T", "<i4"),
                                                       ("Cx", "<i4"), ("block0", "<i4")]))
        blocks = 0
        for i, (w, wk, wt) in enumerate(self.copies.values()):
            cy, cx, kh, kw = w.shape
            rec[i] = (w.data_ptr(), wk.data_ptr(), wt.data_ptr(), cy, kh * kw, cx, blocks)
            blocks += kh * kw * ((cy + 31) // 32) * ((cx + 31) // 32)
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
        self.blocks = blocks

    def get(self, w, version):
        key = id(w)
        cur = (version, w._version)           # the owner counts optimizer steps; torch counts in-place edits (load_state_dict ...)
        if key not in self.copies:
            wk, wt = self._alloc(w)
            self.copies[key] = (w, wk, wt)
            self.table = None
            self._pack_one(w, wk, wt)
            w._mg_copy_version = cur
            return wk, wt
        _, wk, wt = self.copies[key]
        if getattr(w, "_mg_copy_version", None) == cur:
            self._wait()
            return wk, wt
        if self.version != version:
            if self.table is None:
                self._build_table(w.device)
            nat.check(nat.lib().mgvae_pack_conv_weights_grouped(_p(self.table), len(self.copies), self.blocks, self.planes, _s()),
                      "pack_conv_weights_grouped")
            self.version = version
            self.event = torch.cuda.Event()
            self.event.record(cur_stream())
            self.waited = {cur_raw_stream()}
            for ww, _, _ in self.copies.values():
                ww._mg_copy_version = (version, ww._version)
        else:                                  # edited by a torch op inside the step: this one alone
            self._pack_one(w, wk, wt)
            w._mg_copy_version = cur
        self._wait()
        return wk, wt

    def _wait(self):
        if self.event is None:
            return
        sid = cur_raw_stream()
        if sid not in self.waited:
            cur_stream().wait_event(self.event)
            self.waited.add(sid)


def _weight_copies(w, planes, attr):
    owner = getattr(w, "_mg_owner", None)
    if owner is None:
        # a free-standing parameter: repacked whenever torch's version counter moves
        ver = (w._version, w.data_ptr())
        c = getattr(w, attr, None)
        if c is None or c[0] != ver or torch.cuda.is_current_stream_capturing():
            g = _WeightCopies(planes)
            wk, wt = (c[1], c[2]) if c is not None else g._alloc(w)
            g._pack_one(w, wk, wt)
            c = (ver, wk, wt)
            setattr(w, attr, c)
        return c[1], c[2]
    groups = owner.__dict__.setdefault("_mg_weight_copies", {})
    g = groups.get(planes)
    if g is None:
        g = groups[planes] = _WeightCopies(planes)
    return g.get(w, owner.weights_version[0])


def _bf16_weights(w):
    """(wk, wt): bf16 copies of the fp32 channels-last master weight ``w`` -- wk [Cy, T, Cx] for the forward product, wt
    [Cx, T, Cy] for the data gradient (see _WeightCopies)."""
    return _weight_copies(w, 1, "_mg_bf16")


# fp32 islands: which matrix instruction multiplies.  "x3" (default): csrc/conv_nhwc_x3.inc -- every fp32 operand enters the
# bf16 matrix pipe as the exact sum of three bf16 values, every product as six bf16 MFMAs (fp32-grade: relative error
# < 2^-22 per product, fp32 accumulation; tests/test_nhwc_gpu.py holds it to three times the fp32 instruction's own error);
# "mfma32": csrc/conv_nhwc.inc on v_mfma_f32_32x32x2_f32.  Same tensors, same results to fp32 rounding.
FP32_ENGINE = _os.environ.get("MGVAE_FP32_ENGINE", "x3")
if FP32_ENGINE not in ("x3", "mfma32"):
    raise RuntimeError("MGVAE_FP32_ENGINE must be x3 or mfma32, got %r" % FP32_ENGINE)


def _x3_ok(cx, cy):
    return FP32_ENGINE == "x3" and cx % 16 == 0 and cy % 16 == 0


def _x3_weights(w):
    """(wk3, wt3): the fp32 channels-last master weight as three bf16 planes each way -- wk3 (rows Cy, k Cx) for the forward
    product, wt3 [3, Cx, T, Cy] for the data gradient (see _WeightCopies)."""
    return _weight_copies(w, 3, "_mg_x3")


_ws_bytes = {}      # (family, mode, geometry) -> bytes of the largest deterministic split-K the tuner may pick (0: never split)


def _split_ws(fam, d, mode, device):
    """the CALLER-OWNED split-K workspace of one channels-last forward / data-gradient launch (include/mgvae.h:
    mgvae_conv2d_nhwc_{x3,bf16}_workspace): a tensor from torch's caching allocator on the launching stream (the library
    allocates nothing), or None when the geometry never splits.  Returns (tensor-or-None, pointer, bytes)."""
    key = (fam, mode, d.N, d.Cx, d.H, d.W, d.Cy, d.KH, d.KW, d.SH, d.SW, d.PH, d.PW)
    n = _ws_bytes.get(key)
    if n is None:
        fn = nat.lib().mgvae_conv2d_nhwc_x3_workspace if fam == "x3" else nat.lib().mgvae_conv2d_nhwc_bf16_workspace
        n = _ws_bytes[key] = int(fn(ctypes.byref(d), mode))
    if not n:
        return None, None, 0
    t = torch.empty(n, device=device, dtype=torch.uint8)
    return t, _vp(t.data_ptr()), n


class _ToChannelsLastFn(torch.autograd.Function):
    """NCHW fp32 (dense or a channel slice) -> dense channels-last copy of the island's storage type; backward converts the
    gradient back to NCHW fp32"""

    @staticmethod
    def forward(ctx, x, dtype):
        _need_cuda(x, "to_channels_last")
        x, xct = _sliceable(x)
        N, C, H, W = x.shape
        y = new_channels_last(N, C, H, W, x.device, dtype)
        nat.check(nat.lib().mgvae_layout_nchw_to_nhwc(_p(x), _p(y), N, C, H * W, xct, 0, C, 0, _store(y), _s()), "nchw_to_nhwc")
        return y

    @staticmethod
    def backward(ctx, dy):
        ct = _need_cl(dy, "to_channels_last backward")
        N, C, H, W = dy.shape
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_layout_nhwc_to_nchw(_p(dy), _p(dx), N, C, H * W, ct, 0, C, 0, _store(dy), _s()), "nhwc_to_nchw")
        return dx, None


class _ToNchwFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ct = _need_cl(x, "to_nchw")
        N, C, H, W = x.shape
        y = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_layout_nhwc_to_nchw(_p(x), _p(y), N, C, H * W, ct, 0, C, 0, _store(x), _s()), "nhwc_to_nchw")
        ctx.dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy, dct = _sliceable(dy)
        N, C, H, W = dy.shape
        dx = new_channels_last(N, C, H, W, dy.device, ctx.dtype)
        nat.check(nat.lib().mgvae_layout_nchw_to_nhwc(_p(dy), _p(dx), N, C, H * W, dct, 0, C, 0, _store(dx), _s()), "nchw_to_nhwc")
        return dx


def to_channels_last(x, dtype=None):
    return _ToChannelsLastFn.apply(x, dtype if dtype is not None else island_dtype())


def to_nchw(x):
    return _ToNchwFn.apply(x)


def _as_cl(t, like):
    """gradient tensor ``t`` as a channels-last tensor of ``like``'s storage type (autograd may hand over another layout)"""
    if t.dtype != like.dtype:
        t = t.to(like.dtype)
    if cl_pitch(t) is None:
        t = t.contiguous(memory_format=CL)
    return t


class _ConvClFn(torch.autograd.Function):
    """nn.Conv2d on channels-last tensors (graph/encodingBlock.py:74-77,107-108): forward, data gradient (stride
    phases) and weight gradient of csrc/conv_nhwc.inc (fp32) / csrc/conv_nhwc_bf16.inc (bf16 storage); the weight is
    stored [Cy, KH, KW, Cx] and its gradient is accumulated in that same layout (fp32 in both modes)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, slope, out, in_act=None, defer_act_grad=False):
        xct = _need_cl(x, "conv2d (channels-last)")
        _cl_weight(w, "conv2d (channels-last)")
        N, Cx, H, W = x.shape
        Cy, _, KH, KW = w.shape
        OH = (H + 2 * pad[0] - KH) // stride[0] + 1
        OW = (W + 2 * pad[1] - KW) // stride[1] + 1
        y = out if out is not None else new_channels_last(N, Cy, OH, OW, x.device, x.dtype)
        yct = _need_cl(y, "conv2d (channels-last) output")
        _same_storage(x, y, "conv2d (channels-last)")
        d = _desc(N, Cx, H, W, Cy, OH, OW, (KH, KW), stride, pad, xct, yct, act, slope)
        if x.dtype == torch.bfloat16:
            wk, _ = _bf16_weights(w)
            wst, wsp, wsn = _split_ws("bf16", d, 0, x.device)
            nat.check(nat.lib().mgvae_conv2d_nhwc_bf16_fwd(ctypes.byref(d), _p(x), _p(wk), _p(b), _p(y), None, wsp, wsn, _s()), "conv2d_nhwc_bf16_fwd")
        elif _x3_ok(Cx, Cy):
            wk3, _ = _x3_weights(w)
            wst, wsp, wsn = _split_ws("x3", d, 0, x.device)
            nat.check(nat.lib().mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d), _p(x), _p(wk3), _p(b), _p(y), None, wsp, wsn, _s()), "conv2d_nhwc_x3_fwd")
        else:
            nat.check(nat.lib().mgvae_conv2d_nhwc_fwd(ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), None, _s()), "conv2d_nhwc_fwd")
        ctx.geom = (N, Cx, H, W, Cy, OH, OW, (KH, KW), stride, pad, xct, act, slope)
        if not DEFER_ACT_GRAD:
            in_act, defer_act_grad = None, False
        ctx.in_act, ctx.defer = in_act, bool(defer_act_grad)          # see _ConvFn.forward
        ctx.save_for_backward(x, w, y if (act != ACT_NONE and not defer_act_grad) else None)
        ctx.b = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N, Cx, H, W, Cy, OH, OW, k, s, p, xct, act, slope = ctx.geom
        L = nat.lib()
        bf = x.dtype == torch.bfloat16
        if _trunk_streams or _used_sides:
            _ensure_join_callback()
        dy = _as_cl(dy, x)
        if act != ACT_NONE and not ctx.defer:
            dy = _act_bwd_cl(y, dy, act, slope)
        dct = _need_cl(dy, "conv2d (channels-last) backward")
        d = _desc(N, Cx, H, W, Cy, OH, OW, k, s, p, xct, dct, ACT_NONE, 0.0)
        b = ctx.b

        def weight_grads():
            if w.requires_grad:
                fn = L.mgvae_conv2d_nhwc_bf16_bwd_weight if bf else (L.mgvae_conv2d_nhwc_x3_bwd_weight if _x3_ok(Cx, Cy) else L.mgvae_conv2d_nhwc_bwd_weight)
                nat.check(fn(ctypes.byref(d), _p(x), _p(dy), _p(grad_slot(w)), _s()), "conv2d_nhwc_bwd_weight")
            if b is not None and b.requires_grad:
                nat.check(L.mgvae_channel_sum_nhwc_accum(_p(dy), N * OH * OW, Cy, dct, 0, _p(grad_slot(b)), _store(dy), _s()), "bias_grad_nhwc")

        if FORK_WGRAD and N >= FORK_MIN_BATCH and ctx.needs_input_grad[0] and w.requires_grad:
            _wgrad_rr[0] += 1
            with _forked(x, dy, slot=2 + _wgrad_rr[0] % WGRAD_STREAMS):
                weight_grads()
        else:
            weight_grads()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_channels_last(N, Cx, H, W, dy.device, x.dtype)
            d2 = _desc(N, Cx, H, W, Cy, OH, OW, k, s, p, Cx, dct, ACT_NONE, 0.0)
            m = None
            if ctx.in_act:
                m, keep = _cl_mask(x, *ctx.in_act)
            mref = ctypes.byref(m) if m is not None else None
            if bf:
                _, wt = _bf16_weights(w)
                wst, wsp, wsn = _split_ws("bf16", d2, 1, dy.device)
                nat.check(L.mgvae_conv2d_nhwc_bf16_bwd_data(ctypes.byref(d2), _p(dy), _p(wt), None, _p(dx), mref, wsp, wsn, _s()), "conv2d_nhwc_bf16_bwd_data")
            elif _x3_ok(Cx, Cy):
                _, wt3 = _x3_weights(w)
                wst, wsp, wsn = _split_ws("x3", d2, 1, dy.device)
                nat.check(L.mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d2), _p(dy), _p(wt3), None, _p(dx), mref, wsp, wsn, _s()), "conv2d_nhwc_x3_bwd_data")
            else:
                nat.check(L.mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d2), _p(dy), _p(w), None, _p(dx), mref, _s()), "conv2d_nhwc_bwd_data")
        return dx, None, None, None, None, None, None, None, None, None


def _act_bwd_cl(y, dy, act, slope):
    """dx = dy * act'(y) for fp32 channels-last tensors (dense result): the NCHW kernel with rows of C channels at a pitch"""
    if y.dtype != torch.float32:
        raise RuntimeError("act_bwd on bf16 channels-last tensors is not on the hot path (activations are fused or deferred there)")
    yct = _need_cl(y, "act_bwd"); dct = _need_cl(dy, "act_bwd")
    n, c, h, w = y.shape
    dx = new_channels_last(n, c, h, w, y.device)
    rows = n * h * w
    nat.check(nat.lib().mgvae_act_bwd(_p(y), _p(dy), _p(dx), rows, c, 1, yct, 0, dct, 0, c, 0, act, slope, _s()), "act_bwd")
    return dx


def conv2d_cl(x, w, b=None, stride=(1, 1), pad=(0, 0), act=ACT_NONE, slope=0.01, out=None, in_act=None, defer_act_grad=False):
    return _ConvClFn.apply(x, w, b, stride, pad, act, slope, out, in_act, defer_act_grad)


class _CastClFn(torch.autograd.Function):
    """a dense channels-last tensor in the other storage type (fp32 <-> bf16); the gradient comes back in the input's type"""

    @staticmethod
    def forward(ctx, x, dtype):
        ct = _need_cl(x, "cast_cl")
        N, C, H, W = x.shape
        if ct != C:
            raise RuntimeError("cast_cl: dense tensors only")
        ctx.src = x.dtype
        y = new_channels_last(N, C, H, W, x.device, dtype)
        nat.check(nat.lib().mgvae_cast_storage(_p(x), _store(x), _p(y), _store(y), x.numel(), _s()), "cast_storage")
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = dy.shape
        if cl_pitch(dy) != C:
            dy = dy.contiguous(memory_format=CL)
        dx = new_channels_last(N, C, H, W, dy.device, ctx.src)
        nat.check(nat.lib().mgvae_cast_storage(_p(dy), _store(dy), _p(dx), _store(dx), dy.numel(), _s()), "cast_storage")
        return dx, None


def cast_cl(x, dtype):
    return x if x.dtype == dtype else _CastClFn.apply(x, dtype)


class _ConvC1ClFn(torch.autograd.Function):
    """A conv of a ONE-channel NCHW fp32 map into a channels-last tensor of the island's storage type, activation fused
    (csrc/thin_nhwc.hip) -- the encoder stems' first convs (graph/encodingBlock.py:11-14,42-45).  The input carries no
    gradient (it is data); with ``defer_act_grad`` the consumer's data gradient has already applied act'(y), otherwise the
    weight-gradient kernel multiplies it in while it reads dy."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, act, slope, dtype, defer_act_grad):
        _need_cuda(x, "conv2d_c1_cl")
        if x.requires_grad:
            raise RuntimeError("conv2d_c1_cl: the one-channel input is data (no gradient path)")
        x = x.contiguous()
        N, Cx, H, W = x.shape
        Cy, _, KH, KW = w.shape
        if Cx != 1 or x.dtype != torch.float32:
            raise RuntimeError("conv2d_c1_cl: expected an fp32 [N,1,H,W] input, got %s %s" % (tuple(x.shape), x.dtype))
        OH = (H + 2 * pad[0] - KH) // stride[0] + 1
        OW = (W + 2 * pad[1] - KW) // stride[1] + 1
        y = new_channels_last(N, Cy, OH, OW, x.device, dtype)
        wc = w if w.is_contiguous() else w.contiguous()
        d = _desc(N, 1, H, W, Cy, OH, OW, (KH, KW), stride, pad, 1, Cy, act, slope)
        nat.check(nat.lib().mgvae_conv2d_c1_nhwc_fwd(ctypes.byref(d), _p(x), _p(wc), _p(y), _store(y), _s()), "conv2d_c1_nhwc_fwd")
        defer = bool(defer_act_grad) and DEFER_ACT_GRAD
        ctx.geom = (N, H, W, Cy, OH, OW, (KH, KW), stride, pad, act, slope, defer)
        ctx.save_for_backward(x, w, y if (act != ACT_NONE and not defer) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N, H, W, Cy, OH, OW, k, s, p, act, slope, defer = ctx.geom
        if w.requires_grad:
            if _trunk_streams or _used_sides:
                _ensure_join_callback()
            like = y if y is not None else dy
            dy = _as_cl(dy, like)
            if _need_cl(dy, "conv2d_c1_cl backward") != Cy:
                dy = dy.contiguous(memory_format=CL)
            if not w.is_contiguous():
                raise RuntimeError("conv2d_c1_cl: the weight gradient is accumulated in the reference's [Cy,1,KH,KW] layout")
            d = _desc(N, 1, H, W, Cy, OH, OW, k, s, p, 1, Cy, act, slope)
            nat.check(nat.lib().mgvae_conv2d_c1_nhwc_bwd_weight(ctypes.byref(d), _p(x), _p(dy), _p(y) if y is not None else None,
                                                              _p(grad_slot(w)), _store(dy), _s()), "conv2d_c1_nhwc_bwd_weight")
        return None, None, None, None, None, None, None, None


def conv2d_c1_cl(x, w, stride, pad, act=ACT_NONE, slope=0.01, dtype=None, defer_act_grad=False):
    return _ConvC1ClFn.apply(x, w, tuple(stride), tuple(pad), act, slope, dtype if dtype is not None else torch.float32, defer_act_grad)


class _ConvTo1ClFn(torch.autograd.Function):
    """A bias-free 1x1 conv of a channels-last tensor into ONE fp32 channel with its activation (csrc/thin_nhwc.hip) -- the
    decoder's fit2 + Sigmoid (graph/decoder.py:186,217): [N,C,H,W] channels-last -> [N,1,H,W] fp32."""

    @staticmethod
    def forward(ctx, x, w, act, slope):
        xct = _need_cl(x, "conv2d_to1_cl")
        N, C, H, W = x.shape
        if tuple(w.shape) != (1, C, 1, 1):
            raise RuntimeError("conv2d_to1_cl: weight %s does not map %d channels to one" % (tuple(w.shape), C))
        y = torch.empty((N, 1, H, W), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_conv2d_to1_nhwc_fwd(_p(x), _p(w), _p(y), N * H * W, C, xct, 0, act, slope, _store(x), _s()), "conv2d_to1_nhwc_fwd")
        ctx.cfg = (act, slope, xct)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        act, slope, xct = ctx.cfg
        N, C, H, W = x.shape
        if _trunk_streams or _used_sides:
            _ensure_join_callback()
        dy = dy.contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], w.requires_grad
        dx = new_channels_last(N, C, H, W, x.device, x.dtype) if need_dx else None
        if need_dx or need_dw:
            if need_dx and xct != C:
                raise RuntimeError("conv2d_to1_cl: the input gradient is written dense")
            nat.check(nat.lib().mgvae_conv2d_to1_nhwc_bwd(_p(x), _p(w), _p(y), _p(dy), _p(dx) if need_dx else None,
                                                        _p(grad_slot(w)) if need_dw else None, N * H * W, C, xct, 0, act, slope, _store(x), _s()),
                      "conv2d_to1_nhwc_bwd")
        return dx, None, None, None


def conv2d_to1_cl(x, w, act=ACT_NONE, slope=0.01):
    return _ConvTo1ClFn.apply(x, w, act, slope)


class _NormCbamClFn(torch.autograd.Function):
    """InstanceNorm2d -> CBAM -> (+residual) -> activation on channels-last tensors as ONE node
    (graph/encodingBlock.py:48-55,93-100,110-126): csrc/norm_cbam_nhwc.inc"""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, w1, w2, wsp, eps, mode, act, slope, out):
        xct = _need_cl(x, "norm_cbam (channels-last)")
        N, C, H, W = x.shape
        if xct != C:
            raise RuntimeError("norm_cbam (channels-last): the normalised tensor must be dense")
        rct = 0
        if res is not None:
            rct = _need_cl(res, "norm_cbam residual")
            if res.dtype != x.dtype:
                raise RuntimeError("norm_cbam (channels-last): residual and input differ in storage type")
        L = nat.lib()
        y = out if out is not None else new_channels_last(N, C, H, W, x.device, x.dtype)
        yct = _need_cl(y, "norm_cbam output")
        _same_storage(x, y, "norm_cbam (channels-last)")
        save = torch.empty((L.mgvae_norm_cbam_nhwc_save_floats(N, C, H, W),), device=x.device, dtype=torch.float32)
        nat.check(L.mgvae_norm_cbam_nhwc_fwd(_p(x), _p(gamma), _p(beta), _p(res), rct, 0, _p(w1), _p(w2), _p(wsp), _p(y), _p(save),
                                             N, C, H, W, yct, 0, eps, mode, act, slope, _store(x), _s()), "norm_cbam_nhwc_fwd")
        ctx.save_for_backward(x, gamma, beta, y, w1, w2, wsp, save)
        ctx.cfg = (mode, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, y, w1, w2, wsp, save = ctx.saved_tensors
        mode, act, slope = ctx.cfg
        N, C, H, W = x.shape
        L = nat.lib()
        yct = cl_pitch(y)
        dy = _as_cl(dy, x)
        if cl_pitch(dy) != yct:          # the kernels address y and dy with one pitch
            y = y.contiguous(memory_format=CL); dy = dy.contiguous(memory_format=CL); yct = C
        dx = new_channels_last(N, C, H, W, x.device, x.dtype)
        dres = new_channels_last(N, C, H, W, x.device, x.dtype) if mode == 2 else None
        scratch = torch.empty((L.mgvae_norm_cbam_nhwc_scratch_floats(N, C, H, W),), device=x.device, dtype=torch.float32)
        dg = grad_slot(gamma) if gamma.requires_grad else None
        db = grad_slot(beta) if beta.requires_grad else None
        dw1 = grad_slot(w1) if w1.requires_grad else None
        dw2 = grad_slot(w2) if w2.requires_grad else None
        dws = grad_slot(wsp) if wsp.requires_grad else None
        nat.check(L.mgvae_norm_cbam_nhwc_bwd(_p(x), _p(gamma), _p(beta), _p(y), _p(dy), _p(w1), _p(w2), _p(wsp), _p(save), _p(dx),
                                             _p(dres), _p(dg), _p(db), _p(dw1), _p(dw2), _p(dws), _p(scratch), N, C, H, W, yct, 0,
                                             mode, act, slope, _store(x), _s()), "norm_cbam_nhwc_bwd")
        return (dx, None, None, dres) + (None,) * 8


def norm_cbam_cl(x, gamma, beta, w1, w2, wsp, eps=1e-5, mode=0, res=None, act=ACT_NONE, slope=0.01, out=None):
    return _NormCbamClFn.apply(x, gamma, beta, res, w1, w2, wsp, eps, mode, act, slope, out)


class _MeanClFn(torch.autograd.Function):
    """whole-map average of a channels-last tensor -> fp32 [N, C] (graph/encoder.py:20,35)"""

    @staticmethod
    def forward(ctx, x):
        ct = _need_cl(x, "global_avg_pool (channels-last)")
        N, C, H, W = x.shape
        if ct != C:
            raise RuntimeError("global_avg_pool (channels-last): dense input expected")
        out = torch.empty((N, C), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_mean_nhwc_fwd(_p(x), _p(out), N, C, H * W, _store(x), _s()), "mean_nhwc_fwd")
        ctx.shape = (N, C, H, W)
        ctx.dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        N, C, H, W = ctx.shape
        dout = dout.contiguous()
        dx = new_channels_last(N, C, H, W, dout.device, ctx.dtype)
        nat.check(nat.lib().mgvae_mean_nhwc_bwd(_p(dout), _p(dx), N, C, H * W, _store(dx), _s()), "mean_nhwc_bwd")
        return dx


def global_avg_pool_cl(x):
    return _MeanClFn.apply(x)


class _InstNormClFn(torch.autograd.Function):
    """nn.InstanceNorm2d(affine) + fused (Leaky)ReLU on channels-last tensors (graph/decoder.py:81-83,124-126)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, slope, out):
        xct = _need_cl(x, "instance_norm (channels-last)")
        N, C, H, W = x.shape
        if xct != C:
            raise RuntimeError("instance_norm (channels-last): the normalised tensor must be dense")
        y = out if out is not None else new_channels_last(N, C, H, W, x.device, x.dtype)
        yct = _need_cl(y, "instance_norm output")
        _same_storage(x, y, "instance_norm (channels-last)")
        stats = torch.empty((nat.lib().mgvae_instance_norm_nhwc_stats_floats(N, C, H, W),), device=x.device, dtype=torch.float32)
        nat.check(nat.lib().mgvae_instance_norm_nhwc_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), N, C, H, W, yct, 0, eps, act,
                                                         slope, _store(x), _s()), "instance_norm_nhwc_fwd")
        ctx.save_for_backward(x, gamma, stats, y)
        ctx.beta = beta
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats, y = ctx.saved_tensors
        act, slope = ctx.cfg
        N, C, H, W = x.shape
        yct = cl_pitch(y)
        dy = _as_cl(dy, x)
        if cl_pitch(dy) != yct:
            y = y.contiguous(memory_format=CL); dy = dy.contiguous(memory_format=CL); yct = C
        dx = new_channels_last(N, C, H, W, x.device, x.dtype)
        scratch = torch.empty((2 * N * C,), device=x.device, dtype=torch.float32)
        dg = grad_slot(gamma) if gamma.requires_grad else None
        db = grad_slot(ctx.beta) if ctx.beta.requires_grad else None
        nat.check(nat.lib().mgvae_instance_norm_nhwc_bwd(_p(x), _p(gamma), _p(stats), _p(y), _p(dy), _p(dx), _p(dg), _p(db),
                                                         _p(scratch), N, C, H, W, yct, 0, act, slope, _store(x), _s()), "instance_norm_nhwc_bwd")
        return dx, None, None, None, None, None, None


def instance_norm_cl(x, gamma, beta, eps=1e-5, act=ACT_NONE, slope=0.01, out=None):
    return _InstNormClFn.apply(x, gamma, beta, eps, act, slope, out)


class _ConvTClFn(torch.autograd.Function):
    """nn.ConvTranspose2d on channels-last tensors (graph/decoder.py:73-77,116-120): forward is the stride-phase
    data-gradient kernel of csrc/conv_nhwc.inc, d/dx the forward-conv kernel; the weight is stored
    [Cin, KH, KW, Cout] (torch.channels_last of the [Cin, Cout, KH, KW] parameter)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, opad, act, slope, out):
        xct = _need_cl(x, "conv_transpose2d (channels-last)")
        _cl_weight(w, "conv_transpose2d (channels-last)")
        N, Ci, h, wd = x.shape
        _, Co, KH, KW = w.shape
        OH = (h - 1) * stride[0] - 2 * pad[0] + KH + opad[0]
        OW = (wd - 1) * stride[1] - 2 * pad[1] + KW + opad[1]
        y = out if out is not None else new_channels_last(N, Co, OH, OW, x.device, x.dtype)
        yct = _need_cl(y, "conv_transpose2d output")
        _same_storage(x, y, "conv_transpose2d (channels-last)")
        k = (KH, KW)
        # conv geometry: X = y (image side, Cx = Co), Y = x (feature side, Cy = Ci)
        d = _desc(N, Co, OH, OW, Ci, h, wd, k, stride, pad, yct, xct, act, slope)
        if x.dtype == torch.bfloat16:
            _, wt = _bf16_weights(w)
            wst, wsp, wsn = _split_ws("bf16", d, 1, x.device)
            nat.check(nat.lib().mgvae_conv2d_nhwc_bf16_bwd_data(ctypes.byref(d), _p(x), _p(wt), _p(b), _p(y), None, wsp, wsn, _s()), "convT_nhwc_bf16_fwd")
        elif _x3_ok(Co, Ci):
            _, wt3 = _x3_weights(w)
            wst, wsp, wsn = _split_ws("x3", d, 1, x.device)
            nat.check(nat.lib().mgvae_conv2d_nhwc_x3_bwd_data(ctypes.byref(d), _p(x), _p(wt3), _p(b), _p(y), None, wsp, wsn, _s()), "convT_nhwc_x3_fwd")
        else:
            nat.check(nat.lib().mgvae_conv2d_nhwc_bwd_data(ctypes.byref(d), _p(x), _p(w), _p(b), _p(y), None, _s()), "convT_nhwc_fwd")
        ctx.geom = (N, Co, OH, OW, Ci, h, wd, k, stride, pad, xct, act, slope)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.b = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N, Co, OH, OW, Ci, h, wd, k, s, p, xct, act, slope = ctx.geom
        L = nat.lib()
        bf = x.dtype == torch.bfloat16
        if _trunk_streams or _used_sides:
            _ensure_join_callback()
        dy = _as_cl(dy, x)
        if act != ACT_NONE:
            dy = _act_bwd_cl(y, dy, act, slope)
        dct = _need_cl(dy, "conv_transpose2d (channels-last) backward")
        b = ctx.b

        def weight_grads():
            if w.requires_grad:
                d = _desc(N, Co, OH, OW, Ci, h, wd, k, s, p, dct, xct, ACT_NONE, 0.0)
                fn = L.mgvae_conv2d_nhwc_bf16_bwd_weight if bf else (L.mgvae_conv2d_nhwc_x3_bwd_weight if _x3_ok(Co, Ci) else L.mgvae_conv2d_nhwc_bwd_weight)
                nat.check(fn(ctypes.byref(d), _p(dy), _p(x), _p(grad_slot(w)), _s()), "convT_nhwc_bwd_weight")
            if b is not None and b.requires_grad:
                nat.check(L.mgvae_channel_sum_nhwc_accum(_p(dy), N * OH * OW, Co, dct, 0, _p(grad_slot(b)), _store(dy), _s()), "bias_grad_nhwc")

        if FORK_WGRAD and N >= FORK_MIN_BATCH and ctx.needs_input_grad[0] and (w.requires_grad or (b is not None and b.requires_grad)):
            _wgrad_rr[0] += 1
            with _forked(x, dy, slot=2 + _wgrad_rr[0] % WGRAD_STREAMS):
                weight_grads()
        else:
            weight_grads()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_channels_last(N, Ci, h, wd, dy.device, x.dtype)
            d2 = _desc(N, Co, OH, OW, Ci, h, wd, k, s, p, dct, Ci, ACT_NONE, 0.0)
            if bf:
                wk, _ = _bf16_weights(w)
                wst, wsp, wsn = _split_ws("bf16", d2, 0, dy.device)
                nat.check(L.mgvae_conv2d_nhwc_bf16_fwd(ctypes.byref(d2), _p(dy), _p(wk), None, _p(dx), None, wsp, wsn, _s()), "convT_nhwc_bf16_bwd_data")
            elif _x3_ok(Co, Ci):
                wk3, _ = _x3_weights(w)
                wst, wsp, wsn = _split_ws("x3", d2, 0, dy.device)
                nat.check(L.mgvae_conv2d_nhwc_x3_fwd(ctypes.byref(d2), _p(dy), _p(wk3), None, _p(dx), None, wsp, wsn, _s()), "convT_nhwc_x3_bwd_data")
            else:
                nat.check(L.mgvae_conv2d_nhwc_fwd(ctypes.byref(d2), _p(dy), _p(w), None, _p(dx), None, _s()), "convT_nhwc_bwd_data")
        return dx, None, None, None, None, None, None, None, None


def conv_transpose2d_cl(x, w, b=None, stride=(1, 1), pad=(0, 0), opad=(0, 0), act=ACT_NONE, slope=0.01, out=None):
    return _ConvTClFn.apply(x, w, b, stride, pad, opad, act, slope, out)
