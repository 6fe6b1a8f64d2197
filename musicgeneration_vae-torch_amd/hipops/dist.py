"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend
"nccl") over xGMI.  Replaces both data-parallel strategies of the reference:
nn.DataParallel (agent/barGen2.py:89-96: re-broadcasts 382 MB of weights on every
forward, reduces to GPU 0) and the Horovod agent (agent/barGen_horovod.py:91-99,130-134:
per-tensor all-reduce + initial broadcast).

The flat gradient buffer (hipops.flat.FlatParams) is all-reduced (sum) in a few large
contiguous buckets; buckets are launched ASYNCHRONOUSLY as soon as the backward kernels
that write them have been enqueued (decoder slice first, while the encoders' backward
still runs), RCCL runs them on its own stream, and the fused Adam launch waits on them.
The 1/world_size averaging is folded into the Adam kernel (grad_scale).

Two transports per bucket:
  * "f32" (default): one ``all_reduce(sum)`` of the fp32 slice;
  * "bf16" (``MGVAE_GRAD_TRANSPORT=bf16`` / ``GradReducer(transport="bf16")``): the bucket travels as bf16 and is summed
    in fp32 where it ARRIVES -- a direct reduce-scatter + all-gather: every rank rounds its slice to bf16, an
    all-to-all hands rank r the r-th shard of every peer (xGMI is point-to-point: all 7 peer links carry 1/8 of the
    bucket each, instead of a ring bound by one link), the shard is summed in fp32, rounded once, and an all-gather
    returns the reduced shards.  Half the bytes of the fp32 ring, one bf16 rounding of each addend and one of each
    sum (a bf16 all-reduce would re-round the running sum at every hop).  Intent of the reference's Horovod
    averaging: agent/barGen_horovod.py:91-99.
"""
import contextlib
import os
import random

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_dist() else 1


def backend_name():
    """'rccl' for torch's "nccl" backend on ROCm, the backend's own name otherwise, 'none' single-process"""
    if not is_dist():
        return "none"
    b = dist.get_backend()
    return "rccl" if b == "nccl" else str(b)


def broadcast_flat(flat, src=0):
    """one-time broadcast of the initial weights (reference: hvd.broadcast_parameters)"""
    if is_dist():
        dist.broadcast(flat, src)


def shared_seed(seed=None, lo=1, hi=10000):
    """the run's seed, IDENTICAL on every rank: a configured seed is returned as is; otherwise rank 0 draws
    ``random.randint(lo, hi)`` (the reference's agent/barGen2.py:53) and broadcasts it.  Every rank-shared decision of
    the agents (python ``random``: agent/barGen.py's per-epoch ``div_flag``, torch's host RNG) is seeded from it, so
    all ranks run the same schedule and issue the same collectives; only the Philox stream of the HIP dropout / prior
    kernels is offset per rank (functional.manual_seed(seed, rank))."""
    if seed is not None:
        return int(seed)
    box = [random.randint(lo, hi)]
    if is_dist():
        dist.broadcast_object_list(box, src=0)
    return int(box[0])


def split_buckets(start, end, max_elems, align=64):
    """contiguous [s, e) buckets covering [start, end), emitted from the END backwards
    (the tail of a slice is written first during backward)"""
    out, e = [], end
    max_elems = max(align, max_elems // align * align)
    while e > start:
        s = max(start, e - max_elems)
        out.append((s, e))
        e = s
    return out


def _to_bf16(src, dst):
    """dst (bf16) <- round-to-nearest-even(src fp32): HIP kernel on the device; torch on the host tensors that only
    the gloo tests of the exchange logic use"""
    if src.is_cuda:
        import ctypes
        from . import _native as nat
        nat.check(nat.lib().mgvae_f32_to_bf16(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel(),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "f32_to_bf16")
    else:
        dst.copy_(src)


def _sum_bf16_rows(rows, dst):
    """dst (bf16, [n]) <- round(sum over r of rows[r] in fp32), rows bf16 [world, n]"""
    if rows.is_cuda:
        import ctypes
        from . import _native as nat
        nat.check(nat.lib().mgvae_bf16_rows_sum(ctypes.c_void_p(rows.data_ptr()), ctypes.c_void_p(dst.data_ptr()), rows.shape[0],
                                                rows.shape[1], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "bf16_rows_sum")
    else:
        dst.copy_(rows.float().sum(0))


def _from_bf16(src, dst):
    """dst (fp32) <- src (bf16), exact"""
    if src.is_cuda:
        import ctypes
        from . import _native as nat
        nat.check(nat.lib().mgvae_bf16_to_f32(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel(),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "bf16_to_f32")
    else:
        dst.copy_(src)


class _Bf16Work:
    """handle of one bf16 bucket: its kernels and collectives were issued in order on the reducer's communication
    stream; ``wait`` makes the caller's stream wait for that stream and then unpacks the reduced bucket into the fp32
    gradient"""

    def __init__(self, reducer, s, e, packed, comm):
        self.r, self.s, self.e, self.packed, self.comm = reducer, s, e, packed, comm

    def wait(self):
        if self.comm is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.comm)
            self.packed.record_stream(cur)
        _from_bf16(self.packed[: self.e - self.s], self.r.g[self.s:self.e])


class _Entered:
    """an already entered context manager (GradReducer._ordered)"""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return self.ctx.__exit__(*a)


class GradReducer:
    """asynchronous bucketed all-reduce(sum) over slices of one flat gradient tensor"""

    def __init__(self, flat_grad, bucket_elems=16 * 1024 * 1024, transport=None):
        self.g = flat_grad
        self.bucket_elems = bucket_elems
        self.transport = (transport or os.environ.get("MGVAE_GRAD_TRANSPORT", "f32")).lower()
        if self.transport not in ("f32", "bf16"):
            raise ValueError("gradient transport must be 'f32' or 'bf16', got %r" % self.transport)
        self.pending = []
        self.done_ranges = []
        # the stream the step's backward runs on (set by the training step); an early bucket may be launched from an
        # autograd hook that fires on a SIDE stream (the phrase trunk's): the bucket then has to wait for this one too
        self.main_stream = None
        self._comm = None       # side stream of the bf16 transport's pack / sum kernels

    # ------------------------------------------------------------------ ordering
    def _ordered(self, early):
        """context in which a bucket is launched.  The collective is enqueued behind the CURRENT stream (torch's
        process group makes its communication stream wait for it), so the current stream must be ordered behind
        everything that wrote the bucket:
          * end of the step (``early`` False, called from the step's main stream after backward): every side and trunk
            stream is joined into the current stream;
          * an early bucket, launched from an autograd hook: the hook may fire on a SIDE stream (the phrase trunk's --
            autograd runs a node's hooks on the node's forward stream) although the bucket's gradients were written
            on the step's main stream and the weight-gradient stream forked from it.  The launch therefore switches to
            the main stream and joins only ITS forked side streams -- the phrase trunk, still in backward on its own
            stream, is not held up and does not write into an early bucket."""
        if not self.g.is_cuda:
            return contextlib.nullcontext()
        from . import functional as HF
        if early and self.main_stream is not None:
            ctx = torch.cuda.stream(self.main_stream)
            ctx.__enter__()
            try:
                HF.join_side_streams(parent=self.main_stream)
            except Exception:
                ctx.__exit__(None, None, None)
                raise
            return _Entered(ctx)
        HF.join_side_streams()
        return contextlib.nullcontext()

    # ------------------------------------------------------------------ transports
    def _launch(self, s, e):
        if self.transport == "f32":
            return dist.all_reduce(self.g[s:e], op=dist.ReduceOp.SUM, async_op=True)
        # the pack / all-to-all / fp32 sum / all-gather chain runs on a communication stream of its own, behind
        # everything the current stream has enqueued: the backward pass that launched this bucket continues beside it
        comm = None
        if self.g.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.g.device)
            comm = self._comm
            comm.wait_stream(torch.cuda.current_stream())
        world = dist.get_world_size()
        n = e - s
        shard = (n + world - 1) // world
        shard = (shard + 63) // 64 * 64
        with (torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()):
            # [world, shard] bf16: row r = the shard this rank sends to rank r (zero padded past the bucket's end)
            send = torch.zeros((world, shard), device=self.g.device, dtype=torch.bfloat16)
            _to_bf16(self.g[s:e], send.view(-1)[:n])
            recv = torch.empty_like(send)
            # pure data movement: reinterpret as bytes so that every backend moves it (gloo has no 16-bit all-to-all)
            dist.all_to_all_single(recv.view(torch.uint8).view(-1), send.view(torch.uint8).view(-1))
            mine = torch.empty((shard,), device=self.g.device, dtype=torch.bfloat16)
            _sum_bf16_rows(recv, mine)                   # fp32 accumulation of the world addends, one rounding
            out = torch.empty((world * shard,), device=self.g.device, dtype=torch.bfloat16)
            dist.all_gather_into_tensor(out.view(torch.uint8), mine.view(torch.uint8))
        return _Bf16Work(self, s, e, out, comm)

    def reduce_range(self, start, end, early=False):
        """enqueue all-reduces for [start, end); returns immediately.  ``early``: called from an autograd hook while
        backward is still running (see _ordered)"""
        if not is_dist() or end <= start:
            return
        with self._ordered(early):
            for s, e in split_buckets(start, end, self.bucket_elems):
                self.pending.append(self._launch(s, e))
        self.done_ranges.append((start, end))

    def reduce_rest(self):
        """all-reduce whatever part of the buffer has not been reduced yet this step"""
        n, cur = self.g.numel(), 0
        for s, e in sorted(self.done_ranges):
            if s > cur:
                self.reduce_range(cur, s)
            cur = max(cur, e)
        if cur < n:
            self.reduce_range(cur, n)

    def wait(self):
        """make the current stream wait for every pending bucket (no host sync on GPU)"""
        for w in self.pending:
            w.wait()
        self.pending, self.done_ranges = [], []


def all_reduce_mean_scalar(x):
    """average a python float / 0-dim tensor across ranks (the reference forgets to do this
    for the epoch loss that drives ReduceLROnPlateau: agent/barGen_horovod.py:329-334)"""
    if not is_dist():
        return float(x)
    t = torch.as_tensor(float(x), dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()
