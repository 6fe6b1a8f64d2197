"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend
"nccl") over xGMI.  Replaces both data-parallel strategies of the reference:
nn.DataParallel (agent/barGen2.py:89-96: re-broadcasts 382 MB of weights on every
forward, reduces to GPU 0) and the Horovod agent (agent/barGen_horovod.py:91-99,130-134:
per-tensor all-reduce + initial broadcast).

The flat gradient buffer (hipops.flat.FlatParams) is all-reduced (sum) in a few large
contiguous buckets; buckets are launched ASYNCHRONOUSLY as soon as the backward kernels
that write them have been enqueued (decoder slice first, while the encoders' backward
still runs), RCCL runs them on its own stream, and the fused Adam launch waits on them.
The 1/world_size averaging is folded into the Adam kernel (grad_scale)."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_dist() else 1


def broadcast_flat(flat, src=0):
    """one-time broadcast of the initial weights (reference: hvd.broadcast_parameters)"""
    if is_dist():
        dist.broadcast(flat, src)


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


class GradReducer:
    """asynchronous bucketed all-reduce(sum) over slices of one flat gradient tensor"""

    def __init__(self, flat_grad, bucket_elems=16 * 1024 * 1024):
        self.g = flat_grad
        self.bucket_elems = bucket_elems
        self.pending = []
        self.done_ranges = []

    def reduce_range(self, start, end):
        """enqueue all-reduces for [start, end); returns immediately"""
        if not is_dist() or end <= start:
            return
        if self.g.is_cuda:      # weight gradients may still be in flight on the side streams (functional._forked)
            from . import functional as HF
            HF.join_side_streams()
        for s, e in split_buckets(start, end, self.bucket_elems):
            self.pending.append(dist.all_reduce(self.g[s:e], op=dist.ReduceOp.SUM, async_op=True))
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
