"""Flat parameter / gradient / Adam-state buffers.

All parameters of a network are re-pointed into ONE contiguous fp32 device buffer (and
their ``.grad`` into a second one), so that per step
  * zero_grad is one memset,
  * the Adam update (torch.optim.Adam defaults, reference: agent/barGen2.py:60-64) is one
    HIP launch over 7 HBM streams (hipops: mgvae_adam_step),
  * the data-parallel gradient exchange is a handful of large RCCL all-reduces over
    contiguous slices (hipops.dist).
Each tensor starts on a 256-byte boundary.  Parameters that never receive a gradient
(reference defect D5: decoder.layers.{0,1}.bn1.*) simply keep g == 0, for which the
Adam update is exactly zero -- same result as torch skipping ``grad is None``.
"""
import ctypes
import math

import torch

from . import _native as nat

_ALIGN = 64  # floats
_HYPER_RING = 257   # pinned rows for the per-step scalars (row 0 + 256 steps of host run-ahead)


class FlatParams:
    def __init__(self, params, lr=0.002, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatParams: parameters must live on a ROCm device (no CPU fallback)")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                # the parameter keeps its own (dense) memory order inside the flat buffer: a channels-last conv weight
                # stays stored [Cy, KH, KW, Cx] -- Adam and the gradient exchange are elementwise, only the kernels care
                v = self._view(self.flat, p, o)
                v.copy_(p.detach())
                p.data = v
                g = self._view(self.grad, p, o)
                p._mg_grad = g
                p.grad = g
                p._mg_owner = self
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        # counts every update of the flat weights by ANY Adam state over them (second_state shares it): derived copies of
        # the weights (the bf16 operands of the channels-last convs) are refreshed when it moves
        self.weights_version = [0]
        # step-dependent scalars travel through a RING of pinned rows: the host runs several steps ahead of the GPU, and
        # an async copy of step n must not find the scalars of step n+1 in its source
        self._hyper_ring = torch.zeros(_HYPER_RING, 4, dtype=torch.float32).pin_memory()
        self._hyper_host = self._hyper_ring[0]      # the row a captured graph re-reads (GraphedPretrainStep)
        self._hyper = torch.zeros(4, device=dev, dtype=torch.float32)
        self.param_groups = [{"lr": lr, "params": self.params}]   # ReduceLROnPlateau-compatible surface

    @staticmethod
    def _view(buf, p, o):
        """the slice of ``buf`` that parameter ``p`` (offset ``o``) occupies, with p's shape AND memory order"""
        st = p.stride()
        dense = sorted(zip(st, p.shape), reverse=True)
        span, ok = 1, True
        for s_, n_ in reversed(dense):
            if n_ > 1 and s_ != span:
                ok = False
            span *= n_
        if not ok:          # not a dense permutation (never the case for module parameters): fall back to row-major
            return buf[o:o + p.numel()].view(p.shape)
        return torch.as_strided(buf, p.shape, st, o)

    # ---- torch.optim.Optimizer-like surface used by the agents -------------------------
    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:
            if p.grad is None:
                p.grad = p._mg_grad

    def begin_step(self):
        """count the step and upload its scalars (lr / bias corrections) on the current stream"""
        self.step_count += 1
        self.weights_version[0] += 1
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        h = self._hyper_ring[1 + self.step_count % (_HYPER_RING - 1)]      # row 0 belongs to the captured graph
        h[0], h[1], h[2], h[3] = lr / bc1, math.sqrt(bc2), b1, b2
        self._hyper.copy_(h, non_blocking=True)

    def invalidate_weight_copies(self):
        """the flat weights were changed behind the kernels' back (a restore, a graph replay whose captured repack ran
        BEFORE its Adam node): the matrix-pipe copies of the conv weights (hipops.functional._WeightCopies) are stale for
        the next eager launch.  Also forgets an event recorded inside a stream capture."""
        self.weights_version[0] += 1
        for g in self.__dict__.get("_mg_weight_copies", {}).values():
            g.event, g.waited = None, set()

    def step_range(self, start, end, grad_scale=1.0):
        """Adam over the flat slice [start, end) on the current stream (after begin_step on a stream it is ordered behind)"""
        if end <= start:
            return
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        vp = lambda t, o: ctypes.c_void_p(t.data_ptr() + 4 * o)
        nat.check(nat.lib().mgvae_adam_step(vp(self.flat, start), vp(self.grad, start), vp(self.exp_avg, start),
                                            vp(self.exp_avg_sq, start), end - start, ctypes.c_void_p(self._hyper.data_ptr()),
                                            self.eps, grad_scale, s), "adam_step")

    def step(self, grad_scale=1.0):
        self.begin_step()
        self.step_range(0, self.numel, grad_scale)

    def state_dict(self):
        """torch.optim.Adam's state_dict layout (per-parameter step / exp_avg / exp_avg_sq), so the
        reference's checkpoints (agent/barGen2.py:168-177) and this build's are interchangeable"""
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self._view(self.exp_avg, p, o).clone(),
                            "exp_avg_sq": self._view(self.exp_avg_sq, p, o).clone()}
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = float(g["lr"])
        self.betas, self.eps = tuple(g.get("betas", self.betas)), float(g.get("eps", self.eps))
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        steps = []
        for i, st in sd["state"].items():
            i = int(i)
            if i >= len(self.params):
                continue                      # e.g. refiner parameters of a reference checkpoint
            p, o = self.params[i], self.offsets[i]
            n = p.numel()
            self._view(self.exp_avg, p, o).copy_(st["exp_avg"])
            self._view(self.exp_avg_sq, p, o).copy_(st["exp_avg_sq"])
            steps.append(int(float(st["step"])))
        self.step_count = max(steps) if steps else 0

    def second_state(self, lr=None):
        """another Adam over the SAME parameters and gradients with its own moments and step
        count (the reference's agent/barGen.py:60-62 builds opt_gen1 / opt_gen2 this way)"""
        o = object.__new__(FlatParams)
        o.params, o.offsets, o.numel = self.params, self.offsets, self.numel
        o.flat, o.grad = self.flat, self.grad
        o.exp_avg, o.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        o.lr, o.betas, o.eps = (lr if lr is not None else self.lr), self.betas, self.eps
        o.step_count = 0
        o.weights_version = self.weights_version
        o._hyper_ring = torch.zeros(_HYPER_RING, 4, dtype=torch.float32).pin_memory()
        o._hyper_host = o._hyper_ring[0]
        o._hyper = torch.zeros(4, device=self.flat.device, dtype=torch.float32)
        o.param_groups = [{"lr": o.lr, "params": o.params}]
        return o

    def buckets(self, nbuckets):
        """contiguous [start, end) slices of the flat gradient, in REVERSE parameter order
        (gradients of the last layers are ready first in backward)"""
        per = (self.numel + nbuckets - 1) // nbuckets
        per = (per + _ALIGN - 1) // _ALIGN * _ALIGN
        out, end = [], self.numel
        while end > 0:
            start = max(0, end - per)
            out.append((start, end))
            end = start
        return out
