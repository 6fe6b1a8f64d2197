"""One optimizer step of the bar VAE on the HIP path, shared by the agents and bench.py.

``PretrainStep`` is the generator step of agent/barGen2.py:267-292: generator forward
(phrase encoder + 2 x encoder + decoder), three frozen z-discriminator forwards with
"valid" targets (DLoss), Loss(gen, note, is_pretraining), backward, Adam -- with the
data-parallel gradient exchange overlapped with backward when torch.distributed is up."""
import torch

from . import dist as hdist
from .flat import FlatParams


def _stacked(a, b):
    """the [2B, D] tensor that ``a`` and ``b`` are the two halves of (graph.model.Model.encode_pair), else None"""
    base = a._base
    if (base is None or base is not b._base or a.shape != b.shape or base.numel() != 2 * a.numel()
            or a.data_ptr() != base.data_ptr() or b.data_ptr() != a.data_ptr() + a.numel() * a.element_size()):
        return None
    return base.view(2 * a.shape[0], -1)


import os as _os
SPLIT_DECODER = _os.environ.get("MGVAE_SPLIT_DECODER", "0") != "0"      # see PretrainStep._decode

class PretrainStep:
    def __init__(self, generator, z_disc_bar, z_disc_phrase, loss_gen, loss_d, lr=0.002, bucket_elems=16 * 1024 * 1024):
        self.gen, self.zb, self.zp = generator, z_disc_bar, z_disc_phrase
        self.loss_gen, self.loss_d = loss_gen, loss_d
        self.opt = FlatParams(list(generator.parameters()), lr=lr)
        hdist.broadcast_flat(self.opt.flat)
        self.reducer = GradReducer(self.opt.grad, bucket_elems)
        self._steps_checked = 0
        # flat range of the decoder (the bulk of the gradient, finished first in backward)
        ids = {id(p): i for i, p in enumerate(self.opt.params)}
        dec = [ids[id(p)] for p in generator.decoder.parameters()]
        self.dec_range = (self.opt.offsets[min(dec)],
                          self.opt.offsets[max(dec)] + self.opt.params[max(dec)].numel())
        self.dec_range = (self.dec_range[0], (self.dec_range[1] + 63) // 64 * 64)
        # flat range of the bar encoder's trunk (layers + linear; its two stems finish last and go with the rest)
        enc = getattr(generator, "encoder", None)
        self.enc_range = None
        if enc is not None and hasattr(enc, "layers") and hasattr(enc, "linear"):
            idx = [ids[id(p)] for m in (enc.layers, enc.linear) for p in m.parameters()]
            if idx and max(idx) - min(idx) + 1 == len(idx):          # contiguous in the flat buffer
                self.enc_range = (self.opt.offsets[min(idx)],
                                  (self.opt.offsets[max(idx)] + self.opt.params[max(idx)].numel() + 63) // 64 * 64)

    def _arm_overlap(self, tensors):
        """when the gradients of all decoder INPUTS have been produced, every decoder
        parameter gradient is already enqueued: start its all-reduce"""
        enc = getattr(self.gen, "encoder", None)
        t_in = getattr(enc, "trunk_input", None) if enc is not None else None
        if enc is not None:
            enc.trunk_input = None          # one step's tensor: do not keep its autograd graph alive between steps
        if not hdist.is_dist():
            return
        state = {"left": len(tensors)}

        def fire(_g):
            state["left"] -= 1
            if state["left"] == 0:
                self.reducer.reduce_range(*self.dec_range, early=True)

        for t in tensors:
            t.register_hook(fire)
        # second early bucket: the bar-encoder trunk, while the (twice as long) phrase trunk is still in backward.
        # Only when BOTH bar-encoder passes ran as the one stacked pass whose trunk input ``t_in`` is: with two
        # separate passes (variational encoder, unequal shapes) the attribute holds the last pass only, and the other
        # pass would still be accumulating into the range while it is being reduced.
        z, pre_z = tensors[0], tensors[1]
        if self.enc_range is not None and t_in is not None and _stacked(z, pre_z) is not None:
            t_in.register_hook(lambda _g: self.reducer.reduce_range(*self.enc_range, early=True))

    def forward_loss(self, note, pre_note, phrase, position, is_pretraining=True):
        """generator forward + the four loss terms.  The three frozen z-discriminator passes depend only on the
        encoders' outputs, like the decoder: they run on a side stream beside it (and so do their backward passes)."""
        from graph.loss.bar_loss import DLoss
        from . import functional as HF
        gen_m = self.gen

        def z_losses(z, pre_z, pf):
            loss = DLoss.constant(self.zp(pf).view(-1), 1.0)
            zz = _stacked(z, pre_z)
            if zz is not None:      # one discriminator pass over both latents: mean over 2B, twice = the two means over B
                return loss + 2.0 * DLoss.constant(self.zb(zz).view(-1), 1.0)
            return loss + DLoss.constant(self.zb(z).view(-1), 1.0) + DLoss.constant(self.zb(pre_z).view(-1), 1.0)

        if type(gen_m).__module__ == "graph.model" and not getattr(gen_m, "use_refiner", False):
            pf = gen_m.encode_phrase(phrase)
            z, pre_z = gen_m.encode_pair(note, pre_note)
            gen_m.join_phrase()
            with HF.forked_branch(z, pre_z, pf, slot=1):      # slot 1: the decoder forks its own branches on slot 0
                loss_z = z_losses(z, pre_z, pf)
            gen = self._decode(gen_m, z, pre_z, pf, position)
            loss_g = self.loss_gen(gen, note, is_pretraining)
            HF.join_side_streams(slot=1)
            loss = loss_z + loss_g
        else:
            gen, z, pre_z, pf = gen_m(note, pre_note, phrase, position)
            loss = z_losses(z, pre_z, pf) + self.loss_gen(gen, note, is_pretraining)
        return loss, gen, (z, pre_z, pf)

    @staticmethod
    def _decode(gen_m, z, pre_z, pf, position):
        """the decoder; with MGVAE_SPLIT_DECODER=1 and >= 32 bars as two half-batches, the second on the phrase trunk's stream
        (idle once the trunks have joined: no new stream, the hardware queues stay as they are -- a fifth busy stream costs 60 %,
        measured).  The decoder has no cross-sample op, so the halves are exact; a launch's fill and drain (~35 us of its
        ~150) then overlap the other half's steady state instead of leaving the chip half empty."""
        side = getattr(gen_m, "_side", None)
        b = z.shape[0]
        if not SPLIT_DECODER or side is None or b < 32 or b % 2 or gen_m.decoder._drop_masks is not None:
            return gen_m.decoder(z, pre_z, pf, position)
        h = b // 2
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gb = gen_m.decoder(z[h:], pre_z[h:], pf[h:], position[h:])
        ga = gen_m.decoder(z[:h], pre_z[:h], pf[:h], position[:h])
        cur.wait_stream(side)
        gb.record_stream(cur)
        return torch.cat([ga, gb], 0)

    def __call__(self, note, pre_note, phrase, position, is_pretraining=True):
        if self.opt.grad.is_cuda:
            self.reducer.main_stream = torch.cuda.current_stream()
        self.opt.zero_grad()
        loss, gen, latents = self.forward_loss(note, pre_note, phrase, position, is_pretraining)
        self._arm_overlap(latents)
        loss.backward()
        self.reducer.reduce_rest()
        self.reducer.wait()
        self.opt.step(grad_scale=1.0 / hdist.world_size())
        if hdist.is_dist() and self._steps_checked < 2:
            # start-up check: main + phrase trunk + ONE weight-gradient stream (+ the bf16 transport's communication
            # stream when that is on) beside RCCL's own -- never a fifth busy stream
            from . import functional as HF
            self._steps_checked += 1
            n = len(HF.live_streams())
            if n > 2:
                raise RuntimeError("data-parallel step created %d side streams (expected <= 2: phrase trunk + one "
                                   "weight-gradient stream)" % n)
        return loss, gen


class GraphedPretrainStep:
    """``PretrainStep`` with zero_grad + forward + losses + backward + Adam captured ONCE as a HIP graph and replayed.

    At 16-32 bars per GPU (BASELINE.json configs 3-4) the ~900 launches of a step cost the host more time than the
    GPU needs to execute them; a replay is one launch.  What varies between iterations lives outside the capture:
      * the batch: copied into static input buffers;
      * dropout: the decoder's two masks are static tensors refreshed from the Philox stream before each replay
        (a kernel's seed / offset arguments would be frozen by the capture);
      * Adam's step-dependent scalars (lr / bias corrections): a pinned host vector that the captured H2D copy re-reads
        at every replay (hipops.flat.FlatParams.step).
    Single-process only: with torch.distributed up, use PretrainStep (its bucketed all-reduce overlaps backward from
    autograd hooks, which a replay does not run)."""

    def __init__(self, step, note, pre_note, phrase, position, is_pretraining=True, warmup=3):
        import torch.cuda
        from . import functional as HF
        if hdist.is_dist():
            raise RuntimeError("GraphedPretrainStep is single-process; use PretrainStep under torch.distributed")
        self.step_obj, self.HF = step, HF
        self.is_pretraining = is_pretraining
        self._replayed = None
        self.inputs = [t.clone() for t in (note, pre_note, phrase, position)]
        dec = step.gen.decoder
        self.dec = dec
        B = note.shape[0]
        self.masks = [torch.ones(B, 1152, device=note.device), torch.ones(B, 1152, device=note.device)]
        self._ones = torch.ones(B, 1152, device=note.device)
        self.use_masks = dec.training and dec.dropout_p > 0.0
        # the eager warm-up (conv gather tables, autotuner, allocator) runs real optimizer steps: the parameters, both
        # Adam moments and the step counter are restored afterwards, so constructing the object trains nothing
        opt = step.opt
        saved = (opt.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up on a side stream, as torch's capture recipe asks
            for _ in range(warmup):
                self._eager()
                # pinned row 0 of the Adam scalars is rewritten by the next warm-up step: its H2D copy must have read it
                torch.cuda.current_stream().synchronize()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            opt.flat.copy_(saved[0]); opt.exp_avg.copy_(saved[1]); opt.exp_avg_sq.copy_(saved[2])
        opt.step_count = saved[3]
        opt.invalidate_weight_copies()       # the copies belong to the warm-up's weights, not to the restored ones
        del saved
        self.graph = torch.cuda.CUDAGraph()
        self._refresh_masks()
        self._set_hyper()                              # capture records, it does not execute: no step is consumed
        with torch.cuda.graph(self.graph):
            self.loss, self.gen_out = self._body()
        torch.cuda.synchronize()
        # capture executed nothing: the copies still hold the warm-up's weights, and the event the repack "recorded" belongs
        # to the capture (an eager launch must not wait on it)
        opt.invalidate_weight_copies()

    def _refresh_masks(self):
        if self.use_masks:
            for m in self.masks:
                m.copy_(self.HF.dropout(self._ones, self.dec.dropout_p, True))
            self.dec._drop_masks = self.masks
        else:
            self.dec._drop_masks = None

    def _set_hyper(self):
        import math
        opt = self.step_obj.opt
        b1, b2 = opt.betas
        h = opt._hyper_host
        t = max(1, opt.step_count)         # (0 only while the graph is being captured: nothing executes then)
        h[0] = opt.param_groups[0]["lr"] / (1.0 - b1 ** t)
        h[1] = math.sqrt(1.0 - b2 ** t)
        h[2], h[3] = b1, b2

    def _body(self):
        """the captured program (also what the eager warm-up runs)"""
        import ctypes
        from graph.loss.bar_loss import DLoss
        from . import _native as nat
        st, opt = self.step_obj, self.step_obj.opt
        note, pre_note, phrase, position = self.inputs
        opt.weights_version[0] += 1          # the matrix-pipe copies of the conv weights are re-made once per step (captured too)
        opt.zero_grad()
        loss, gen, _ = st.forward_loss(note, pre_note, phrase, position, self.is_pretraining)
        loss.backward()
        opt._hyper.copy_(opt._hyper_host, non_blocking=True)       # re-read from the pinned vector at every replay
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        nat.check(nat.lib().mgvae_adam_step(vp(opt.flat), vp(opt.grad), vp(opt.exp_avg), vp(opt.exp_avg_sq), opt.numel,
                                            vp(opt._hyper), opt.eps, 1.0, s), "adam_step")
        return loss.detach(), gen.detach()

    def _eager(self):
        self._refresh_masks()
        self.step_obj.opt.step_count += 1
        self._set_hyper()
        return self._body()

    def __call__(self, note, pre_note, phrase, position):
        # the replayed copy node re-reads ONE pinned row: the previous replay must have read it before it is rewritten
        if self._replayed is not None:
            self._replayed.synchronize()
        for dst, src in zip(self.inputs, (note, pre_note, phrase, position)):
            dst.copy_(src, non_blocking=True)
        self._refresh_masks()
        self.step_obj.opt.step_count += 1
        self._set_hyper()
        self.graph.replay()
        self._replayed = torch.cuda.Event()
        self._replayed.record()
        # the replayed repack ran in FRONT of the replayed Adam: the copies hold the weights of the step before.  The next
        # replay repacks by itself; an eager forward (validation, sampling) has to as well
        self.step_obj.opt.invalidate_weight_copies()
        return self.loss, self.gen_out


from .dist import GradReducer  # noqa: E402
