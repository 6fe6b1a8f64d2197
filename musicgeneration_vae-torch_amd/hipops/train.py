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


class PretrainStep:
    def __init__(self, generator, z_disc_bar, z_disc_phrase, loss_gen, loss_d, lr=0.002, bucket_elems=16 * 1024 * 1024):
        self.gen, self.zb, self.zp = generator, z_disc_bar, z_disc_phrase
        self.loss_gen, self.loss_d = loss_gen, loss_d
        self.opt = FlatParams(list(generator.parameters()), lr=lr)
        hdist.broadcast_flat(self.opt.flat)
        self.reducer = GradReducer(self.opt.grad, bucket_elems)
        # flat range of the decoder (the bulk of the gradient, finished first in backward)
        ids = {id(p): i for i, p in enumerate(self.opt.params)}
        dec = [ids[id(p)] for p in generator.decoder.parameters()]
        self.dec_range = (self.opt.offsets[min(dec)],
                          self.opt.offsets[max(dec)] + self.opt.params[max(dec)].numel())
        self.dec_range = (self.dec_range[0], (self.dec_range[1] + 63) // 64 * 64)

    def _arm_overlap(self, tensors):
        """when the gradients of all decoder INPUTS have been produced, every decoder
        parameter gradient is already enqueued: start its all-reduce"""
        if not hdist.is_dist():
            return
        state = {"left": len(tensors)}

        def fire(_g):
            state["left"] -= 1
            if state["left"] == 0:
                self.reducer.reduce_range(*self.dec_range)

        for t in tensors:
            t.register_hook(fire)

    def __call__(self, note, pre_note, phrase, position, is_pretraining=True):
        from graph.loss.bar_loss import DLoss
        self.opt.zero_grad()
        gen, z, pre_z, pf = self.gen(note, pre_note, phrase, position)
        self._arm_overlap((z, pre_z, pf))
        loss = DLoss.constant(self.zp(pf).view(-1), 1.0)
        zz = _stacked(z, pre_z)
        if zz is not None:      # one discriminator pass over both latents: mean over 2B, twice = the two means over B
            loss = loss + 2.0 * DLoss.constant(self.zb(zz).view(-1), 1.0)
        else:
            loss = loss + DLoss.constant(self.zb(z).view(-1), 1.0) + DLoss.constant(self.zb(pre_z).view(-1), 1.0)
        loss = loss + self.loss_gen(gen, note, is_pretraining)
        loss.backward()
        self.reducer.reduce_rest()
        self.reducer.wait()
        self.opt.step(grad_scale=1.0 / hdist.world_size())
        return loss, gen


from .dist import GradReducer  # noqa: E402
