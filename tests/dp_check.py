"""helper for tests/test_dp_gpu.py: run under torch.distributed.run with 2 ranks (gloo backend,
both ranks on the one GPU of the test box).  Every rank builds the same generator, takes its
shard r::world of one fixed global batch, runs ONE PretrainStep (hooks -> early decoder
all-reduce -> reduce_rest -> fused Adam with 1/world), then rank 0 repeats the step single-process
on the full batch from the same initial weights and compares gradients and updated weights."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "musicgeneration_vae-torch_amd"))


def build(dev, seed):
    from graph.model import Model
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.loss.bar_loss import Loss, DLoss
    from hipops.train import PretrainStep
    torch.manual_seed(seed)
    gen, zb, zp = Model().to(dev).eval(), BarZDiscriminator().to(dev), PhraseZDiscriminator().to(dev)
    with torch.no_grad():                      # tame the N(-1,1) init so gradients are well above the noise floor
        for m in (gen, zb, zp):
            for p in m.parameters():
                if p.dim() > 1:
                    p.mul_(0.05)
    for d in (zb, zp):
        for p in d.parameters():
            p.requires_grad = False
    return PretrainStep(gen, zb, zp, Loss().to(dev), DLoss(), lr=0.002)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    import __graft_entry__ as g
    g.build()
    from oracle.weights import make_inputs
    batch = make_inputs(2 * world, seed=99)
    shard = tuple(t[rank::world].contiguous().to(dev) for t in batch)
    step = build(dev, 1)
    w0 = step.opt.flat.clone()
    loss, _ = step(*shard)
    torch.cuda.synchronize()
    g_dp = step.opt.grad.clone() / world           # the buffer holds the all-reduced SUM
    w_dp = step.opt.flat.clone()
    ok = True
    # all ranks must hold identical weights after the step
    ws = [torch.zeros_like(w_dp) for _ in range(world)]
    dist.all_gather(ws, w_dp)
    same = all(torch.equal(ws[0], w) for w in ws)
    if rank == 0:
        dist_was = dist.is_initialized()
        # single-process reference on the full batch: tear the group down so hipops.dist sees world 1
        pass
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        ref = build(dev, 1)
        assert torch.equal(ref.opt.flat, w0), "same seed must give same initial weights"
        full = tuple(t.to(dev) for t in batch)
        ref(*full)
        torch.cuda.synchronize()
        g_ref = ref.opt.grad
        num = (g_dp - g_ref).norm().item(); den = g_ref.norm().item()
        l2 = num / den
        frac = ((g_dp - g_ref).abs() > 1e-3 * g_ref.abs().max()).float().mean().item()
        dw = (w_dp - ref.opt.flat).abs().max().item()
        moved = (w_dp - w0).abs().max().item()
        print("DPCHECK same_across_ranks=%s grad_l2_rel=%.3e outlier_frac=%.3e max_dw=%.3e moved=%.3e" % (same, l2, frac, dw, moved))
        ok = same and l2 < 2e-2 and frac < 2e-2 and moved > 0
        print("DPCHECK", "PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
