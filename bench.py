"""Headline benchmark: bars/sec of one bar-VAE training step (forward + backward + Adam, incl.
the data-parallel gradient all-reduce) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the barGen2 pre-training generator step
(reference: agent/barGen2.py:267-292) -- PhraseEncoder + 2 x Encoder + Decoder forward, three
frozen z-discriminator forwards, Loss(is_pretraining=True), backward, Adam(lr 0.002) -- at
batch 64 per GPU, fp32, synthetic Bernoulli(0.05) piano rolls, weights_init (D4) weights.
The Refiner is excluded (it raises in the reference: defect D2).  Weak scaling: the per-GPU
batch stays 64 as N grows.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel family (the channels-last convs): algorithmic FLOPs / HIP-event time
                  measured live, per kernel variant, each variant against the peak of the matrix
                  instruction it runs on -- 2500 / 6 = 416.7 TFLOP/s fp32-equivalent for the default fp32
                  engine (six dense-bf16 MFMAs per fp32 product, csrc/conv_nhwc_x3.inc), 2500 for bf16
                  storage, 157.3 for the kernels on v_mfma_f32_32x32x2_f32; ``traffic`` / ``mfma_busy_pmc``
                  come from a committed PMC summary of THESE kernel sources, else null;
  cpu_baseline -- the CPU oracle (a restatement pinned bit-for-bit to the reference import)
                  running the same step on this host's cores (rank 0, N=1 only): batch 64 on all cores
                  (the GPU's workload), batch 4 on all cores and on 8 threads beside it.
``value`` is K steps / the barrier-bracketed wall time of the K steps (the driver's contract);
``ms_per_step_median`` / ``value_median_step`` give SURVEY 8d's median step from per-step events.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "musicgeneration_vae-torch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MATRIX_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MATRIX_PEAK_TFLOPS = 2500.0         # dense bf16 (the 5 PF headline figure includes 2:1 sparsity)
FLOP_PER_BAR_STEP = 29.5e9               # 3 x 9.83 GFLOP forward
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured streaming)


def synth_batch(batch, seed, device):
    rng = np.random.default_rng(seed)
    note = (rng.random((batch, 1, 96, 60)) < 0.05).astype(np.float32)
    pre = (rng.random((batch, 1, 96, 60)) < 0.05).astype(np.float32)
    phrase = (rng.random((batch, 1, 384, 60)) < 0.05).astype(np.float32)
    pos = rng.integers(0, 332, size=(batch,), dtype=np.int64)
    return tuple(torch.from_numpy(a).to(device) for a in (note, pre, phrase, pos))


def host_cores():
    """cores this process may really use (affinity / cgroup quota), capped at 16: the GPU
    box gives one GPU's job a 16-core share, and oversubscribing torch threads stalls it"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


T0 = time.perf_counter()


def _cpu_leg(batch, threads, min_seconds, max_steps, min_steps):
    """median step time of the oracle's training step (torch CPU fp32, autograd + Adam) after one warm-up step"""
    from oracle import restate as R
    from oracle import weights as W
    torch.set_num_threads(threads)
    gsd = {k: v.requires_grad_(True) for k, v in W.make_state_dict(W.manifest_generator(), 0, "d4").items()}
    zsd = W.make_state_dict(W.manifest_z_discriminator(), 0, "d4")
    params = [gsd[n] for n in gsd]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    note, pre, phrase, pos = W.make_inputs(batch, seed=1234)
    times = []
    it = 0
    while it == 0 or (sum(times) < min_seconds and len(times) < max_steps) or len(times) < min_steps:
        t0 = time.perf_counter()
        loss, _ = R.pretrain_step_loss(gsd, zsd, zsd, note, pre, phrase, pos, True)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
        R.adam_step(params, grads, m, v, it + 1)
        if it > 0:
            times.append(time.perf_counter() - t0)
        it += 1
    dt = float(np.median(times))
    return {"value": batch / dt, "unit": "bars/s", "cores": torch.get_num_threads(), "batch": batch,
            "sample": "%d timed steps = %.1f s of CPU work (after 1 warm-up) of the same pre-training step at batch %d, fp32, "
                      "torch %s CPU; median step" % (len(times), sum(times), batch, torch.__version__)}


def cpu_baseline():
    """the oracle's training step on this host's cores, a bounded sample: the headline leg runs the GPU's own
    workload (batch 64) on every core of the box's share; beside it the reference's operating scale (batch 4, BASELINE.json
    configs[0]) on all cores and on 8 threads (the survey container's thread count)."""
    n = host_cores()
    main = _cpu_leg(64, n, 15.0, 3, 2)
    main["kind"] = "port"
    main["batch4"] = _cpu_leg(4, n, 6.0, 30, 3)
    main["batch4_8_threads"] = _cpu_leg(4, min(8, n), 6.0, 30, 3)
    return main


def kernel_source_tag():
    """sha256 over the kernel sources (csrc/*.hip, *.inc, *.h and include/mgvae.h): identifies WHICH kernels a PMC summary
    was taken on (tools/pmc_summary.py stamps it into the summary; pmc_record only quotes a summary whose tag is this one)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.inc")) +
                   glob.glob(os.path.join(PKG, "csrc", "*.h"))) + [os.path.join(ROOT, "include", "mgvae.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_record(kernel):
    """HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) and the
    MFMA-busy share of ``kernel`` from a committed PMC summary (profiles/rN?_pmc_summary.json, written by
    tools/pmc_collect.sh + tools/pmc_summary.py: counters cannot be read from inside this process) -- but ONLY from a summary
    that was collected on these kernel sources (its ``_kernel_source_tag`` equals kernel_source_tag()).  Counters of another
    build say nothing about this run: the fields are then null and ``traffic_source`` says which summary was passed over."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    tag = kernel_source_tag()
    stale = None
    for f in reversed(files):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        if doc.get("_kernel_source_tag") != tag:
            stale = stale or os.path.basename(f)
            continue
        rec = doc.get(kernel)
        if rec:
            return dict(rec, source="profiles/" + os.path.basename(f))
    return {"source": "none for this build (newest summary, profiles/%s, is of other kernel sources)" % stale} if stale else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="bars per GPU")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 is BASELINE.json's metric config (configs[1]): fp32 storage and fp32-grade products; bf16 is "
                         "configs[2]/[3]'s: the channels-last island stores activations / gradients in bf16 and multiplies bf16 "
                         "weight copies (fp32 accumulation, statistics, weight gradients, master weights and Adam)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step (zero_grad, fwd, losses, bwd, Adam) as one captured HIP graph; single GPU "
                         "only.  Pays at small per-GPU batches (configs 3-4), where the host's launch rate bounds the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prof-detail", default="", help="write one CSV row per conv launch of the profiled step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP hot path has no CPU fallback")
    if os.environ.get("MGVAE_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())   # gloo test ranks may share the one GPU of a test box
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("LOCAL_RANK=%d but %d GPU(s) visible: one process per GPU" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MGVAE_DIST_BACKEND", "nccl")     # "nccl" IS RCCL on ROCm; gloo only for tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, "launch one process per GPU (WORLD_SIZE=%d, --gpus %d)" % (world, args.gpus)

    import __graft_entry__ as ge
    ge.build()
    from hipops import _native as nat
    from hipops import functional as HF
    from hipops.train import PretrainStep
    from hipops import dist as hdist
    from graph.model import Model
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.loss.bar_loss import Loss, DLoss
    HF.set_compute_dtype(args.dtype)
    peak = FP32_MATRIX_PEAK_TFLOPS if args.dtype == "f32" else BF16_MATRIX_PEAK_TFLOPS

    arch = ctypes.create_string_buffer(64)
    cus = ctypes.c_int(0)
    nat.check(nat.lib().mgvae_device_info(arch, 64, ctypes.byref(cus)), "device_info")

    torch.manual_seed(0)
    gen, zb, zp = Model().to(dev).train(), BarZDiscriminator().to(dev), PhraseZDiscriminator().to(dev)
    for d in (zb, zp):
        for p in d.parameters():
            p.requires_grad = False
    HF.manual_seed(1234, rank)
    step = PretrainStep(gen, zb, zp, Loss().to(dev), DLoss(), lr=0.002)
    step_transport = step.reducer.transport if world > 1 else None
    batch = synth_batch(args.batch, 1234 + rank, dev)
    eager_step = step
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is single-GPU: the bucketed all-reduce overlaps backward from autograd hooks")
        from hipops.train import GraphedPretrainStep
        step = GraphedPretrainStep(eager_step, *batch)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    log("model built on %s (%d CUs); set-up step" % (arch.value.decode(), cus.value))
    # set-up (not one of the W warm-up steps): the first step uploads the per-geometry gather tables and runs the conv
    # autotuner's trial launches, like the reference's first cudnn.benchmark iteration; it must not leak into the timed
    # region even with --warmup 0
    step(*batch)
    torch.cuda.synchronize()
    log("set-up step done; %d warm-up steps" % args.warmup)
    for i in range(args.warmup):
        step(*batch)
    sync_all()
    log("timing %d steps" % args.steps)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step boundaries on the step's stream
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss, _ = step(*batch)
        marks[i + 1].record()
    t_host = time.perf_counter() - t0       # the host's share: time to ENQUEUE the steps (it runs ahead of the GPU)
    sync_all()
    dt = time.perf_counter() - t0
    log("host enqueue time: %.1f ms/step" % (1e3 * t_host / args.steps))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    med_ms = step_ms[len(step_ms) // 2] if args.steps % 2 else 0.5 * (step_ms[args.steps // 2 - 1] + step_ms[args.steps // 2])
    if world > 1:
        t = torch.tensor([med_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med_ms = float(t.item())
    log("timed region: %.2f ms/step (mean over the region), %.2f ms median step" % (1e3 * dt / args.steps, med_ms))

    roof = None
    if not args.no_roofline and rank != 0:
        # the extra step below contains the gradient all-reduce: every rank has to take part in it
        import graph.model as gm
        saved = (HF.FORK_WGRAD, HF.FORK_BRANCHES, gm.OVERLAP_TRUNKS)
        HF.FORK_WGRAD = HF.FORK_BRANCHES = gm.OVERLAP_TRUNKS = False
        eager_step(*batch)
        torch.cuda.synchronize()
        HF.FORK_WGRAD, HF.FORK_BRANCHES, gm.OVERLAP_TRUNKS = saved
    if not args.no_roofline and rank == 0:
        # one extra (untimed) step with every conv launch bracketed by hipEvents on its stream
        L = nat.lib()
        if args.prof_detail:
            L.mgvae_prof_detail(args.prof_detail.encode())
        # ... on ONE stream: the timed steps overlap independent kernels on side streams (phrase / bar trunks, weight
        # vs data gradients), which would charge every kernel for its neighbours' share of the chip
        import graph.model as gm
        saved = (HF.FORK_WGRAD, HF.FORK_BRANCHES, gm.OVERLAP_TRUNKS)
        HF.FORK_WGRAD = HF.FORK_BRANCHES = gm.OVERLAP_TRUNKS = False
        L.mgvae_prof_enable(1)
        eager_step(*batch)           # per-launch events need real launches (not a graph replay)
        torch.cuda.synchronize()
        HF.FORK_WGRAD, HF.FORK_BRANCHES, gm.OVERLAP_TRUNKS = saved
        recs = (nat.ProfRec * 128)()
        n = L.mgvae_prof_collect(recs, 128)
        L.mgvae_prof_enable(0)
        L.mgvae_prof_detail(b"")
        conv = [r for r in recs[:n] if r.kind < 5 or r.kind >= 8]          # NCHW + channels-last conv families
        # the HBM-bound kernels (flat Adam, InstanceNorm): `flops` carries their ALGORITHMIC bytes (include/mgvae.h)
        hbm = [{"kernel": L.mgvae_kernel_name(r.kind, r.tile).decode(), "launches": r.launches, "ms": r.ms,
                "avg_us": 1e3 * r.ms / r.launches, "achieved": r.flops / (r.ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": r.flops / (r.ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": r.flops / r.launches} for r in recs[:n] if 5 <= r.kind < 8]
        def kernel_peak(name):
            """the matrix peak a kernel family is priced against: the fp32 instruction's 157.3, the bf16 instruction's 2500,
            and 2500 / 6 for the fp32-storage kernels that spend six bf16 MFMAs per fp32 product (csrc/conv_nhwc_x3.inc)"""
            if "_x3" in name:
                return BF16_MATRIX_PEAK_TFLOPS / 6.0
            return BF16_MATRIX_PEAK_TFLOPS if "bf16" in name else FP32_MATRIX_PEAK_TFLOPS
        fam = [{"kernel": L.mgvae_kernel_name(r.kind, r.tile).decode(), "launches": r.launches, "ms": r.ms,
                "avg_us": 1e3 * r.ms / r.launches, "tflops": r.flops / (r.ms * 1e-3) / 1e12} for r in conv]
        for f_ in fam:
            f_["peak"] = kernel_peak(f_["kernel"]); f_["frac"] = f_["tflops"] / f_["peak"]
        fam.sort(key=lambda f: -f["ms"])
        tot_ms = sum(f["ms"] for f in fam)
        tot_fl = sum(r.flops for r in conv)
        top = fam[0]
        pmc = pmc_record(top["kernel"])
        roof = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["tflops"], "peak": top["peak"],
                "unit": "TFLOP/s", "frac": top["frac"],
                "peak_note": "fp32-equivalent: 2500 TFLOP/s dense bf16 / 6 MFMAs per fp32 product" if "_x3" in top["kernel"] else None,
                "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
                "traffic_source": pmc.get("source") if pmc else None,
                "mfma_busy_pmc": pmc.get("mfma_busy") if pmc else None,
                "avg_launch_us": top["avg_us"], "launches_per_step": top["launches"],
                "measured_in": "one extra untimed step with the side streams off (kernels alone on the chip); "
                               "profiles/ holds rocprofv3 of the same command under MGVAE_SERIAL=1",
                "all_conv_kernels": {"ms_per_step": tot_ms, "tflops": tot_fl / (tot_ms * 1e-3) / 1e12,
                                     # each variant priced against its own family's peak: flops / sum(time_k * peak_k)
                                     "frac": tot_fl / (sum(f_["ms"] * 1e-3 * f_["peak"] * 1e12 for f_ in fam)),
                                     "share_of_step_time": tot_ms / (1e3 * dt / args.steps)},
                "variants": fam,
                "hbm": hbm}
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline on %d cores" % host_cores())
        base = cpu_baseline()
        log("cpu baseline done")

    if rank == 0:
        value = args.batch * world * args.steps / dt
        out = {
            "metric": "bars/sec VAE training step (fwd+bwd+opt) at 1/2/4/8 MI355X", "value": value, "unit": "bars/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_median": med_ms, "value_median_step": args.batch * world / (med_ms * 1e-3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "barGen2 pre-training generator step: PhraseEncoder + 2x Encoder + Decoder fwd, 3 frozen "
                                   "z-discriminators, Loss, bwd, Adam; Refiner excluded (reference defect D2)",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "hip_graph": bool(args.graph),
                       "rccl_ranks": {"world_size": world, "backend": hdist.backend_name()},
                       "grad_transport": step_transport,
                       "fp32_engine": HF.FP32_ENGINE if args.dtype == "f32" else None,
                       "weights": "weights_init (D4) random", "device": arch.value.decode(), "cus": cus.value},
            "step_tflops": value * FLOP_PER_BAR_STEP / 1e12, "loss": final_loss,
            "roofline": roof, "cpu_baseline": base,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
