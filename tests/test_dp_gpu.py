"""GPU suite: the data-parallel training step end to end with 2 ranks (gloo transport, both ranks
on the single GPU of the test box -- RCCL needs one GPU per rank): hooks, early decoder / bar-encoder bucket
all-reduces launched from the step's main stream, reduce_rest, fused Adam with 1/world; result equals the
single-process step on the concatenated batch.  Variants: the two-pass (variational) encoder, the bf16 gradient
transport, and the first agent across its pre-training boundary with config.seed = None."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(port, *args, env_extra=None, timeout=600):
    # MGVAE_FORK_MIN_BATCH=1: the weight gradients are forked onto the side stream even at this tiny batch, so the
    # stream-count check (<= 2 side streams under torch.distributed) sees the streams a full-size step creates
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MGVAE_FORK_MIN_BATCH="1", MGVAE_FORK_WGRAD="1",
               MGVAE_STACK_RANKS="1")     # both gloo ranks share the one GPU of the test box (agent/base.py)
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_check.py")] + list(args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    out = r.stdout + r.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dp_check_%d.log" % port), "w") as f:
        f.write(out)
    assert "DPCHECK PASS" in out, "\n".join(l for l in out.splitlines() if "Error" in l or "error" in l or "DPCHECK" in l)[-3000:]
    return out


def test_two_rank_step_equals_single_process_step():
    out = _run(29611)
    assert "side_streams=2" in out or "side_streams=1" in out


def test_two_rank_step_with_two_pass_encoder_arms_only_the_decoder_bucket():
    _run(29613, "--variational")


def test_two_rank_step_bf16_gradient_transport():
    _run(29614, "--transport", "bf16")


def test_two_rank_step_bf16_storage():
    """BASELINE.json configs[2]'s shape of run (bf16 storage inside the island + data parallel) at two ranks: identical
    weights on both ranks, stream budget, early buckets; the all-reduced gradient equals (1e-2 or 3x the reference's own
    run-to-run noise) the average of the two shards' single-process steps -- same launches: the tuner is pinned off -- and
    agrees with the single-process 4-bar bf16 step in direction and size"""
    out = _run(29616, "--dtype", "bf16", env_extra={"MGVAE_AUTOTUNE": "0"})
    assert "DPCHECK bf16 cosine" in out and "DPCHECK bf16 micro-step reference" in out


def test_first_agent_two_ranks_unseeded_across_the_pretraining_boundary(tmp_path):
    _run(29615, "--agent", env_extra={"MGVAE_TEST_ROOT": str(tmp_path)}, timeout=900)


def test_bench_two_ranks_gloo():
    """bench.py's own multi-rank path exactly as the driver launches it -- default flags, i.e. WITH the extra
    roofline step (it contains the gradient all-reduce, so every rank must take part in it): barrier, max-over-ranks
    timing, one JSON line from rank 0"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGVAE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29612", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"] is not None and d["cpu_baseline"] is None      # cpu baseline is an N=1 leg
    assert d["config"]["rccl_ranks"] == {"world_size": 2, "backend": "gloo"}
