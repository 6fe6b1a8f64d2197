"""GPU suite: the data-parallel training step end to end with 2 ranks (gloo transport, both ranks
on the single GPU of the test box -- RCCL needs one GPU per rank): hooks, early decoder bucket
all-reduce, reduce_rest, fused Adam with 1/world; result equals the single-process step on the
concatenated batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_equals_single_process_step():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "tests", "dp_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "DPCHECK PASS" in out, out[-3000:]


def test_bench_two_ranks_gloo():
    """bench.py's own multi-rank path exactly as the driver launches it -- default flags, i.e. WITH the extra
    roofline step (it contains the gradient all-reduce, so every rank must take part in it): barrier, max-over-ranks
    timing, one JSON line from rank 0"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGVAE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29612", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"] is not None and d["cpu_baseline"] is None      # cpu baseline is an N=1 leg
