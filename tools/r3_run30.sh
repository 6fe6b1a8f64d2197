#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3cs; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_nhwc_gpu.py tests/test_dp_gpu.py tests/test_agent_gpu.py -x -q --tb=short -k "chained or two_rank_step_equals or bargen2 or gan_agents" 2>&1 | tail -4
timeout -k 10 200 python3 tools/host_profile.py 32 bf16 2>&1 | grep "20 steps"
for r in 1 2; do
  MGVAE_AUTOTUNE_FILE=$O/tune.txt timeout -k 10 250 python3 bench.py --no-cpu-baseline --no-roofline --dtype bf16 --batch 32 --steps 40 --warmup 5 2> $O/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r bf16 b32: %.3f ms/step (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))"
  timeout -k 10 200 python3 tools/bench_gan.py 16 bf16 20 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r gan16 bf16: %.3f ms/iteration' % d['ms_per_iteration'])"
done
