#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab5; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_nhwc_gpu.py -x -q --tb=short -k "norm_cbam or chained or blocks" 2>&1 | tail -4
for r in 1 2; do
  for n in old new; do
    if [ $n = old ]; then D=$R/_ab/old; else D=$R; fi
    cd $D
    for cfg in "f32 64" "bf16 32"; do
      set -- $cfg
      MGVAE_AUTOTUNE_FILE=$O/tune_$n.txt timeout -k 10 250 python3 bench.py --no-cpu-baseline --no-roofline --dtype $1 --batch $2 --steps 30 --warmup 5 2> $O/err_$n.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n round $r $1 b$2: %.3f ms/step (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))" || tail -3 $O/err_$n.txt
    done
  done
done
