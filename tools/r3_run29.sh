#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3mt2; mkdir -p $O
cd $R
for r in 1 2; do
  for v in 1 0; do
    MGVAE_AUTOGRAD_THREAD=$v MGVAE_AUTOTUNE_FILE=$O/tune.txt timeout -k 10 250 python3 bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 2> $O/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('autograd_thread=$v round $r f32 b64: %.3f ms/step (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))" || tail -3 $O/err.txt
  done
done
bash tools/r3_fullsuite.sh
