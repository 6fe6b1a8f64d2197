#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3split; mkdir -p $O
cd $R
for r in 1 2; do
  for v in 0 1; do
    MGVAE_SPLIT_DECODER=$v MGVAE_AUTOTUNE_FILE=$O/tune_$v.txt timeout -k 10 250 python3 bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 2> $O/err_$v.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split_decoder=$v round $r: %.3f ms/step (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))" || tail -3 $O/err_$v.txt
  done
done
