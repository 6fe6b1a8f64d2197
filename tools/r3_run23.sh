#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pair; mkdir -p $O
LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $O/lib_orig.so
for r in 1 2; do
for n in pair0 pair1; do
  cp $R/_ab/lib_$n.so $LIB
  timeout -k 10 300 python3 tools/conv_x3_bench.py 2>&1 | grep "all cases" | cut -c1-100 | sed "s/^/$n round $r: /"
done
done
for n in pair0 pair1; do
  cp $R/_ab/lib_$n.so $LIB
  MGVAE_AUTOTUNE_FILE=$O/tune_$n.txt timeout -k 10 250 python3 bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 2> $O/err_$n.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n: %.3f ms/step (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))"
done
cp $O/lib_orig.so $LIB
