#!/bin/bash
# A/B of environment switches on ONE box: tools/ab_env.sh <rounds> "<VAR=a ...>" "<VAR=b ...>" ...   (bench.py per setting, interleaved)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e MGVAE_AUTOTUNE_FILE=$O/abenv_$i.txt timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2> $O/abenv_$i.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$e] round $r: %.3f ms  %s %.1f us x%d %.1f TF  conv %.1f TF' % (d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['launches_per_step'], r['achieved'], r['all_conv_kernels']['tflops']))" || { tail -3 $O/abenv_$i.err; exit 1; }
  done
done
