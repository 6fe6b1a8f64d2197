#!/bin/bash
# A/B of prebuilt library variants on ONE box: for each _ab/lib_<name>.so, install it as libmgvae_hip.so and run bench.py
# (autotune decisions per variant, interleaved rounds).   usage: tools/ab_libs.sh <rounds> <name> [<name> ...]
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $O/../_ab/lib_orig.so
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for n in "$@"; do
    cp $R/_ab/lib_$n.so $LIB
    MGVAE_AUTOTUNE_FILE=$O/ab_$n.txt timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2> $O/ab_$n.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$n round $r: %.3f ms  %s %.1f us x%d  conv %.1f TF' % (d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['launches_per_step'], r['all_conv_kernels']['tflops']))" || { tail -3 $O/ab_$n.err; exit 1; }
  done
done
cp $R/_ab/lib_orig.so $LIB
