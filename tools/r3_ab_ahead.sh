#!/bin/bash
# A/B of the fragment-read schedule of the x3 forward / data-gradient kernels (-DMGVAE_X3_AHEAD=0/1), prebuilt libs in _ab/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab; mkdir -p $O
LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $O/lib_orig.so
for n in ahead0 ahead1; do
  cp $R/_ab/lib_$n.so $LIB
  timeout -k 10 300 python3 tools/conv_x3_bench.py > $O/x3_bench_$n.txt 2>&1
  grep "res\|pool\|convT\|1x1\|all cases" $O/x3_bench_$n.txt | cut -c1-100
done
for r in 1 2; do
  for n in ahead0 ahead1; do
    cp $R/_ab/lib_$n.so $LIB
    MGVAE_AUTOTUNE_FILE=$O/ab_$n.txt timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 2> $O/ab_$n.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$n round $r: %.3f ms  %s %.1f us x%d %.1f TF  conv %.2f ms %.1f TF' % (d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['launches_per_step'], r['achieved'], r['all_conv_kernels']['ms_per_step'], r['all_conv_kernels']['tflops']))" || { tail -3 $O/ab_$n.err; }
  done
done
cp $O/lib_orig.so $LIB
