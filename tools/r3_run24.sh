#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3aux; mkdir -p $O
LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $O/lib_orig.so
for r in 1 2; do
for n in aux0 aux2 aux1 aux3; do
  cp $R/_ab/lib_$n.so $LIB
  timeout -k 10 300 python3 tools/conv_x3_bench.py 2>&1 | grep "res128 96\|res256 48\|res512 24\|all cases" | cut -c1-100 | sed "s/^/$n round $r: /"
done
done
cp $O/lib_orig.so $LIB
