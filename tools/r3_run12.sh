#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3diag; mkdir -p $O
LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $O/lib_orig.so
for n in diag0 diag1 diag2 diag3 diag5; do
  cp $R/_ab/lib_$n.so $LIB
  echo "== $n" | tee -a $O/diag.txt
  timeout -k 10 200 python3 tools/x3_diag_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/diag.txt
done
cp $O/lib_orig.so $LIB
