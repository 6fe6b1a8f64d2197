#!/bin/bash
# tools/bench_gan.py under each _ab/lib_<name>.so: tools/ab_gan.sh <batch> <name>...
R=$GRAFT_REPO_ROOT; LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $R/_ab/lib_orig.so
B=$1; shift
for n in "$@"; do cp $R/_ab/lib_$n.so $LIB; echo "== $n"; MGVAE_AUTOTUNE_FILE=$R/gpurun_out/abgan_$n.txt timeout -k 10 300 python3 tools/bench_gan.py $B f32 2>&1 | tail -1 | cut -c1-300; done
cp $R/_ab/lib_orig.so $LIB
