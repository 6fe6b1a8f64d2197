#!/bin/bash
# tools/bench_gan.py per environment setting: tools/ab_gan_env.sh <batch> "VAR=a" "VAR=b" ...
R=$GRAFT_REPO_ROOT; B=$1; shift
for e in "$@"; do echo "== $e"; env $e MGVAE_AUTOTUNE_FILE=$R/gpurun_out/abgan_env.txt timeout -k 10 300 python3 tools/bench_gan.py $B f32 2>&1 | tail -1 | cut -c100-300; done
