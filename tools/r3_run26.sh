#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3bf16; mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
timeout -k 10 300 python3 bench.py --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 --prof-detail $O/conv_detail_bf16.csv > $O/bench_bf16.json 2> $O/err.txt
tail -c 300 $O/err.txt
