#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3small; mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
cd $R && python3 bench.py --no-cpu-baseline --no-roofline --dtype bf16 --batch 32 --steps 5 --warmup 2 > /dev/null 2>&1   # tuner file
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b32 -o b32 -- python3 $R/bench.py --no-cpu-baseline --no-roofline --dtype bf16 --batch 32 --steps 10 --warmup 2 > $O/b32.log 2>&1
f=$(find $O/b32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/b32_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gan -o gan -- python3 $R/tools/bench_gan.py 16 bf16 10 > $O/gan.log 2>&1
f=$(find $O/gan -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/gan16_kernel_stats.csv
rm -rf $O/b32 $O/gan
ls -la $O
