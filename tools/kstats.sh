#!/bin/bash
# per-kernel totals of a few bench steps (rocprofv3 --kernel-trace --stats, side streams off so a kernel's duration is its own).
# usage: tools/kstats.sh <tag> [extra bench.py flags]     -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:-kst}; shift || true
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/${TAG}_autotune.txt MGVAE_SERIAL=1
cd $R && python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $O/${TAG}_plain.json 2> $O/${TAG}_plain.err || { tail -5 $O/${TAG}_plain.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $O/${TAG}_prof.log 2>&1 || { echo "rocprof failed"; tail -5 $O/${TAG}_prof.log; exit 1; }
f=$(find $O/${TAG}_prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/${TAG}_kernel_stats.csv && echo "wrote $O/${TAG}_kernel_stats.csv"
