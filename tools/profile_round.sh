#!/bin/bash
# One round's evidence on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes.
# usage: tools/profile_round.sh <tag>      (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-rX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/${TAG}_autotune.txt
rm -f $MGVAE_AUTOTUNE_FILE
cd $R
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
echo "bench done"; tail -c 400 $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
# per-kernel statistics with the side streams off (MGVAE_SERIAL=1): every kernel alone on the chip, as in the roofline step of bench.py
MGVAE_SERIAL=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_prof.log 2>&1 || { echo "rocprof failed"; tail -5 $O/${TAG}_prof.log; exit 1; }
echo "kernel stats done"
cd $R && MGVAE_SERIAL=1 bash tools/pmc_collect.sh ${TAG}_pmc
