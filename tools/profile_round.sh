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
python3 bench.py --prof-detail $O/${TAG}_conv_launch_detail.csv > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
echo "bench done"; tail -c 400 $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
# per-kernel statistics with the side streams off (MGVAE_SERIAL=1): every kernel alone on the chip, as in the roofline step of bench.py
MGVAE_SERIAL=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_prof.log 2>&1 || { echo "rocprof failed"; tail -5 $O/${TAG}_prof.log; exit 1; }
echo "kernel stats done"
cd $R && MGVAE_SERIAL=1 bash tools/pmc_collect.sh ${TAG}_pmc

# the PMC summary is made HERE, on the kernel sources that were measured: it carries their tag (bench.py::kernel_source_tag)
cd $R && python3 tools/pmc_summary.py $O/${TAG}_pmc $O/${TAG}_pmc_summary.csv > $O/${TAG}_pmc_summary.txt 2>&1 || tail -5 $O/${TAG}_pmc_summary.txt
f=$(find $O/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats.csv
# BASELINE.json's other configurations (one JSON line each; not bench lines of the contract)
python3 bench.py --dtype bf16 --no-cpu-baseline > $O/${TAG}_bench_bf16.json 2> $O/${TAG}_bench_bf16.err
python3 bench.py --dtype bf16 --batch 32 --no-cpu-baseline --no-roofline > $O/${TAG}_bench_bf16_b32.json 2> $O/${TAG}_bench_bf16_b32.err
MGVAE_CHAIN=0 python3 bench.py --dtype bf16 --batch 32 --no-cpu-baseline --no-roofline > $O/${TAG}_bench_bf16_b32_perop.json 2> $O/${TAG}_bench_bf16_b32_perop.err
python3 tools/bench_gan.py 16 bf16 20 > $O/${TAG}_gan_b16_bf16.json 2>/dev/null
MGVAE_CHAIN=0 python3 tools/bench_gan.py 16 bf16 20 > $O/${TAG}_gan_b16_bf16_perop.json 2>/dev/null
python3 tools/bench_gan.py 16 f32 20 > $O/${TAG}_gan_b16_f32.json 2>/dev/null
python3 tools/bench_gan.py 64 bf16 10 > $O/${TAG}_gan_b64_bf16.json 2>/dev/null
python3 tools/bench_sampling.py 32 10 > $O/${TAG}_sampling_32songs.txt 2>&1
tail -3 $O/${TAG}_sampling_32songs.txt
grep -h "host enqueue\|timed region" $O/${TAG}_bench*.err
cat $O/${TAG}_gan_*.json
