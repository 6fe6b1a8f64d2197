#!/bin/bash
# Collect the PMC passes behind bench.py's roofline.traffic on the GPU box (separate --pmc passes,
# kernel trace only, as MI355X_MICROARCH.md prescribes).  Usage: tools/pmc_collect.sh <out-dir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
mkdir -p $OUT
export MGVAE_AUTOTUNE_FILE=$OUT/autotune.txt
cd $GRAFT_REPO_ROOT
# 1) un-profiled run: fills the autotune file so that the profiled runs contain no trial launches
python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_plain.json 2> $OUT/bench_plain.err || exit 1
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; exit 1; }
  echo "pass $tag done"
done
ls -R $OUT | head -40
