#!/bin/bash
# Diagnostic PMC passes for the conv kernels (stall breakdown, instruction mix, vector-memory pipe, L2): separate
# (at most two TA / TCP counters per pass: four at once exceed the block's counter slots -- rocprofiler error 38 in round 1)
# --pmc passes with --kernel-trace only, MGVAE_SERIAL=1 (kernels alone on the chip).  usage: tools/pmc_diag.sh <dir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_diag}
mkdir -p $OUT
export MGVAE_AUTOTUNE_FILE=$OUT/autotune.txt MGVAE_SERIAL=1
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/plain.json 2> $OUT/plain.err || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for pass in \
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
 "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
 "TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; continue; }
  echo "pass $i done"
done
