#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3tl; mkdir -p $O
cd $R

export MGVAE_AUTOTUNE_FILE=$O/tune.txt
python3 bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 > $O/tr.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1); echo "trace: $f"; head -1 "$f"
cd $R && python3 tools/trace_timeline.py "$f" | tee $O/timeline_f32_b64.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr2 -o tr -- python3 $R/bench.py --no-cpu-baseline --no-roofline --dtype bf16 --batch 32 --steps 8 --warmup 2 > $O/tr2.log 2>&1
f=$(find $O/tr2 -name "*kernel_trace.csv" | head -1)
cd $R && python3 tools/trace_timeline.py "$f" | tee $O/timeline_bf16_b32.txt
rm -rf $O/tr $O/tr2
