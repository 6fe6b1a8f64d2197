#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3blk; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_nhwc_gpu.py -x -q --tb=short -k "x3 or halo or workspace or chained_blocks" 2>&1 | tail -6
timeout -k 10 300 python3 tools/x3_variant_sweep.py > $O/sweep.txt 2>&1; tail -2 $O/sweep.txt
timeout -k 10 300 python3 tools/conv_x3_bench.py 2>&1 | grep "res\|pool\|convT\|1x1\|all cases" | cut -c1-100 | tee $O/x3_bench.txt
