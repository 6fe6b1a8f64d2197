#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i6
timeout -k 10 200 python3 tools/x3_thin_bench.py 2>&1 | grep -v "amdgpu.ids\|warning\|^ " | tee gpurun_out/r3i6/x3_thin_bench.txt
timeout -k 10 300 python3 -m pytest tests/test_nhwc_gpu.py -x -q --tb=short -k "one_channel or to_one or cast or chained_ends" 2>&1 | tail -5 | tee gpurun_out/r3i6/pytest1.txt
timeout -k 10 200 python3 tools/x3_twin_bench.py 2>&1 | grep -v "amdgpu.ids\|warning\|^ " | tee gpurun_out/r3i6/x3_twin_bench.txt
