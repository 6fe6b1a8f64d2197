#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ab
timeout -k 10 300 python3 tools/r3_front_noise.py > gpurun_out/r3ab/front_noise.txt 2>&1; tail -24 gpurun_out/r3ab/front_noise.txt
bash tools/r3_ab_ahead.sh
