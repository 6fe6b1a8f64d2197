#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3i6; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu --tb=short --durations=8 > $O/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_full.log
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
timeout -k 10 250 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
