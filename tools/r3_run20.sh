#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab4; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_nhwc_gpu.py -x -q --tb=short -k "batch_norm or discriminator or chained_ends or refiner" 2>&1 | tail -5
for r in 1 2; do
  for n in old new; do
    if [ $n = old ]; then D=$R/_ab/old; else D=$R; fi
    cd $D
    for cfg in "16 bf16" "16 f32"; do
      set -- $cfg
      timeout -k 10 200 python3 tools/bench_gan.py $1 $2 20 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n round $r gan $1 $2: %.3f ms/iteration' % d['ms_per_iteration'])"
    done
  done
done
