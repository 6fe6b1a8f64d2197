#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3mt; mkdir -p $O
cd $R
cat > /tmp/mt.py <<'PY'
import sys, os, runpy
import torch
if os.environ.get("MT", "1") == "0":
    torch.autograd.set_multithreading_enabled(False)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
for r in 1 2; do
  for v in 1 0; do
    MT=$v MGVAE_AUTOTUNE_FILE=$O/tune.txt timeout -k 10 250 python3 /tmp/mt.py bench.py --no-cpu-baseline --no-roofline --dtype bf16 --batch 32 --steps 40 --warmup 5 2> $O/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('multithreading=$v round $r bf16 b32: %.3f ms/step (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))" || tail -3 $O/err.txt
    MT=$v timeout -k 10 200 python3 /tmp/mt.py tools/bench_gan.py 16 bf16 20 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('multithreading=$v round $r gan16 bf16: %.3f ms/iteration' % d['ms_per_iteration'])"
  done
done
