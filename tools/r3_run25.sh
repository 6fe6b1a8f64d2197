#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3gan; mkdir -p $O
for i in 1 2 3 4 5; do
  rm -f gpurun_out/parity_report.txt
  timeout -k 10 200 python3 -m pytest tests/test_gan_parity_gpu.py -q --tb=line -k test_train_gan_iteration_bf16_storage 2>&1 | tail -2 | head -1
  grep "bf16 train_gan" gpurun_out/parity_report.txt
done
