#!/bin/bash
set -o pipefail
O=gpurun_out/r3full; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout -k 10 1150 python -m pytest tests -m gpu -q --tb=short --durations=25 --maxfail=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -60 $O/pytest.log
