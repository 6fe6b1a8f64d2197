#!/bin/bash
set -o pipefail
O=gpurun_out/r3e8; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_nhwc_gpu.py -x -q --tb=short -k "halo" > $O/pytest.log 2>&1
echo "pytest halo rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
for rep in 1 2; do
for c in 1 0; do
MGVAE_CHAIN=$c timeout -k 10 300 python -m pytest "tests/test_hip_parity.py::test_bf16_storage_step_against_bf16_rounding_oracle" -x -q --tb=line > $O/bf16step_chain${c}_$rep.log 2>&1
echo "bf16 step chain=$c rep $rep rc=$?"; grep "^E  \|AssertionError\|passed\|failed" $O/bf16step_chain${c}_$rep.log | head -4
done; done
grep "bf16 step" gpurun_out/parity_report.txt | tail -8
timeout -k 10 300 python -m pytest tests/test_gan_parity_gpu.py -x -q --tb=short -k "segmented_against_rounding" > $O/pytest_seg.log 2>&1
echo "pytest seg rc=$?"; tail -12 $O/pytest_seg.log
MGVAE_X3_FORMS=0123 timeout -k 10 300 python tools/conv_x3_bench.py > $O/x3_bench_nohalo.txt 2>&1
timeout -k 10 300 python tools/conv_x3_bench.py > $O/x3_bench_halo.txt 2>&1
grep "res\|all cases" $O/x3_bench_nohalo.txt | cut -c1-100; grep "res\|all cases" $O/x3_bench_halo.txt | cut -c1-100
B="--steps 30 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; echo "== $name: $*"; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }; grep -E "timed region" $O/$name.err; python - $O/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('  %.3f ms/step  top %s %.1f us x%d %.1f TF (frac %.3f)  conv family %.2f ms %.1f TF' % (d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['launches_per_step'], r['achieved'], r['frac'], r['all_conv_kernels']['ms_per_step'], r['all_conv_kernels']['tflops']))
for v in r['variants'][:12]: print('     %-40s x%3d %7.1f us  %6.1f TF' % (v['kernel'], v['launches'], v['avg_us'], v['tflops']))
PY
}
for rep in 1 2; do
MGVAE_AUTOTUNE_FILE=$O/tune_halo.txt run f32_halo_$rep python bench.py $B
MGVAE_X3_FORMS=0123 MGVAE_AUTOTUNE_FILE=$O/tune_nohalo.txt run f32_nohalo_$rep python bench.py $B
done
