#!/bin/bash
# round 3, GPU call 4: the halo form of the 3x3 stride-1 x3 convs -- parity, then per-layer and whole-step A/B against the
# implicit-GEMM forms alone (MGVAE_X3_FORMS=0123 hides form 4 from the tuner)
set -o pipefail
O=gpurun_out/r3e4; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_nhwc_gpu.py -x -q --tb=short --durations=8 -k "halo or three_products or workspace or chained" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -30 $O/pytest.log
MGVAE_X3_FORMS=0123 timeout -k 10 300 python tools/conv_x3_bench.py > $O/x3_bench_nohalo.txt 2>&1
timeout -k 10 300 python tools/conv_x3_bench.py > $O/x3_bench_halo.txt 2>&1
paste -d'\n' <(grep "3x3\|res\|all cases" $O/x3_bench_nohalo.txt | cut -c1-100) <(grep "3x3\|res\|all cases" $O/x3_bench_halo.txt | cut -c1-100) | head -60
B="--steps 30 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; echo "== $name: $*"; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }; grep -E "host enqueue|timed region" $O/$name.err; python - $O/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('  %.3f ms/step  top %s %.1f us x%d %.1f TF (frac %.3f)  conv family %.2f ms %.1f TF' % (d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['launches_per_step'], r['achieved'], r['frac'], r['all_conv_kernels']['ms_per_step'], r['all_conv_kernels']['tflops']))
for v in r['variants'][:8]: print('     %-40s x%3d %7.1f us  %6.1f TF' % (v['kernel'], v['launches'], v['avg_us'], v['tflops']))
PY
}
for rep in 1 2; do
MGVAE_AUTOTUNE_FILE=$O/tune_halo.txt run f32_halo_$rep python bench.py $B
MGVAE_X3_FORMS=0123 MGVAE_AUTOTUNE_FILE=$O/tune_nohalo.txt run f32_nohalo_$rep python bench.py $B
done

timeout -k 10 700 python -m pytest tests/test_gan_parity_gpu.py -x -q --tb=short --durations=12 > $O/pytest_gan.log 2>&1
echo "pytest gan rc=$?" | tee -a $O/pytest_gan.log
tail -40 $O/pytest_gan.log
