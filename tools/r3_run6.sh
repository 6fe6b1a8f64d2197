#!/bin/bash
set -o pipefail
O=gpurun_out/r3e6; mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
timeout -k 10 300 python -m pytest tests/test_nhwc_gpu.py -x -q --tb=short -k "chained" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log; tail -25 $O/pytest.log
timeout -k 10 700 python -m pytest tests/test_gan_parity_gpu.py "tests/test_hip_parity.py::test_discriminators_against_oracle_and_golden" -x -q --tb=short --durations=12 -k "segmented_against_rounding or train_gan_iteration_against or bargen_adversarial or gan2 or discriminators" > $O/pytest_gan.log 2>&1
echo "pytest gan rc=$?" | tee -a $O/pytest_gan.log; tail -30 $O/pytest_gan.log
run() { name=$1; shift; echo "== $name: $*"; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }; grep -E "host enqueue|timed region" $O/$name.err; tail -1 $O/$name.json | cut -c1-170; }
MGVAE_CHAIN=1 run gan16_bf16_chain python tools/bench_gan.py 16 bf16 20
MGVAE_CHAIN=0 run gan16_bf16_perop python tools/bench_gan.py 16 bf16 20
MGVAE_CHAIN=1 run gan16_f32_chain python tools/bench_gan.py 16 f32 20
MGVAE_CHAIN=1 run gan64_bf16_chain python tools/bench_gan.py 64 bf16 10
