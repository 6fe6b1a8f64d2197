#!/bin/bash
set -o pipefail
O=gpurun_out/r3e10; mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
timeout -k 10 300 python -m pytest tests/test_nhwc_gpu.py -x -q --tb=short -k "chained" > $O/pytest.log 2>&1
echo "pytest chained rc=$?" | tee -a $O/pytest.log; tail -25 $O/pytest.log
timeout -k 10 800 python -m pytest tests/test_gan_parity_gpu.py tests/test_agent_gpu.py "tests/test_hip_parity.py::test_train_step_against_oracle_and_golden" "tests/test_hip_parity.py::test_generator_forward_against_golden" -x -q --tb=short --durations=8 > $O/pytest_gan.log 2>&1
echo "pytest gan rc=$?" | tee -a $O/pytest_gan.log; tail -30 $O/pytest_gan.log
run() { name=$1; shift; echo "== $name: $*"; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }; grep -E "host enqueue|timed region" $O/$name.err; grep "^{" $O/$name.json | tail -1 | cut -c1-170; }
run gan16_bf16_chain python tools/bench_gan.py 16 bf16 20
run gan16_f32_chain python tools/bench_gan.py 16 f32 20
run bf16_32_chain python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --dtype bf16 --batch 32
run f32_chain python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline
python tools/bench_sampling.py 32 10 2>/dev/null | grep "^{" | cut -c1-300
