#!/bin/bash
# round 3, GPU call 3: the new chains (MLP, trunk entry), segmented backward, refiner loop; host-bound configs again
set -o pipefail
O=gpurun_out/r3e3; mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
timeout -k 10 1000 python -m pytest tests/test_nhwc_gpu.py tests/test_gan_parity_gpu.py tests/test_agent_gpu.py tests/test_dp_gpu.py "tests/test_hip_parity.py::test_train_step_against_oracle_and_golden" "tests/test_hip_parity.py::test_full_size_step_parity_and_batch_properties" "tests/test_hip_parity.py::test_discriminators_against_oracle_and_golden" "tests/test_hip_parity.py::test_graphed_train_step_matches_eager" -q --maxfail=10 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -12 $O/pytest.log
B="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline"
run() { name=$1; shift; echo "== $name: $*"; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }; grep -E "host enqueue|timed region" $O/$name.err; tail -1 $O/$name.json | cut -c1-160; }
run f32_chain python bench.py $B
run bf16_32_chain python bench.py $B --dtype bf16 --batch 32
run bf16_64_chain python bench.py $B --dtype bf16
run gan16_bf16_chain python tools/bench_gan.py 16 bf16 20
run gan16_f32_chain python tools/bench_gan.py 16 f32 20
python tools/host_cprofile.py bf16 > $O/cprofile_bf16_chain.txt 2>&1
python tools/host_profile.py bf16 > $O/hostprofile_bf16_chain.txt 2>&1
