#!/bin/bash
# round 3, first GPU call: default-config agent tests (spawned loader worker) + where hipGraphLaunch's time goes
set -o pipefail
O=gpurun_out/r3e1; mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
python -m pytest tests/test_agent_gpu.py tests/test_checkpoint_gpu.py "tests/test_hip_parity.py::test_graphed_train_step_matches_eager" \
   "tests/test_dp_gpu.py::test_first_agent_two_ranks_unseeded_across_the_pretraining_boundary" -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
run() { name=$1; shift; echo "== $name: $*"; "$@" > $O/$name.json 2> $O/$name.err; grep -E "host enqueue|timed region" $O/$name.err; }
run f32_eager python bench.py $B
run f32_graph_multi python bench.py $B --graph
MGVAE_SERIAL=1 run f32_graph_serial python bench.py $B --graph
MGVAE_SERIAL=1 run f32_eager_serial python bench.py $B
run bf16_32_eager python bench.py $B --dtype bf16 --batch 32
MGVAE_SERIAL=1 run bf16_32_graph_serial python bench.py $B --dtype bf16 --batch 32 --graph
MGVAE_SERIAL=1 run bf16_32_eager_serial python bench.py $B --dtype bf16 --batch 32
MGVAE_SERIAL=1 run bf16_64_graph_serial python bench.py $B --dtype bf16 --graph
run bf16_64_eager python bench.py $B --dtype bf16
