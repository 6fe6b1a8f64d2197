#!/bin/bash
# round 3, GPU call 2: full GPU suite with the chained blocks, then chain on/off A/B of the host-bound configurations
set -o pipefail
O=gpurun_out/r3e2; mkdir -p $O
export MGVAE_AUTOTUNE_FILE=$O/tune.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
B="--steps 30 --warmup 5 --no-cpu-baseline --no-roofline"
run() { name=$1; shift; echo "== $name: $*"; "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }; grep -E "host enqueue|timed region" $O/$name.err; tail -1 $O/$name.json | cut -c1-200; }
for rep in 1 2; do
MGVAE_CHAIN=1 run f32_chain_$rep python bench.py $B
MGVAE_CHAIN=0 run f32_perop_$rep python bench.py $B
done
MGVAE_CHAIN=1 run bf16_32_chain python bench.py $B --dtype bf16 --batch 32
MGVAE_CHAIN=0 run bf16_32_perop python bench.py $B --dtype bf16 --batch 32
MGVAE_CHAIN=1 run bf16_64_chain python bench.py $B --dtype bf16
MGVAE_CHAIN=0 run bf16_64_perop python bench.py $B --dtype bf16
MGVAE_CHAIN=1 run gan16_bf16_chain python tools/bench_gan.py 16 bf16 20
MGVAE_CHAIN=0 run gan16_bf16_perop python tools/bench_gan.py 16 bf16 20
MGVAE_CHAIN=1 run gan16_f32_chain python tools/bench_gan.py 16 f32 20
MGVAE_CHAIN=1 python tools/host_cprofile.py bf16 > $O/cprofile_bf16_chain.txt 2>&1
