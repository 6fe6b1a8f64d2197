#!/bin/bash
# timing-only ablation study of the tiled conv kernel: tools/conv_bench.py under each _ab/lib_<name>.so, all with the
# tile / split-K decisions of the first variant.   usage: tools/ab_conv.sh <name> [<name> ...]
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $R/_ab/lib_orig.so
export MGVAE_AUTOTUNE_FILE=$O/ab_conv_choices.txt CONV_BENCH_NO_DIRECT=1
rm -f $MGVAE_AUTOTUNE_FILE
for n in "$@"; do
  cp $R/_ab/lib_$n.so $LIB
  echo "== $n"; timeout -k 10 200 python3 tools/conv_bench.py 2> $O/ab_conv_$n.err | tee $O/ab_conv_$n.txt || { tail -3 $O/ab_conv_$n.err; break; }
done
cp $R/_ab/lib_orig.so $LIB
