#!/bin/bash
# run one pytest selection under each _ab/lib_<name>.so several times: tools/ab_test.sh <k-expr> <reps> <name>...
R=$GRAFT_REPO_ROOT; LIB=$R/musicgeneration_vae-torch_amd/libmgvae_hip.so
cp $LIB $R/_ab/lib_orig.so
K=$1; REPS=$2; shift; shift
for n in "$@"; do cp $R/_ab/lib_$n.so $LIB; for i in $(seq 1 $REPS); do echo "== $n $i"; timeout -k 10 200 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | grep -E "AssertionError:|passed|failed" | cut -c1-200; done; done
cp $R/_ab/lib_orig.so $LIB
