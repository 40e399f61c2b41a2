#!/bin/bash
# usage (on the GPU box): tools/layer_prof.sh <fp32|bf16> <tag> [crops]  -> gpurun_out/layers_<tag>.txt
set -e
PREC=$1; TAG=$2; N=${3:-4096}
ROOT=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/lp_$TAG
timeout -k 10 200 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/lp_$TAG -o t --output-format csv -- python3 $ROOT/tools/profile_forward.py $N 3 $PREC > $ROOT/gpurun_out/lp_$TAG.log 2>&1
F=$(find $ROOT/gpurun_out/lp_$TAG -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/layer_table.py $F $N > $ROOT/gpurun_out/layers_$TAG.txt
tail -2 $ROOT/gpurun_out/lp_$TAG.log
cat $ROOT/gpurun_out/layers_$TAG.txt
