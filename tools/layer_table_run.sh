cd /tmp; export TMPDIR=/tmp
P=${1:-bf16}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/lt; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/lt
timeout -k 10 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/lt -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/profile_forward.py 4096 3 $P > $GRAFT_REPO_ROOT/gpurun_out/lt.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/layer_table.py $(find $GRAFT_REPO_ROOT/gpurun_out/lt -name '*kernel_trace.csv' | head -1) 4096
