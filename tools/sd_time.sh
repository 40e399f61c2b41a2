#!/bin/bash
# time of the bf16 backbone kernels at 4096 crops (rocprofv3 kernel stats) + the bit-identity test of the depth-first kernel
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest $R/tests/test_gpu_bf16.py -x -q --timeout 120 --timeout-method thread -k "bit_identical" > $R/gpurun_out/sd_test.log 2>&1 || { tail -20 $R/gpurun_out/sd_test.log; exit 1; }
tail -n 1 $R/gpurun_out/sd_test.log
rm -rf $R/gpurun_out/lt_bf16
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lt_bf16 -o t --output-format csv -- python3 $R/tools/profile_forward.py ${1:-4096} 3 bf16 > $R/gpurun_out/lt_bf16.log 2>&1
grep "crops" $R/gpurun_out/lt_bf16.log
