#!/bin/bash
# kernel stats of a forward pass with another build of the library:  tools/lib_time.sh <lib.so|base> <fp32|bf16> <pattern>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ "$1" = base ] || export UMT_LIB_PATH=$R/$1
rm -rf /tmp/lt_x
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/lt_x -o t --output-format csv -- python3 $R/tools/profile_forward.py 4096 3 $2 > /tmp/lt_x.log 2>&1 || { tail -5 /tmp/lt_x.log; exit 1; }
grep -E "$3" /tmp/lt_x/t_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3
