#!/bin/bash
# kernel stats of the per-frame tracker path (256 frames twice): tools/pf_stats.sh [pattern]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pfs
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pfs -o p --output-format csv -- python3 $R/tools/per_frame_prof.py plain 256 > /tmp/pfs.log 2>&1 || { tail -5 /tmp/pfs.log; exit 1; }
tail -n 1 /tmp/pfs.log
grep -E "${1:-small|halo_f32<4|s2p|chain3|decode|igemm}" /tmp/pfs/p_kernel_stats.csv | sed 's/(umt::[A-Za-z0-9]*)//; s/"//g' | awk -F, '{printf "%-70s calls %6d avg %8.1f us\n", substr($1,1,70), $(NF-6), $(NF-4)/1000}'
