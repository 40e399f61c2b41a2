#!/bin/bash
# head stage time of the bench step with another build of the library: tools/head_time.sh <lib.so|base>
R=$GRAFT_REPO_ROOT
[ "$1" = base ] || export UMT_LIB_PATH=$R/$1
timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 2 --cpu-hands 0 --no-pcie --no-x3 --per-frame-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'head ms', [round(s['ms'],3) for s in d['stages'] if s['stage'].startswith('head')], 'step', round(d['ms_per_step'],2))"
