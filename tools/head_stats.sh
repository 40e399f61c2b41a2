#!/bin/bash
# kernel stats of one bench run, head-side kernels only (GPU box); extra arguments go to bench.py (e.g. --precision bf16)
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/hs; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-hands 0 --no-pcie --no-x3 --per-frame-frames 0 "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('halo_f32<2, 2','wlds32','s2_ds','stem','at::native','rocclr')): continue
    print(f"{n[:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
