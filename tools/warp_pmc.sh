#!/bin/bash
# PMC pass over tools/warp_bench.py (GPU box): instruction mix and stall counters of warp_kernel
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/wpmc; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/a -o a --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/warp_bench.py umetrack_amd/libumetrack_hip.so > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE -d $OUT/b -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/warp_bench.py umetrack_amd/libumetrack_hip.so > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for tag in ('a','b'):
    f=glob.glob('$OUT/%s/**/*counter_collection.csv'%tag,recursive=True)
    if not f: print(tag,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        if 'warp_kernel' in k: print(k, {c: round(x/1e6,2) for c,x in v.items()})
PY
