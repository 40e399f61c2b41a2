#!/bin/bash
# usage: tools/pmc_one.sh <binary> <args...> ; runs SQ counter passes for one conv_bench shape, prints per-kernel averages
set -e
BIN=$1; shift
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- $GRAFT_REPO_ROOT/$BIN "$@" > $OUT/p1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- $GRAFT_REPO_ROOT/$BIN "$@" > $OUT/p2.log 2>&1
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    f=glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True)
    if not f: print("no counter file for",p); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k)
        for c,vals in v.items():
            print("   %-32s n=%d avg=%.4g"%(c,len(vals),sum(vals)/len(vals)))
PY
