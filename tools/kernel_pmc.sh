#!/bin/bash
# PMC passes over the bf16 forward (tools/profile_forward.py), summarised per kernel
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/sp1 -o p --output-format csv -- python3 $R/tools/profile_forward.py 4096 1 bf16 > /tmp/sp1.log 2>&1 || { tail -5 /tmp/sp1.log; exit 1; }
python3 $R/tools/pmc_table.py $(find /tmp/sp1 -name "*counter_collection.csv") | grep -E "${KPAT:-stem|stage1}"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/sp2 -o p --output-format csv -- python3 $R/tools/profile_forward.py 4096 1 bf16 > /tmp/sp2.log 2>&1 || { tail -5 /tmp/sp2.log; exit 1; }
python3 - $(find /tmp/sp2 -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    import os, re
    if re.search(os.environ.get("KPAT", "stem|stage1"), k):
        agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    wc = v["SQ_WAVE_CYCLES"]
    print(k, {c: round(x / wc, 3) if c.startswith("SQ_ACTIVE") else x for c, x in v.items()})
PY
