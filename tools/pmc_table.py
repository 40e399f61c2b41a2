#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
seen = set()
for r in rows:
    k = r["Kernel_Name"].replace("void umt::", "").split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    if dur[k] < 1e5:
        continue
    line = f"{k:40s} {dur[k] / 1e6:8.2f} ms"
    if "GRBM_GUI_ACTIVE" in v:
        line += f"  clk {v['GRBM_GUI_ACTIVE'] / 8 / dur[k]:.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        line += f"  mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (v['GRBM_GUI_ACTIVE'] / 8):.3f}"
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
        if c in v and "SQ_WAVE_CYCLES" in v:
            line += f"  {c[3:].lower()} {v[c] / v['SQ_WAVE_CYCLES']:.2f}"
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "TCC_HIT_sum", "TCC_MISS_sum"):
        if c in v:
            line += f"  {c} {v[c]:.4g}"
    print(line)
