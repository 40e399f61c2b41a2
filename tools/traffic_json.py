#!/usr/bin/env python3
"""HBM traffic of the backbone per crop from two rocprofv3 --pmc passes over tools/profile_forward.py.
   python tools/traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <crops> <iters>
Prints one JSON object.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section), so it is doubled."""
import csv
import json
import sys


def total(path, counter):
    s = 0.0
    hits = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not ("conv" in k or "stem" in k):
            continue
        if r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
        elif r["Counter_Name"] in ("TCC_HIT_sum", "TCC_MISS_sum"):
            hits[r["Counter_Name"]] = hits.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return s, hits


fetch, _ = total(sys.argv[1], "FETCH_SIZE")
write, hits = total(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[3]) * int(sys.argv[4])
out = {"fetch_size_KiB_per_crop_raw": fetch / n, "read_bytes_per_crop": int(2 * fetch * 1024 / n),
       "write_bytes_per_crop": int(write * 1024 / n)}
if hits.get("TCC_HIT_sum"):
    out["l2_hit_rate"] = hits["TCC_HIT_sum"] / (hits["TCC_HIT_sum"] + hits.get("TCC_MISS_sum", 0.0))
print(json.dumps(out))
