#!/usr/bin/env python3
"""Per-layer backbone table from a rocprofv3 kernel trace of tools/profile_forward.py.
   python tools/layer_table.py <kernel_trace.csv> [crops]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ks = [r for r in rows if "conv" in r["Kernel_Name"] or "stem" in r["Kernel_Name"] or "block32_fused" in r["Kernel_Name"] or "stage1_depth" in r["Kernel_Name"]]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ks) if "stem" in r["Kernel_Name"]]
seq = ks[idx[-1]:]


def conv(cin, cout, hw, k, stride=1):
    ho = hw // stride
    return 2 * cin * cout * k * k * ho * ho


L = [("stem", 2 * 9 * 32 * 96 * 96)]
hw = 48
for cin, cout, st, nb in [(32, 32, 1, 2), (32, 64, 2, 3), (64, 128, 2, 5), (128, 256, 2, 2)]:
    for b in range(nb):
        ci, s = (cin, st) if b == 0 else (cout, 1)
        if b == 0 and st != 1:      # conv1 and the 1x1 down-sample branch are one launch (conv3x3s2_ds_*)
            L.append((f"c1+ds {ci}->{cout} hw{hw} s{s}", conv(ci, cout, hw, 3, s) + conv(ci, cout, hw, 1, s)))
        else:
            L.append((f"c1 {ci}->{cout} hw{hw} s{s}", conv(ci, cout, hw, 3, s)))
        hw //= s
        L.append((f"c2 {cout}->{cout} hw{hw}", conv(cout, cout, hw, 3)))
L.append(("proj", conv(256, 72, 6, 1)))
tot = 0
li = 0
for r in seq:
    if li >= len(L):
        break
    name, fl = L[li]
    li += 1
    if "stage1_depth" in r["Kernel_Name"]:       # both stage-1 blocks + the 32 -> 64 stride-2 layer + down-sample in one launch (bf16 mode)
        for _ in range(4):
            n2, f2 = L[li]
            li += 1
            fl += f2
        name = "stage 1 + c1+ds 32->64 (depth-first)"
    if "block32_fused" in r["Kernel_Name"]:      # conv1 + conv2 of a stage-1 block in one launch (bf16 mode)
        n2, f2 = L[li]
        li += 1
        name, fl = "c1+c2 32->32 hw48 (fused)", fl + f2
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    kn = r["Kernel_Name"].replace("void umt::", "").split("(")[0]
    # rocprofv3's VGPR_Count is half the kernel's register allocation per lane (.amdhsa_next_free_vgpr, accumulators
    # included) on this toolchain; `regs` is the allocation that sets the waves per SIMD (512 / regs)
    print(f"{name:36s} {kn:34s} {d:9.1f} us {fl * n / d / 1e6:7.1f} TF regs {2 * int(r['VGPR_Count']):>4d} lds {r['LDS_Block_Size']}")
print(f"total {tot / 1e3:.3f} ms -> {sum(f for _, f in L) * n / tot / 1e6:.1f} TFLOP/s")
