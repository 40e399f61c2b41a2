#!/bin/bash
# copies what tools/round_profiles.sh left under gpurun_out/rp/ into profiles/ (run in the build container)
set -e
cd "$(dirname "$0")/.."
R=${1:-r03}
export UMT_MEASURED_HEAD="$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- umetrack_amd include || echo '+worktree')"
cp $(find gpurun_out/rp/bench -name '*kernel_stats.csv' | head -1) profiles/${R}_bench_kernel_stats.csv
for P in fp32 bf16 fp32x3; do
  python3 tools/layer_table.py gpurun_out/rp/lt_$P/t_kernel_trace.csv 4096 > profiles/${R}_layer_table_$P.txt
  cp gpurun_out/rp/bench_$P.json profiles/${R}_bench_line_$P.json
  [ -s gpurun_out/rp/mfma_busy_$P.txt ] && cp gpurun_out/rp/mfma_busy_$P.txt profiles/${R}_mfma_busy_$P.txt
done
cp gpurun_out/rp/stage_roofline.jsonl profiles/${R}_stage_roofline_host_timed.jsonl
[ -s gpurun_out/rp/mfma_peak.txt ] && cp gpurun_out/rp/mfma_peak.txt profiles/${R}_mfma_peak.txt
if [ -d gpurun_out/rp/pf_track ]; then
  cp $(find gpurun_out/rp/pf_track -name '*kernel_stats.csv' | head -1) profiles/${R}_per_frame_kernel_stats.csv
  { echo "per-frame tracker path, 256 frames x 2 hands (tools/per_frame_prof.py plain 256; second line = measured pass):"; echo "fp32:"; cat gpurun_out/rp/per_frame_plain_fp32.txt; echo "bf16:"; cat gpurun_out/rp/per_frame_plain_bf16.txt; } > profiles/${R}_per_frame_ms.txt
fi
python3 - <<PY
import json
f=json.load(open('gpurun_out/rp/traffic_fp32.json')); b=json.load(open('gpurun_out/rp/traffic_bf16.json')); x=json.load(open('gpurun_out/rp/traffic_fp32x3.json'))
p='profiles/${R}_backbone_traffic.json'
import os
old=json.load(open(p)) if os.path.exists(p) else {"note": "HBM bytes per crop of the backbone launches (stem + convolutions) from rocprofv3 --pmc passes over tools/profile_forward.py at 16384 crops per forward call (the bench's call size): FETCH_SIZE (x2: gfx950 reports half the bytes of 16-byte-per-lane streaming reads) and WRITE_SIZE, KiB units; tools/round_profiles.sh, tools/traffic_json.py"}
old['fp32']=dict(f, algorithmic_bytes_per_crop=8396928); old['bf16']=dict(b, algorithmic_bytes_per_crop=4222080); old['fp32x3']=dict(x, algorithmic_bytes_per_crop=8396928)
old['measured_at_head']=os.environ.get('UMT_MEASURED_HEAD','unknown')  # git HEAD of the tree the PMC passes ran on (+worktree: uncommitted kernel changes on top)
json.dump(old,open(p,'w'),indent=1)
for q in ('fp32','bf16','fp32x3'):
    d=json.load(open('profiles/${R}_bench_line_%s.json'%q)); r=d['roofline']
    print(q, round(d['value']), round(d['ms_per_step'],2), 'TF', round(r['achieved'],1), round(r['frac'],4), 'hbm', round(r['hbm_view']['achieved']), round(r['hbm_view']['frac'],3), 'traffic', r['traffic'], 'pcie', round(d['pcie_inclusive']['value']), 'cpu', round(d['cpu_baseline']['value'],1), 'parity mm', d['parity_sample_max_mm'], 'rec', round(d['recurrent']['value_per_gpu']))
PY
tail -1 profiles/${R}_layer_table_fp32.txt; tail -1 profiles/${R}_layer_table_bf16.txt; tail -1 profiles/${R}_layer_table_fp32x3.txt
