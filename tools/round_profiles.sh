#!/bin/bash
# Runs on the GPU box (gpurun): collects the evidence that profiles/ keeps.  Everything goes to gpurun_out/rp/.
#   tools/round_profiles.sh
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/rp
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
set -e
# 1. kernel stats of the bench command itself (fp32 headline)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/bench -o b --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-hands 0 --no-pcie > $OUT/bench_prof.json 2> $OUT/bench_prof.err
echo "bench under rocprof done"
# 2. per-layer tables
for P in fp32 bf16 fp32x3; do
  timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/lt_$P -o t --output-format csv -- python3 $ROOT/tools/profile_forward.py 4096 3 $P > $OUT/lt_$P.log 2>&1
  python3 $ROOT/tools/layer_table.py $(find $OUT/lt_$P -name '*kernel_trace.csv' | head -1) 4096 > $OUT/layer_table_$P.txt
  tail -1 $OUT/layer_table_$P.txt
done
# 3. HBM traffic: separate PMC passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2), at the bench's call size
#    (16384 crops in one forward call: 1.2 GB per stage-1 tensor, far beyond the 256 MB Infinity Cache)
TC=16384
for P in fp32 bf16 fp32x3; do
  timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d $OUT/pf_$P -o f --output-format csv -- python3 $ROOT/tools/profile_forward.py $TC 1 $P > $OUT/pf_$P.log 2>&1
  timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pw_$P -o w --output-format csv -- python3 $ROOT/tools/profile_forward.py $TC 1 $P > $OUT/pw_$P.log 2>&1
  python3 $ROOT/tools/traffic_json.py $(find $OUT/pf_$P -name '*counter_collection.csv' | head -1) $(find $OUT/pw_$P -name '*counter_collection.csv' | head -1) $TC 1 > $OUT/traffic_$P.json
  cat $OUT/traffic_$P.json
done
# 3b. matrix-pipe busy fraction and clock per kernel (its own PMC pass: 2 SQ + 1 GRBM counters)
for P in fp32 bf16 fp32x3; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pm_$P -o m --output-format csv -- python3 $ROOT/tools/profile_forward.py 2048 2 $P > $OUT/pm_$P.log 2>&1
  python3 $ROOT/tools/pmc_table.py $(find $OUT/pm_$P -name '*counter_collection.csv' | head -1) > $OUT/mfma_busy_$P.txt
  head -4 $OUT/mfma_busy_$P.txt
done
# 4. the small stages, host-timed
timeout -k 10 120 python3 $ROOT/tools/stage_roofline.py > $OUT/stage_roofline.jsonl 2>/dev/null
cat $OUT/stage_roofline.jsonl | cut -c1-170
# 5. plain bench lines (no profiler attached)
timeout -k 10 300 python3 $ROOT/bench.py > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err
timeout -k 10 300 python3 $ROOT/bench.py --precision bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
timeout -k 10 300 python3 $ROOT/bench.py --precision fp32x3 > $OUT/bench_fp32x3.json 2> $OUT/bench_fp32x3.err
cut -c1-200 $OUT/bench_fp32.json; cut -c1-200 $OUT/bench_bf16.json; cut -c1-200 $OUT/bench_fp32x3.json
# 6. the bare-loop matrix-pipe probe (the sustained ceilings frac_vs_sustained refers to)
timeout -k 10 120 $ROOT/tools/bin/mfma_peak > $OUT/mfma_peak.txt 2>&1
grep -E "LDS|fp32" $OUT/mfma_peak.txt | head -4
# 7. the per-frame tracker path: kernel stats of 256 frames (2 hands, 4 crops per frame), fp32 and bf16
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/pf_track -o p --output-format csv -- python3 $ROOT/tools/per_frame_prof.py plain 256 > $OUT/per_frame_fp32.txt 2>&1
timeout -k 10 120 python3 $ROOT/tools/per_frame_prof.py plain 256 > $OUT/per_frame_plain_fp32.txt 2>/dev/null
UMT_PF_PREC=bf16 timeout -k 10 120 python3 $ROOT/tools/per_frame_prof.py plain 256 > $OUT/per_frame_plain_bf16.txt 2>/dev/null
cat $OUT/per_frame_plain_fp32.txt $OUT/per_frame_plain_bf16.txt
