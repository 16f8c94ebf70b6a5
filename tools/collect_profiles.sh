#!/bin/bash
# Round evidence, collected on the GPU box in one go (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02
# writes gpurun_out/prof_<round>/ : bench lines (default, sustained clock, causal, configs 2 / 3), rocprofv3 kernel
# stats of the bench command, PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters; counters in their own runs, never
# combined with trace domains other than --kernel-trace), the fp8 bench and one sweep of the reference-shaped harness.
set -u
rm -rf "$PWD/gpurun_out/prof_${1:-r02}"   # a second collection would mix its run directories with the first one's
R=${1:-r02}
OUT=$PWD/gpurun_out/prof_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py --cpu-seconds 0"
echo "== bench lines"; date
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"     # the driver's command line
$B > "$OUT/bench_default_300_steps.json" 2>> "$OUT/bench.err"                                # bench.py's own defaults: 300 steps
$B --steps 20 --warmup 5 --causal > "$OUT/bench_causal.json" 2>> "$OUT/bench.err"
$B --steps 20 --warmup 5 --batch 8 --heads 16 --seqlen 2048 --head-dim 64 > "$OUT/bench_config2.json" 2>> "$OUT/bench.err"
$B --steps 20 --warmup 5 --batch 4 --heads 32 --seqlen 8192 --causal > "$OUT/bench_config3.json" 2>> "$OUT/bench.err"
$B --steps 20 --warmup 5 --dtype fp16 > "$OUT/bench_fp16.json" 2>> "$OUT/bench.err"
echo "== kernel stats"; date
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -- python3 "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 > "$OUT/ks.log" 2>&1)
echo "== pmc"; date
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 "$OLDPWD/bench.py" --steps 3 --warmup 1 --settle-seconds 0 --cpu-seconds 0 > "$OUT/pmc_$C.log" 2>&1)
done
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$OLDPWD/bench.py" --steps 3 --warmup 1 --settle-seconds 0 --cpu-seconds 0 > "$OUT/pmc_sq.log" 2>&1)
echo "== fp8 + harness"; date
python3 tools/bench_fp8.py > "$OUT/fp8_forward_config5.json" 2> "$OUT/fp8.err"
(cd benchmarks && python3 bench_compare_all.py --seqlen 512 2048 8192 --head-dim 64 128 --batch-size 2 --num-heads 4 --dtypes bf16 --iters 10 --warmup 3 --fp8 --directions forward backward --no-plot --tag "$R" > "$OUT/harness_table.txt" 2> "$OUT/harness.err")
cp benchmarks/results/*"$R"*.json "$OUT/" 2>/dev/null
python3 tools/w4_cycles.py --kernel dkdv > "$OUT/cycles_dkdv.md" 2>/dev/null
python3 tools/w4_cycles.py --kernel dq > "$OUT/cycles_dq.md" 2>/dev/null
python3 tools/ds_store_cycles.py > "$OUT/diag_cycles.md" 2>/dev/null
python3 tools/bwd_variant_sweep.py --causal > "$OUT/bwd_variants_causal.md" 2>/dev/null
python3 tools/bwd_variant_sweep.py > "$OUT/bwd_variants.md" 2>/dev/null
python3 tools/collect_traffic.py "$OUT/traffic.json" $(find "$OUT/pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1) $(find "$OUT/pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1) > /dev/null 2>> "$OUT/bench.err"
python3 tools/pmc_summary.py $(find "$OUT/pmc_sq" -name "*counter_collection.csv" | head -1) > "$OUT/pmc_sq_counters.txt" 2>> "$OUT/bench.err"
cp $(find "$OUT/ks" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT" -name "*.csv" | head -40
echo "== done"; date
