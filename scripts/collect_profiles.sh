#!/usr/bin/env bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh <tag>        e.g. r03
# Outputs land in gpurun_out/prof_<tag>/ ; scripts/summarize_profiles.py turns them into the files kept under profiles/.
# Every profiled program is started directly after `--` (python3 <script>); counters are collected in passes of their own.
set -euo pipefail
tag="${1:-r03}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py"
PMC3="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
# 1. kernel trace + stats of the bench command (fp32 headline), then its PMC passes (separate runs, no trace flags;
#    FETCH_SIZE and WRITE_SIZE do not fit one pass: "Request exceeds the capabilities of the hardware to collect")
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
    python3 "$B" --steps 2 --warmup 1 --cpu-frames 0 --no-extra --no-config4 > "$out/bench_under_rocprof.json" 2> "$out/stats.log"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "$PMC3"; do
    name="${pmc%% *}"
    timeout -k 10 400 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc_$name" -o bench -- \
        python3 "$B" --steps 1 --warmup 0 --cpu-frames 0 --no-extra --no-config4 --no-kernel-timing > /dev/null 2> "$out/pmc_$name.log"
done
# 2. the fp16 / split-fp16 WaveGlow kernels: kernel stats and the same three PMC passes
for prec in f16 f16x3; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$prec" -o "$prec" -- \
        python3 "$root/scripts/x3_time.py" $prec > "$out/${prec}_time.txt" 2> "$out/stats_$prec.log"
    for pmc in "FETCH_SIZE" "WRITE_SIZE" "$PMC3"; do
        name="${pmc%% *}"
        timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc_${prec}_$name" -o "$prec" -- \
            python3 "$root/scripts/x3_time.py" $prec > /dev/null 2> "$out/pmc_${prec}_$name.log"
    done
done
# 3. Tacotron2 decoder, every machine at the batch sizes it serves: kernel stats ...
for spec in "1 256 f32 persistent" "2 256 f32 persistent" "4 256 f32 fused" "8 256 f32 fused" "8 256 f16 fused" "8 256 f32 graph" "1 256 f32 graph"; do
    set -- $spec
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/taco_b$1_$3_$4" -o taco -- \
        python3 "$root/scripts/run_taco.py" $1 $2 $3 $4 > "$out/taco_b$1_$3_$4.txt" 2> "$out/taco_b$1_$3_$4.log"
done
# ... and the bytes they fetch (persistent: weights once per CALL; fused / graph: once per STEP)
for spec in "1 64 f32 persistent" "8 64 f32 fused" "8 64 f32 graph"; do
    set -- $spec
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/taco_pmc_b$1_$4" -o taco -- \
        python3 "$root/scripts/run_taco.py" $1 $2 $3 $4 > /dev/null 2> "$out/taco_pmc_b$1_$4.log"
done
find "$out" -name "*.csv" | sort
