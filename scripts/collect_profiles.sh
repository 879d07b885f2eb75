#!/usr/bin/env bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh <tag>        e.g. r01b
# Outputs land in gpurun_out/prof_<tag>/ ; scripts/summarize_profiles.py turns them into the files kept under profiles/.
set -euo pipefail
tag="${1:-r01}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py"
if [[ -z "${SKIP_BENCH:-}" ]]; then
# 1. kernel trace + stats of the bench command (fp32 headline; the fp16 extra is profiled separately below)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
    python3 "$B" --steps 2 --warmup 1 --cpu-frames 0 --no-extra > "$out/bench_under_rocprof.json" 2> "$out/stats.log"
# 2. PMC passes (separate runs, no trace flags)
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
    name="$(echo "$pmc" | cut -d' ' -f1)"
    timeout -k 10 400 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc_$name" -o bench -- \
        python3 "$B" --steps 1 --warmup 0 --cpu-frames 0 --no-extra --no-kernel-timing > /dev/null 2> "$out/pmc_$name.log"
done
fi
# 2b. the fp16 / split-fp16 in-layer kernels: kernel stats and one PMC pass each (bytes, MFMA busy, active clock)
for prec in f16 f16x3; do
    [[ -n "${SKIP_X3:-}" ]] && continue
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$prec" -o "$prec" -- \
        python3 "$root/scripts/x3_time.py" $prec > "$out/${prec}_time.txt" 2> "$out/stats_$prec.log"
    # (FETCH_SIZE and WRITE_SIZE do not fit one pass: "Request exceeds the capabilities of the hardware to collect")
    for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
        name="$(echo "$pmc" | cut -d' ' -f1)"
        timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc_${prec}_$name" -o "$prec" -- \
            python3 "$root/scripts/x3_time.py" $prec > /dev/null 2> "$out/pmc_${prec}_$name.log" || echo "pmc pass $prec $name failed"
    done
done
if [[ -z "${WITH_LEGACY_F16:-}" ]]; then
# 4. Tacotron2 decoder: batch 1 = persistent weight-stationary kernel, batch 8 = per-step graph; plus batch 1 on the graph path
for spec in "1 256 f32 persistent" "8 256 f32 persistent" "1 256 f32 graph"; do
    set -- $spec
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/taco_b$1_$4" -o taco -- \
        python3 "$root/scripts/run_taco.py" $1 $2 $3 $4 > "$out/taco_b$1_$4.txt" 2> "$out/taco_b$1_$4.log"
done
# 5. HBM/L2 fetch bytes of the decoder kernels, batch 1 (persistent: weights are read once per CALL) and batch 8 (graph)
for spec in "1 64 f32 persistent" "8 64 f32 persistent"; do
    set -- $spec
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/taco_pmc_b$1" -o taco -- \
        python3 "$root/scripts/run_taco.py" $1 $2 $3 $4 > /dev/null 2> "$out/taco_pmc_b$1.log"
done
find "$out" -name "*.csv" | sort
exit 0
fi
# 3. fp16 mode kernel stats (same shape)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_f16" -o f16 -- \
    python3 "$root/scripts/f16_time.py" f16 > "$out/f16_time.txt" 2> "$out/stats_f16.log"
# 4. Tacotron2 decoder kernels, batch 1 and 8
for b in 1 8; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/taco_b$b" -o taco -- \
        python3 "$root/scripts/run_taco.py" $b 256 > "$out/taco_b$b.txt" 2> "$out/taco_b$b.log"
done
# 5. HBM/L2 fetch bytes of the decoder-step kernels (weight streaming), batch 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/taco_pmc_b1" -o taco -- \
    python3 "$root/scripts/run_taco.py" 1 64 > /dev/null 2> "$out/taco_pmc_b1.log"
find "$out" -name "*.csv" | sort
