#!/usr/bin/env bash
# Builds libtts_hip.so (gfx950 only) next to the Python package.  Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libtts_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$here/build"
objs=()
pids=()
for src in engine waveglow tacotron2 mel_stft; do
  obj="$here/build/$src.o"
  objs+=("$obj")
  if [[ ! -f "$obj" || "$here/$src.hip" -nt "$obj" || "$here/gemm_f32.h" -nt "$obj" || "$here/engine.h" -nt "$obj" \
        || "$here/../../include/tts_hip.h" -nt "$obj" ]]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$here/$src.hip" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
