#!/usr/bin/env bash
# Builds libtts_hip.so (gfx950 only) next to the Python package.  Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# TTS_BUILD_TAG=<tag> builds a separate variant (objects in build_<tag>/, library libtts_hip_<tag>.so), e.g. the
# measurement build `TTS_BUILD_TAG=dbg csrc/build.sh -DTTS_DEBUG_HOOKS` that scripts/persist_probe.py loads through
# TTS_HIP_LIBRARY; the default library never contains debug hooks.
tag="${TTS_BUILD_TAG:-}"
out="$here/../libtts_hip${tag:+_$tag}.so"
bdir="$here/build${tag:+_$tag}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$bdir"
objs=()
pids=()
for src in engine waveglow wn_wino tacotron2 taco_persist taco_fused mel_stft; do
  obj="$bdir/$src.o"
  objs+=("$obj")
  if [[ ! -f "$obj" || "$here/$src.hip" -nt "$obj" || "$here/gemm_f32.h" -nt "$obj" || "$here/engine.h" -nt "$obj" \
        || "$here/taco_persist.h" -nt "$obj" || "$here/ttsw_host.h" -nt "$obj" || "$here/xch_util.h" -nt "$obj" || "$here/taco_fused.h" -nt "$obj" \
        || "$here/../../include/tts_hip.h" -nt "$obj" ]]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$here/$src.hip" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
