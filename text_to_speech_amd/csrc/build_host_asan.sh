#!/usr/bin/env bash
# CPU sanitizer build of the host-only loader code (csrc/ttsw_host.h): build_host_asan/ttsw_check_asan
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/build_host_asan"
g++ -std=c++17 -g -O1 -Wall -Wextra -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer \
    "$here/host_check.cpp" -o "$here/build_host_asan/ttsw_check_asan"
echo "built $here/build_host_asan/ttsw_check_asan"
