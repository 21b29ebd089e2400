#!/usr/bin/env bash
# Host-side sanitizer pass over the product library's host glue (scopa_host.hip state functions, scopa_rules.h, scopa_full_rules.h,
# scopa_team_rules.h, scopa_mt.h ...).  CPU build box only: the device code objects are compiled as usual (-fno-gpu-sanitize; GPU
# sanitizers are not available on the pool), the HOST side of every .hip file is built with AddressSanitizer + UBSan, and the
# non-GPU tests that drive the host entry points run through that build.  Any report aborts the run (halt_on_error=1).
#     bash tests/tools/sanitize_host.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/build/asan"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" -O1 -g --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
    -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wall -Wno-unused-function \
    -o "$OUT/libscopa_hip.so" "$ROOT"/scopa_amd/csrc/*.hip
RT="$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)"
test "$(nm -D "$OUT/libscopa_hip.so" | grep -c '__asan_\|__ubsan_')" -gt 0
cd "$ROOT"
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    SCOPA_HIP_LIBRARY="$OUT/libscopa_hip.so" \
    python -m pytest tests/test_abi_host.py tests/test_full_scopa.py tests/test_team_mini_scopa.py tests/test_env_mirror.py -m "not gpu" -x -q
