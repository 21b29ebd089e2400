#!/usr/bin/env bash
# A/B timing of kernel variants on ONE box (step times differ by +-0.3 us between boxes): every library given is timed at every batch,
# interleaved, three rounds.      gpurun -- 'bash tests/tools/ab_time.sh "4096 2000 65536 300" build/libscopa_base.so scopa_amd/libscopa_hip.so'
set -euo pipefail
export SCOPA_AB_OLD_LIBRARY=1     # a baseline built from an older revision may lack entry points added since (scopa_amd/_lib.py)
SPEC=($1); shift
for round in 1 2 3; do
  for ((i = 0; i < ${#SPEC[@]}; i += 2)); do
    for LIB in "$@"; do
      SCOPA_HIP_LIBRARY="$PWD/$LIB" timeout -k 10 120 python tests/tools/time_iter.py "${SPEC[i]}" "${SPEC[i+1]}" 2> /tmp/ab_time.err | sed "s|$PWD/||" || { cat /tmp/ab_time.err >&2; exit 1; }
    done
  done
done
