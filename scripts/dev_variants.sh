#!/bin/bash
# usage: scripts/dev_variants.sh <variant>:<blocks_per_cu> ...   (timing of scenes 8 and 6 with variants/librt_amd_<variant>.so)
for cfg in "$@"; do
  export RT_AMD_LIBRARY=$PWD/variants/librt_amd_${cfg%%:*}.so RT_AMD_EXTEND_BLOCKS_PER_CU=${cfg##*:}
  echo "== $cfg"
  ONLY="8 6" timeout -k 10 120 python scripts/dev_time.py 2>&1 | grep -E "^scene" || exit 1
done
