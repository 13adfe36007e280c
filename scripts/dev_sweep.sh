#!/bin/bash
# usage: scripts/dev_sweep.sh "<blocks_per_cu>:<inner>,<refill>,<medium>" ...   (one timing run of scenes 8 and 6 per setting)
for cfg in "$@"; do
  export RT_AMD_EXTEND_BLOCKS_PER_CU=${cfg%%:*} RT_AMD_EXTEND_TUNING=${cfg##*:}
  echo "== $cfg"
  ONLY="8 6" timeout -k 10 120 python scripts/dev_time.py 2>&1 | grep -E "^scene" || exit 1
done
