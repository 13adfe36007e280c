#!/bin/bash
# The 10 M-sphere scene on several builds and grid sizes (development aid; needs builds with RT_DEV_KNOBS).
# usage: c5_ab.sh <lib>[:blocks per CU] ...
for spec in "$@"; do
  lib=${spec%%:*}; per=${spec#*:}; [ "$per" = "$spec" ] && per=
  echo "== $lib blocks/CU ${per:-default}"
  RT_AMD_LIBRARY=$lib RT_AMD_EXTEND_BLOCKS_PER_CU=$per timeout -k 10 200 python scripts/dev_configs.py c5 || exit 1
done
