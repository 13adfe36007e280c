#!/bin/bash
# All bench configurations with the extend kernel's grid for two pipelines set to the given numbers of blocks (development builds).
# usage: dev_blocks_configs.sh <library with RT_DEV_KNOBS> <blocks in the grid> ...
lib=$1; shift
for t in "$@"; do
  RT_AMD_LIBRARY=$lib RT_AMD_EXTEND_BLOCKS_SHARED=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value'], {k[:2]:(v['msamples_per_s'], v['ms_per_call']) for k,v in d['configs'].items()})" || exit 1
done
