#!/bin/bash
# The headline bench with the extend kernel's grid limited to fewer blocks (development builds).
# usage: dev_blocks_ab.sh <library with RT_DEV_KNOBS> <blocks in the grid> ...
lib=$1; shift
for t in "$@"; do
  out=$(RT_AMD_LIBRARY=$lib RT_AMD_EXTEND_BLOCKS_SHARED=$t timeout -k 10 120 python bench.py --no-cpu-baseline --no-configs --no-live-traffic 2>/dev/null) || exit 1
  echo "$t $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["alone"]["avg_launch_ms"])')"
done
