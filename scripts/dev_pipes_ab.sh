#!/bin/bash
# The headline bench with several numbers of pipelines (development builds: RT_AMD_PIPES).
# usage: dev_pipes_ab.sh <library with RT_DEV_KNOBS> <pipelines> ...
lib=$1; shift
for t in "$@"; do
  out=$(RT_AMD_LIBRARY=$lib RT_AMD_PIPES=$t timeout -k 10 120 python bench.py --no-cpu-baseline --no-configs --no-live-traffic 2>/dev/null) || exit 1
  echo "$t $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["alone"]["avg_launch_ms"])')"
done
