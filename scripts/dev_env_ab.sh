#!/bin/bash
# The headline bench for several values of one RT_AMD_* environment variable (development builds).
# usage: dev_env_ab.sh <library with RT_DEV_KNOBS> <variable> <value> ...   ("-" as a value: variable unset)
lib=$1; var=$2; shift 2
for t in "$@"; do
  if [ "$t" = "-" ]; then unset $var; else export $var=$t; fi
  out=$(RT_AMD_LIBRARY=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-configs --no-live-traffic 2>/dev/null) || exit 1
  echo "$var=$t $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["alone"]["avg_launch_ms"], d["roofline"]["phase_ms"]["shade"])')"
done
