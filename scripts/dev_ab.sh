#!/bin/bash
# A/B of two builds on one box: usage ab.sh <variant .so> (runs bench alternately, 3 times each)
for i in 1 2 3; do
  for lib in "" "$1"; do
    out=$(RT_AMD_LIBRARY=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-configs 2>/dev/null) || exit 1
    echo "${lib:-current} $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["alone"]["avg_launch_ms"])')"
  done
done
