#!/bin/bash
# A/B of several builds on one box (development aid): usage dev_ab.sh <variant .so> ...  - the headline bench with the
# library as built and with each variant in turn, three rounds
for i in $(seq 1 ${ROUNDS:-3}); do
  for lib in "" "$@"; do
    out=$(RT_AMD_LIBRARY=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-configs --no-live-traffic 2>/dev/null) || exit 1
    echo "$(basename "${lib:-current}") $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["alone"]["avg_launch_ms"])')"
  done
done
