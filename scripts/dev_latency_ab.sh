#!/bin/bash
# The latency cases (one interactive frame of the Book-2 scene at 800x800, the Book-1 configuration's call) on several builds.
# usage: dev_latency_ab.sh <variant .so> ...
for i in 1 2; do
  for lib in "$@"; do
    f=$(RT_AMD_LIBRARY=$lib COPY=0 FRAMES=200 timeout -k 10 120 python scripts/dev_realtime_one.py 2>/dev/null | tail -1) || exit 1
    c=$(RT_AMD_LIBRARY=$lib timeout -k 10 120 python scripts/dev_c1.py 2>/dev/null | tail -1) || exit 1
    echo "$(basename $lib): $f | $c"
  done
done
