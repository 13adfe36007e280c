#!/bin/bash
# One against two pipelines by job size (development builds): usage dev_pipes_sizes.sh <library with RT_DEV_KNOBS>
export RT_AMD_LIBRARY=$1
for cfg in "0 400 225 50 100" "0 400 225 50 200" "0 400 225 50 400" "0 400 225 50 800" "6 1024 1024 25 8" "6 1024 1024 25 16" "6 1024 1024 25 32" "6 1024 1024 25 64" "8 1920 1080 25 4" "8 1920 1080 25 8" "8 1920 1080 25 16" "8 1920 1080 25 32"; do
  set -- $cfg
  for p in 1 2; do
    SCENE=$1 W=$2 H=$3 DEPTH=$4 SPP=$5 RT_AMD_PIPES=$p timeout -k 10 120 python scripts/dev_call.py 2>/dev/null | tail -1 || exit 1
  done
done
