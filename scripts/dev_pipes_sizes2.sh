#!/bin/bash
export RT_AMD_LIBRARY=$1
for cfg in "0 400 225 50 133" "0 400 225 50 160" "6 1024 1024 25 12" "8 1920 1080 25 6" "1 400 225 50 133" "3 800 800 25 16" "3 800 800 25 24" "100 1920 1080 25 6"; do
  set -- $cfg
  for p in 1 2; do
    SCENE=$1 W=$2 H=$3 DEPTH=$4 SPP=$5 RT_AMD_PIPES=$p timeout -k 10 120 python scripts/dev_call.py 2>/dev/null | tail -1 || exit 1
  done
done
