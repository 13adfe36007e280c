#!/bin/bash
# The Book-1 configuration's time per call for several values of one RT_AMD_* environment variable (development builds).
# usage: dev_c1_env.sh <library with RT_DEV_KNOBS> <variable> <value> ...   ("-" as a value: variable unset)
lib=$1; var=$2; shift 2
for t in "$@"; do
  if [ "$t" = "-" ]; then unset $var; else export $var=$t; fi
  echo "$var=$t $(RT_AMD_LIBRARY=$lib timeout -k 10 120 python scripts/dev_c1.py 2>/dev/null | tail -1)" || exit 1
done
