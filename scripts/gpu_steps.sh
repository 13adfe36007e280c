#!/bin/bash
# Runs GPU steps one after the other, each under its own time limit; stops at the first step that ends in a fault, an
# abort, a kill or a time limit (a failed assertion of a test run - exit status 1 - does not stop the sequence).
# usage: scripts/gpu_steps.sh "<seconds> <log name> <command...>" ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for spec in "$@"; do
  set -- $spec
  limit=$1; log=$2; shift 2
  echo "== step: $* (limit ${limit}s, log gpurun_out/$log)"
  timeout -k 10 "$limit" "$@" > "gpurun_out/$log" 2>&1
  rc=$?
  echo "== exit status $rc"
  tail -n 6 "gpurun_out/$log"
  case $rc in
    0|1) ;;
    *) echo "== stopping: step ended with status $rc"; exit $rc ;;
  esac
done
exit 0
