#!/bin/bash
# Collects the PMC passes used for the roofline discussion, each in a run of its own.
# usage: scripts/pmc_passes.sh <tag>    (writes gpurun_out/pmc_<tag>/passN/)
set -u
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  PIPES=1 SPP=${SPP:-8} STEPS=1 timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scripts/profile_step.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
echo passes done
