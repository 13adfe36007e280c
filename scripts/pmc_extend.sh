#!/bin/bash
# Counter passes that say what the extend kernel waits for: vector ALU, scalar ALU, and the vector memory path
# (texture addresser TA, L1 cache TCP, texture data TD), each pass in a run of its own.
# usage: scripts/pmc_extend.sh <tag>   (RT_AMD_LIBRARY selects another build; writes gpurun_out/pmc_<tag>/)
set -u
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  PIPES=1 SPP=${SPP:-8} STEPS=1 timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scripts/profile_step.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE &&
run sq3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_CVT SQ_IFETCH GRBM_GUI_ACTIVE &&
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
run ta2 TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum &&
run td TD_TD_BUSY_sum TD_TC_STALL_sum &&
run tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum &&
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum &&
run tcp3 TCP_TOTAL_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum
echo passes done
python3 scripts/pmc_summary.py $TAG > $OUT/summary.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if "k_extend" not in k: continue
    r=v["raw"]; sec=v["seconds_profiled"]; cyc=sec*2.4e9
    print(k, "seconds", sec, "dispatches", v["dispatches"])
    for n in sorted(r):
        print(f"  {n:44s} {r[n]:16.0f}   per CU-cycle@2.4GHz {r[n]/(256*cyc):8.4f}")
PY
