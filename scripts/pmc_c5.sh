#!/bin/bash
# Counter passes for the 10 M-sphere scene: what the memory side does under the extend kernel's node gathers.
# usage: scripts/pmc_c5.sh <tag>   (RT_AMD_LIBRARY selects another build; writes gpurun_out/pmc_<tag>/)
set -u
TAG=${1:-c5}
export TMPDIR=/tmp SCENE=102 W=1920 H=1080 SPP=16 DEPTH=2 STRESS=${STRESS:-10000000}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { local name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scripts/profile_scene.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum &&
run tcc2 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_REQ_sum &&
run tcc3 TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_READ_REQ_LATENCY_sum &&
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
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
