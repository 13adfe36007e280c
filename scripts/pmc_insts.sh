#!/bin/bash
# One counter pass: what the kernels of a 16-spp batch of the headline scene execute, by instruction kind.
# usage: scripts/pmc_insts.sh <tag>   (RT_AMD_LIBRARY selects another build; writes gpurun_out/pmc_<tag>/)
set -u
TAG=${1:-insts}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
PIPES=1 SPP=${SPP:-16} STEPS=1 timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq1 -- python3 scripts/profile_step.py > $OUT/sq1.log 2>&1 || { echo "pass failed"; tail -5 $OUT/sq1.log; exit 1; }
python3 scripts/pmc_summary.py $TAG > $OUT/summary.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    r=v["raw"]
    turns=r.get("SQ_INSTS_VMEM_RD",0)/4 if "k_extend" in k else 0
    print(f"{k[:44]:44s} {v['seconds_profiled']*1e3:8.3f} ms  VALU {r.get('SQ_INSTS_VALU',0)/1e9:7.3f} G  SALU {r.get('SQ_INSTS_SALU',0)/1e9:7.3f} G  BRANCH {r.get('SQ_INSTS_BRANCH',0)/1e9:6.3f} G  VMEM_RD {r.get('SQ_INSTS_VMEM_RD',0)/1e9:6.3f} G  LDS {r.get('SQ_INSTS_LDS',0)/1e9:6.3f} G  SMEM {r.get('SQ_INSTS_SMEM',0)/1e9:6.3f} G")
PY
