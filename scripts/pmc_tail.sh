#!/bin/bash
# SQ counters of the tail kernel on a tiny frame (one ray per wave): instructions and cycles per wave.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_tail
mkdir -p $OUT
run() { local name=$1; shift
  RT_TAIL=0 SCENE=${SCENE:-8} W=${W:-64} H=${H:-48} SPP=1 timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scripts/profile_scene.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS &&
run sq3 SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
python3 - <<'PY'
import csv, glob
from collections import defaultdict
tot = defaultdict(float)
for p in ("sq1", "sq2", "sq3"):
    for f in glob.glob(f"gpurun_out/pmc_tail/{p}/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "k_tail" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        last = ids[-1] if ids else None
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print(f"{k:28s} {v:14.0f}")
PY
