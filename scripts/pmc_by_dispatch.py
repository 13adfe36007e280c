"""Per-dispatch view of the SQ passes written by scripts/pmc_passes.sh for one kernel:
duration, resident waves per SIMD, issue shares, active lanes.  usage: pmc_by_dispatch.py <tag> [kernel substring]"""
import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_extend"
CLOCK = 2.4e9
def load(p):
    f = glob.glob(f"gpurun_out/pmc_{tag}/{p}/*/*counter_collection.csv")[0]
    d = defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    return d
def dur(p):
    f = glob.glob(f"gpurun_out/pmc_{tag}/{p}/*/*kernel_trace.csv")[0]
    return {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]}
a, du, b = load("sq1"), dur("sq1"), load("sq2")
ids, idb = sorted(a), sorted(b)
if len(ids) > 25:                   # two renders profiled: keep the second
    half = len(ids) // 2
    ids, idb = ids[half:], idb[half:]
print("  # dur_us waves/simd valu_issue any_issue lanes valu_M salu_M vmem_M lds_M wait%")
tot = 0.0
for i, (k, kb) in enumerate(zip(ids, idb)):
    c, cb = a[k], b[kb]
    cyc = du[k] * 1e-6 * CLOCK
    tot += du[k]
    print(f"{i:3d} {du[k]:8.1f} {c['SQ_WAVE_CYCLES']*4/(1024*cyc):6.2f} {c['SQ_ACTIVE_INST_VALU']*4/(1024*cyc):8.3f} "
          f"{c['SQ_ACTIVE_INST_ANY']*4/(1024*cyc):8.3f} {cb['SQ_THREAD_CYCLES_VALU']/max(1,c['SQ_ACTIVE_INST_VALU']):6.1f} "
          f"{c['SQ_INSTS_VALU']/1e6:8.1f} {cb['SQ_INSTS_SALU']/1e6:7.1f} {(cb['SQ_INSTS_VMEM_RD']+cb['SQ_INSTS_VMEM_WR'])/1e6:6.1f} {cb['SQ_INSTS_LDS']/1e6:6.1f} "
          f"{100*c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:5.1f}")
print(f"total {tot/1e3:.2f} ms")
