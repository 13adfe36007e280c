"""Summarises the CSVs written by scripts/pmc_passes.sh: per kernel, counter sums over all
dispatches and the derived figures used in the roofline discussion."""
import csv, glob, json, os, sys
from collections import defaultdict
tag = sys.argv[1]
base = f"gpurun_out/pmc_{tag}"
tot = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(int)
dur = defaultdict(float)
for f in glob.glob(f"{base}/*/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(f"{base}/sq1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        ndisp[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out = {}
for k, c in tot.items():
    if not k.startswith("rt::"): continue
    d = dict(c)
    g = d.get("GRBM_GUI_ACTIVE", 0)
    res = {"dispatches": ndisp.get(k), "seconds_profiled": round(dur.get(k, 0), 6)}
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3; SQ_* are summed over all SEs, quad-cycles for ACTIVE_INST
    if g:
        cycles = g / 8.0
        res["valu_busy_pct"] = round(100 * d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024.0 * cycles), 2)   # 1024 SIMDs
        res["occupancy_waves_per_simd"] = round(d.get("SQ_WAVE_CYCLES", 0) * 4 / (1024.0 * cycles), 2)
        res["wait_any_share_of_wave_cycles"] = round(d.get("SQ_WAIT_ANY", 0) / max(1.0, d.get("SQ_WAVE_CYCLES", 1)), 3)
        res["valu_insts_per_wave"] = round(d.get("SQ_INSTS_VALU", 0) / max(1.0, d.get("SQ_WAVES", 1)), 1)
    if d.get("SQ_THREAD_CYCLES_VALU") and d.get("SQ_ACTIVE_INST_VALU"):
        res["active_lanes_per_valu_inst"] = round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"], 2)
    if dur.get(k):
        # wall-clock based: assumes 2.4 GHz; VALU busy = issue cycles / (1024 SIMDs x cycles)
        cyc = dur[k] * 2.4e9
        res["valu_issue_share_at_2p4GHz"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024.0 * cyc), 3)
        # the unambiguous form: vector instructions per SIMD and cycle (a SIMD retires at most one per 2.4 cycles of the
        # cheap opcodes - add, mul, mov, logic - and one per 4.2 of fma, min / max, compares, conversions, packed ones:
        # scripts/micro/valu_issue.hip)
        res["valu_insts_per_simd_cycle_at_2p4GHz"] = round(d.get("SQ_INSTS_VALU", 0) / (1024.0 * cyc), 4)
        res["waves_per_simd_at_2p4GHz"] = round(d.get("SQ_WAVE_CYCLES", 0) * 4 / (1024.0 * cyc), 2)
    if d.get("TCC_HIT_sum") is not None:
        h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
        res["l2_hit_rate"] = round(h / max(1.0, h + m), 4)
    if "FETCH_SIZE" in d: res["fetch_bytes"] = d["FETCH_SIZE"] * 1024 * 2   # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    if "WRITE_SIZE" in d: res["write_bytes"] = d["WRITE_SIZE"] * 1024
    if "fetch_bytes" in res and "write_bytes" in res and dur.get(k):
        res["hbm_bytes_per_launch"] = round((res["fetch_bytes"] + res["write_bytes"]) / max(1, ndisp.get(k, 1)), 1)
        res["hbm_gbs"] = round((res["fetch_bytes"] + res["write_bytes"]) / dur[k] / 1e9, 1)
    res["raw"] = {n: v for n, v in sorted(d.items())}
    out[k] = res
print(json.dumps(out, indent=1))
