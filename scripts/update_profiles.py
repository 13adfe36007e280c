"""Copies what scripts/final_profiles.sh left under gpurun_out/ into profiles/ and rewrites profiles/traffic.json from the
new PMC summary.  usage: python scripts/update_profiles.py <tag in gpurun_out, e.g. r02d> <prefix in profiles, e.g. r02_c>
[<tag of a scripts/pmc_c5.sh run>: the 10 M-sphere scene's counters, copied to <prefix>_pmc_c5_summary.json]"""
import datetime, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, dst = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
shutil.copy(f"{G}/{tag}_bench.json", f"{P}/{dst}_bench.json")
shutil.copy(f"{G}/{tag}_bench_under_rocprof.json", f"{P}/{dst}_bench_under_rocprof.json")
stats = sorted(glob.glob(f"{G}/{tag}_kstats/*/*_kernel_stats.csv"), key=os.path.getmtime)[-1]
shutil.copy(stats, f"{P}/{dst}_kernel_stats_bench.csv")
shutil.copy(f"{G}/{tag}_pmc_summary.log", f"{P}/{dst}_pmc_summary.json")
shutil.copy(f"{G}/{tag}_extend_by_bounce.txt", f"{P}/{dst}_extend_by_bounce.txt")
d = json.load(open(f"{P}/{dst}_pmc_summary.json"))
t = json.load(open(f"{P}/traffic.json"))
e = next(v for k, v in d.items() if k.startswith("rt::k_extend<false"))
segments = t["pipeline"]["segments"]  # of 16 spp of the headline scene: the PMC runs render the same samples
src = f"profiles/{dst}_pmc_summary.json"
t["extend"].update({"source": f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, one pipeline, 16 spp of the 4K headline scene; FETCH_SIZE doubled as on gfx950)",
                    "hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "fetch_bytes": e["fetch_bytes"], "write_bytes": e["write_bytes"], "dispatches": e["dispatches"]})
hbm = sum(v.get("fetch_bytes", 0) + v.get("write_bytes", 0) for v in d.values())
t["pipeline"].update({"hbm_bytes_per_segment": round(hbm / segments, 1), "hbm_bytes_per_16spp_step": hbm})
valu = sum(v["raw"].get("SQ_ACTIVE_INST_VALU", 0) for v in d.values())
secs = sum(v["seconds_profiled"] for v in d.values())
t["extend_issue"].update({"source": f"{src} (SQ passes, one pipeline, 16 spp of the 4K headline scene)",
                          "valu_issue_share": e["valu_issue_share_at_2p4GHz"], "active_lanes_per_valu_inst": e["active_lanes_per_valu_inst"],
                          "wait_share_of_wave_cycles": e["wait_any_share_of_wave_cycles"],
                          "valu_insts_per_segment_wave": round(e["raw"]["SQ_INSTS_VALU"] / (segments / 64), 1),
                          "whole_step_valu_issue_share": round(valu * 4 / (1024 * 2.4e9 * secs), 3)})
if len(sys.argv) > 3:  # the 10 M-sphere scene (scripts/pmc_c5.sh): memory-side read traffic of the extend kernel
    shutil.copy(f"{G}/pmc_{sys.argv[3]}/summary.json", f"{P}/{dst}_pmc_c5_summary.json")
    c5 = next(v for k, v in json.load(open(f"{P}/{dst}_pmc_c5_summary.json")).items() if k.startswith("rt::k_extend<false"))
    r = c5["raw"]
    read_bytes = r["TCC_EA0_RDREQ_128B_sum"] * 128 + r["TCC_EA0_RDREQ_64B_sum"] * 64 + r["TCC_EA0_RDREQ_32B_sum"] * 32
    t["c5"] = {"source": f"profiles/{dst}_pmc_c5_summary.json (10 M spheres 1920x1080, depth 2, 3 calls of 16 spp; scripts/pmc_c5.sh)",
               "kernel": "k_extend", "seconds": c5["seconds_profiled"], "dispatches": c5["dispatches"],
               "memory_side_read_bytes": read_bytes, "memory_side_gbs": round(read_bytes / c5["seconds_profiled"] / 1e9, 1),
               "l2_hit_rate": round(r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"]), 4),
               "note": "read requests of the L2 to the fabric (TCC_EA0_RDREQ by size, Infinity Cache hits included) over the "
                       "extend launches' time: what the node and sphere gathers pull in; write traffic (hit records) is 1 % of it"}
# which code the numbers belong to: the commit the working tree was at when they were copied in, and the day
t["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
t["date"] = datetime.date.today().isoformat()
json.dump(t, open(f"{P}/traffic.json", "w"), indent=1)
b = json.load(open(f"{P}/{dst}_bench.json"))
print("bench", b["value"], "Msamples/s; extend", b["roofline"]["avg_launch_ms"], "ms per launch, alone", b["roofline"]["alone"]["avg_launch_ms"])
print("extend (PMC run):", e["seconds_profiled"] * 1e3, "ms,", e["raw"]["SQ_INSTS_VALU"] / 1e9, "G VALU,", e["raw"]["SQ_INSTS_SALU"] / 1e9, "G SALU;", t["extend_issue"])
for line in open(f"{P}/{dst}_kernel_stats_bench.csv"):
    if "k_extend<false" in line:
        print("kernel trace:", line.split('",')[1:4])
print("under rocprof:", json.load(open(f"{P}/{dst}_bench_under_rocprof.json"))["roofline"]["avg_launch_ms"])
