"""Shares of the tail kernel's wave time by part of a bounce (development aid; needs a build with -DRT_SECTION_TIMES, which prints a
line per 64th wave).  usage: RT_AMD_LIBRARY=variants/librt_amd_sect.so python scripts/dev_realtime_one.py > log; python scripts/tail_parts.py log"""
import re, sys
import numpy as np
rows = [[int(x) for x in re.findall(r"\d+", l)] for l in open(sys.argv[1]) if l.startswith("tailparts ")]
a = np.array([r for r in rows if len(r) == 8], dtype=np.float64)
a = a[a[:, 1] <= 2 * np.percentile(a[:, 1], 10)]  # (waves held up by other waves' printf calls are left out)
tot = a[:, 1].sum()
print(f"{len(a)} waves, {a[:, 1].mean():.0f} cycles per wave, {a[:, 6].mean():.1f} rounds per wave, {a[:, 7].sum() / a[:, 6].sum():.1f} lanes per round, {a[:, 1].sum() / a[:, 6].sum():.0f} cycles per round")
for i, nm in enumerate(["fetch", "first leaves", "tree walk", "shade"]):
    print(f"   {nm:14s} {100 * a[:, 2 + i].sum() / tot:5.1f} %   {a[:, 2 + i].sum() / a[:, 6].sum():8.0f} cycles per round")
