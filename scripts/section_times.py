"""Shares of the extend kernel's wave time by part of its loop (development aid; needs a build with -DRT_SECTION_TIMES, which
prints a line per 64th wave).  usage: RT_AMD_LIBRARY=variants/librt_amd_sect.so PIPES=1 SPP=16 STEPS=1 python scripts/profile_step.py > log;
python scripts/section_times.py log"""
import re, sys
from collections import defaultdict
acc = defaultdict(lambda: [0] * 12)
rows = [[int(x) for x in re.findall(r"\d+", l)] for l in open(sys.argv[1]) if l.startswith("sections ")]
rows = [v for v in rows if len(v) == 13]
# waves that were held up by other waves' printf calls (tens of times slower than the rest of their launch) are left out
low = {}
for v in rows: low.setdefault(v[0], []).append(v[2])
low = {b: sorted(t)[len(t) // 10] for b, t in low.items()}
for v in rows:
    b = v[0]
    if v[2] > 2 * low[b]: continue
    for key in (b if b < 4 else "4+", "all"):
        a = acc[key]
        a[0] += 1
        for i, x in enumerate(v[2:13]): a[1 + i] += x
names = ["fetch / retire / decisions", "node loop", "sphere leaves", "quad leaves", "media"]
for key, a in sorted(acc.items(), key=lambda kv: str(kv[0])):
    total = a[1]
    print(f"bounce {key}: {a[0]} waves, {total / a[0]:.0f} cycles per wave")
    for i, nm in enumerate(names):
        t, n = a[2 + 2 * i], a[3 + 2 * i]
        print(f"   {nm:28s} {100.0 * t / total:5.1f} %   {n / a[0]:8.1f} runs per wave   {t / max(n, 1):8.0f} cycles per run")
