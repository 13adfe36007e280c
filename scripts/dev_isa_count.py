"""Instruction mix of one kernel in the device assembly of rt_kernels.hip (development aid).
usage: python scripts/dev_isa_count.py <assembly.s> <substring of the kernel's mangled name>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r"^_Z\w+:", l) and want in l.split(":")[0]:
        start = i
    elif start is not None and l.strip().startswith(".Lfunc_end"):
        end = i
        break
body = lines[start:end]
c = Counter()
for l in body:
    t = l.strip().split()
    if t and not t[0].endswith(":") and not t[0].startswith((".", ";")):
        c[t[0]] += 1
print(lines[start][:80])
print("lines", len(body), "VALU", sum(v for k, v in c.items() if k.startswith("v_")), "SALU", sum(v for k, v in c.items() if k.startswith("s_")))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"{k:28s}{v}")
