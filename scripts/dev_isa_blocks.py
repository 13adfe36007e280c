"""Basic blocks of one kernel in an assembly dump, in order, with what each holds (development aid).
usage: python scripts/dev_isa_blocks.py file.s <substring of the kernel's mangled name>"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and ":" in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], {"label": "entry", "ops": []}
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t[0] in ";." and not re.match(r"\.LBB\d+_\d+:", t): continue
    m = re.match(r"(\.LBB\d+_\d+):", t)
    if m:
        blocks.append(cur); cur = {"label": m.group(1), "ops": []}; continue
    cur["ops"].append(t)
blocks.append(cur)
tot = {}
for b in blocks:
    c = {"v": 0, "s": 0, "mem": 0, "ds": 0}
    flags = set()
    for t in b["ops"]:
        op = t.split()[0]
        k = "v" if op.startswith("v_") else "s" if op.startswith("s_") else "ds" if op.startswith("ds_") else "mem"
        c[k] += 1
        for key in ("sqrt", "div_", "cvt_f32_ubyte", "rcp", "memtime", "scratch", "atomic", "global_store", "readfirstlane", "bcnt"):
            if key in op: flags.add(key)
        if op.startswith("s_cbranch") or op == "s_branch": flags.add(op.replace("s_cbranch_", "br:").replace("s_branch", "jmp") + ">" + t.split()[-1])
    for k in c: tot[k] = tot.get(k, 0) + c[k]
    print(f"{b['label']:12s} v {c['v']:4d} s {c['s']:4d} mem {c['mem']:3d} ds {c['ds']:3d}  {' '.join(sorted(flags))}")
print("total", tot)
