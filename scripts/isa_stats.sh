#!/bin/bash
# Register / scratch / occupancy figures of every kernel and device function (from the .s)
cd "$(dirname "$0")/../raytracing_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o /tmp/rt_kernels.s rt_kernels.hip 2>/dev/null
python3 - <<'PY'
import re
s=open('/tmp/rt_kernels.s').read()
for m in re.finditer(r'; NumVgprs: (\d+)', s):
    pre=s[:m.start()]
    lab=re.findall(r'^(_Z\S+):', pre, re.M)[-1]
    tail=s[m.start():m.start()+800]
    g=lambda k: (re.search(k+r': (\d+)',tail) or [None,'-'])[1]
    print(lab[:52].ljust(54), 'vgpr', m.group(1), 'sgpr', g('NumSgprs'), 'scratch', g('ScratchSize'), 'occ', g('Occupancy'), 'lds', g('LDSByteSize'))
PY
