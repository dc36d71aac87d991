#!/bin/bash
# development helper: compile one .hip to gfx950 assembly and list the barriers / vmcnt waits of a kernel's main loop
# usage: tools/isa_waits.sh <file.hip> <mangled-name-substring>
R=/root/repo
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -S --cuda-device-only -o /tmp/isa.s $R/qrec_amd/csrc/$1 2>&1 | grep -v warning | tail -3
python3 - "$2" <<'PY'
import sys, re
s = open('/tmp/isa.s').read()
m = re.search(r'^(_Z\w*%s\w*):' % re.escape(sys.argv[1]), s, re.M)
a = m.end(); b = s.index('.Lfunc_end', a)
lines = s[a:b].split('\n'); print(m.group(1)[:80], len(lines), "lines")
inl = False
for i, l in enumerate(lines):
    t = l.strip()
    if 'Loop Header' in t: inl = True
    if inl and any(k in t for k in ('s_barrier', 'vmcnt', 'Loop Header', 'scratch_')):
        print(i, t[:100])
k = s.index(m.group(1), b)
print(re.findall(r'\.(?:vgpr_count|vgpr_spill_count|sgpr_count):\s+\d+', s[k:k+3000])[:3])
PY
