"""Instruction-mix summary of the kernels in a hipcc -S --cuda-device-only assembly file (memory/LDS/barrier ops and
the resource metadata), used while tuning.  usage: python tools/isa_mix.py file.s [name-substring ...]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
want = sys.argv[2:]
names = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z[0-9A-Za-z_]+):', s, flags=re.M)]
for (pos, name), nxt in zip(names, names[1:] + [(len(s), None)]):
    if want and not any(w in name for w in want):
        continue
    body = s[pos:nxt[0]].split('.Lfunc_end')[0]
    ins = re.findall(r'^\s+([a-z][a-z_0-9]+)\s', body, flags=re.M)
    c = Counter(ins)
    keys = sorted(k for k in c if k.startswith(('ds_', 'flat_', 'global_', 'buffer_', 's_barrier', 'scratch_', 's_waitcnt')))
    md = re.search(r'\.name:\s+' + re.escape(name) + r'.*?\.vgpr_count:\s+(\d+)', s, flags=re.S)
    print(f"{name}: {len(ins)} instr")
    print("   ", {k: c[k] for k in keys})
