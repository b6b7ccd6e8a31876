"""List vector-memory instructions, vmcnt waits and loop headers of one kernel in a hipcc -S listing.
usage: isa_waits.py listing.s kernel-name-substring"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    if sys.argv[2] in m.group(1):
        b = m.group(2)
        print(m.group(1)[:60], 'vgpr', re.search(r'next_free_vgpr (\d+)', b).group(1), 'scratch',
              re.search(r'private_segment_fixed_size (\d+)', b).group(1))
names = [m.group(1) for m in re.finditer(r'\.amdhsa_kernel (\S+)', s) if sys.argv[2] in m.group(1)]
for n in names[:1]:
    body = s[s.index('\n' + n + ':'):]
    body = body[:body.index('.end_amdhsa_kernel')].split('\n')
    for i, l in enumerate(body):
        if 'vmcnt' in l or 'global_store' in l or 'global_load' in l or 'Loop Header' in l or 's_barrier' in l:
            print(i, l.strip()[:100])
