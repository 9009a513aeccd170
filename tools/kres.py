#!/usr/bin/env python3
"""usage: tools/kres.py <gfx950 .s file> [substring...]  -- VGPRs / scratch / LDS of every kernel in a --save-temps assembly
(the numbers the code object carries; occupancy = 512 // vgprs waves per SIMD)."""
import re
import subprocess
import sys

path, pats = sys.argv[1], sys.argv[2:]
name, info = None, {}
for l in open(path):
    m = re.match(r'\s*\.amdhsa_kernel (\S+)', l)
    if m:
        name = m.group(1)
        info[name] = {}
    if name:
        for k, short in (('.amdhsa_next_free_vgpr', 'vgpr'), ('.amdhsa_private_segment_fixed_size', 'scratch'),
                         ('.amdhsa_group_segment_fixed_size', 'lds')):
            m = re.match(r'\s*' + re.escape(k) + r' (\d+)', l)
            if m:
                info[name][short] = int(m.group(1))
    if '.end_amdhsa_kernel' in l:
        name = None
names = [n for n in info if not pats or any(p in n for p in pats)]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for n, d in zip(names, dem):
    i = info[n]
    d = re.sub(r'^void ttemb::', '', d.split('(')[0])
    print(f"{d[:90]:90s} vgpr {i.get('vgpr', 0):4d} waves/SIMD {512 // max(i.get('vgpr', 1), 1):2d} scratch {i.get('scratch', 0):5d} lds {i.get('lds', 0)}")
