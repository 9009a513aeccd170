#!/bin/bash
# prints registers / occupancy of the ttemb kernels in one .hip file (compile-only)
f=${1:-ttemb_fast3.hip}; shift
cd /root/repo/falcon-ttdforgnns_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include "$@" -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur={}
def show(c):
    if c.get('n','').startswith('_ZN5ttemb'): print(c)
for l in sys.stdin:
    m=re.search(r': Name: (\S+)',l) or re.search(r'Function Name: (\S+)',l)
    if m:
        show(cur); cur={'n':m.group(1)[9:60]}
    for k in ('VGPRs','AGPRs','Occupancy [waves/SIMD]','LDS Size [bytes/block]','ScratchSize [bytes/lane]'):
        m=re.search(re.escape(k)+r': (\d+)',l)
        if m: cur[k.split()[0]]=int(m.group(1))
show(cur)
"
