#!/usr/bin/env python3
"""Static instruction mix of every kernel in a hipcc -save-temps device assembly (*.s): MFMA, packed / other VALU, LDS, SALU,
memory.  tools/isa_count.py file.s   (hipcc --offload-arch=gfx950 -O3 -c x.hip -save-temps=obj)"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
for f in re.split(r'\n(?=_Z[\w]+:\s)', txt):
    m = re.match(r'(_Z\w+):', f)
    if not m or 'kernel' not in m.group(1):
        continue
    body = f.split('.Lfunc_end')[0]
    c = collections.Counter()
    for line in body.splitlines():
        t = line.strip().split()
        if not t or t[0].endswith(':') or t[0].startswith('.') or t[0].startswith(';'):
            continue
        op = t[0]
        key = ('mfma' if op.startswith('v_mfma') else 'v_pk' if op.startswith('v_pk_') else 'accvgpr' if op.startswith('v_accvgpr')
               else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'waitcnt' if op.startswith('s_waitcnt')
               else 'salu' if op.startswith('s_') else 'vmem' if op.startswith(('buffer_', 'global_')) else 'other')
        c[key] += 1
    print(m.group(1)[:80], dict(c))
