#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in one csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py conv.hip [filter-substring ...] [-DNAME=VALUE ...]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gan_lab_amd', 'csrc')


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith('-')]
    defs = [a for a in sys.argv[2:] if a.startswith('-')]
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-c', src, '-o', '/dev/null',
                        '-Rpass-analysis=kernel-resource-usage'] + defs, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-3000:])
        sys.exit(1)
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r'remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass', ln) or re.search(r'remark: (.*?) \[-Rpass', ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:') or t.startswith('Name:'):
            cur = {'name': t.split(':', 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.split(':', 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(['c++filt'], input='\n'.join(x['name'] for x in rows),
                           capture_output=True, text=True).stdout.splitlines()
    for x, n in zip(rows, names):
        n = n.replace('(anonymous namespace)::', '').replace('(ConvArgs)', '')
        if flt and not all(f in n for f in flt):
            continue
        print(f"{x.get('VGPRs', '?'):>4} vgpr {x.get('AGPRs', '0'):>3} agpr  occ {x.get('Occupancy [waves/SIMD]', '?')}  "
              f"lds {x.get('LDS Size [bytes/block]', '?'):>6}  scratch {x.get('ScratchSize [bytes/lane]', '?'):>3}  {n[:150]}")


if __name__ == '__main__':
    main()
