#!/usr/bin/env python3
"""Per-kernel time summary from a rocprofv3 rocpd database (sqlite): tools/rocpd_stats.py DB [n_steps] [csv_out]."""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(db.execute('select name, start, end from kernels order by start'))
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for n, s, e in rows:
        n = n.replace('void ', '').replace('(anonymous namespace)::', '')
        n = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*(\[clone.*\])?$', '', n)
        a = agg[n]
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(v[1] for v in agg.values())
    lines = ['Name,Calls,TotalDurationUs,AverageUs,Percentage,MinUs,MaxUs']
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{k}",{v[0]},{v[1]:.1f},{v[1] / v[0]:.2f},{100 * v[1] / tot:.2f},{v[2]:.2f},{v[3]:.2f}')
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write('\n'.join(lines) + '\n')
    print(f'total {tot / 1e3:.2f} ms over {len(rows)} dispatches; per step {tot / 1e3 / div:.2f} ms')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'{v[1] / 1e3 / div:9.2f} ms/step {v[0] / div:7.1f} calls  avg {v[1] / v[0]:9.1f} us  {k[:110]}')


if __name__ == '__main__':
    main()
