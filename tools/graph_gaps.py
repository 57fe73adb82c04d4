#!/usr/bin/env python3
"""GPU-side picture of a launch-bound configuration from a rocprofv3 rocpd database: the longest stretch of dispatches
without a host-sized hole (> 2 ms) - the timed, replayed iterations of `bench.py --config 2 --step-graph 1` - with its
busy / idle time, the idle time between consecutive kernels and the per-kernel sums per iteration.
    tools/graph_gaps.py DB steps_in_that_stretch [csv_out]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = list(db.execute('select name, start, end from kernels order by start'))
    best, cur0 = (0, 0), 0
    for i in range(1, len(rows) + 1):
        if i == len(rows) or rows[i][1] - rows[i - 1][2] > 2e6:
            if i - cur0 > best[1] - best[0]:
                best = (cur0, i)
            cur0 = i
    seg = rows[best[0]:best[1]]
    span = (seg[-1][2] - seg[0][1]) / 1e6
    busy = sum(e - s for _, s, e in seg) / 1e6
    gaps = [max(0, seg[i][1] - seg[i - 1][2]) / 1e3 for i in range(1, len(seg))]
    print(f'{len(seg)} dispatches in {span:.2f} ms: busy {busy:.2f} ms, idle {span - busy:.2f} ms; per iteration '
          f'({steps:g}): {len(seg) / steps:.0f} dispatches, span {span / steps:.3f} ms, busy {busy / steps:.3f} ms')
    gs = sorted(gaps)
    print(f'gap between consecutive kernels: median {gs[len(gs) // 2]:.2f} us, mean {sum(gs) / len(gs):.2f} us, '
          f'p90 {gs[int(len(gs) * .9)]:.2f} us, max {gs[-1]:.1f} us')
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in seg:
        n = n.replace('void ', '').replace('(anonymous namespace)::', '')
        n = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*(\[clone.*\])?$', '', n)
        agg[n][0] += 1
        agg[n][1] += (e - s) / 1e3
    lines = ['Name,CallsPerIteration,UsPerIteration,AverageUs']
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{k}",{v[0] / steps:.1f},{v[1] / steps:.1f},{v[1] / v[0]:.2f}')
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write('\n'.join(lines) + '\n')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f'{v[1] / 1e3 / steps:8.3f} ms/iter {v[0] / steps:7.1f} calls  avg {v[1] / v[0]:8.1f} us  {k[:120]}')


if __name__ == '__main__':
    main()
