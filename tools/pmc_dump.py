#!/usr/bin/env python3
"""Counter values per kernel dispatch from a rocprofv3 --pmc rocpd database: tools/pmc_dump.py DB [kernel-substring].
(The rocpd schema keeps them in pmc_events joined to kernel dispatches; table / view names carry a uuid suffix.)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = next((n for n in names if n == 'counters_collection'), None)
    if view is None:
        print('tables/views:', names)
        return
    cols = [r[1] for r in db.execute(f'pragma table_info({view})')]
    print('#', cols)
    kcol = 'kernel_name' if 'kernel_name' in cols else next(c for c in cols if 'name' in c and 'counter' not in c)
    rows = db.execute(f'select dispatch_id, {kcol}, counter_name, sum(value), min(start), max(end) from {view} '
                      f'group by dispatch_id, counter_name order by dispatch_id')
    for d, k, c, v, s, e in rows:
        if want in k:
            print(f"{d},{k[:240]},{c},{v:.0f},{(e - s) if (s and e) else 0}")


if __name__ == '__main__':
    main()
