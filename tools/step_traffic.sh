#!/bin/bash
# HBM bytes per kernel over the 4 steps of a short bench run (1 warm-up + 2 timed + the FLOP-counting step), per-step figures printed: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on a short bench run,
# summed per kernel symbol.   tools/step_traffic.sh <tag>  -> gpurun_out/<tag>_step_traffic.txt
set -e
TAG=${1:-traffic}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  D=$OUT/${TAG}_tmp; rm -rf "$D"; mkdir -p "$D"
  rocprofv3 --pmc $ctr --kernel-trace -d "$D" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline \
      > /dev/null 2> "$OUT/${TAG}_rocprof.err" || true
  DB=$(find "$D" -name '*.db' | head -1)
  python3 "$ROOT/tools/pmc_dump.py" "$DB" "" > "$OUT/${TAG}_$ctr.txt" 2>&1 || true
  rm -rf "$D"
done
python3 - "$OUT/${TAG}_FETCH_SIZE.txt" "$OUT/${TAG}_WRITE_SIZE.txt" "$OUT/${TAG}_step_traffic.json" > "$OUT/${TAG}_step_traffic.txt" <<'PY'
import collections, json, re, sys
STEPS = 4     # the profiled command runs 1 warm-up + 2 timed steps + the FLOP-counting step
def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', name)[:70]
def load(p):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for line in open(p):
        if line.startswith('#'): continue
        parts = line.rstrip('\n').split(',')
        if len(parts) < 5: continue
        a = agg[short(','.join(parts[1:-3]))]; a[0] += 1; a[1] += float(parts[-2]); a[2] += float(parts[-1])
    return agg
f, w = load(sys.argv[1]), load(sys.argv[2])
rows, table = [], {}
for k, (n, kb, ns) in f.items():
    rb = kb * 2 * 1e3          # FETCH_SIZE is in KB and reads half on gfx950 (MI355X_MICROARCH.md)
    wb = w.get(k, [0, 0, 0])[1] * 1e3
    rows.append((rb + wb, k, n, rb, wb, ns))
    table[k] = {'launches': n, 'read_bytes_per_launch': rb / n, 'written_bytes_per_launch': wb / n,
                'average_us_under_pmc': ns / n / 1e3}
json.dump({'steps_profiled': STEPS, 'command': 'bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline',
           'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x 2 x 1e3 (gfx950: KB, '
                     'counts half), WRITE_SIZE x 1e3', 'kernels': table}, open(sys.argv[3], 'w'), indent=1)
print(f'kernel, launches ({STEPS} steps), HBM read GB per step, written GB per step, ms per step, average TB/s')
for tot, k, n, rb, wb, ns in sorted(rows, reverse=True)[:40]:
    print(f'{k:70s} {n:5d} {rb / 1e9 / STEPS:8.2f} {wb / 1e9 / STEPS:8.2f} {ns / 1e6 / STEPS:8.2f} {(rb + wb) / max(ns, 1) / 1e3:6.2f}')
PY
head -45 "$OUT/${TAG}_step_traffic.txt"
