#!/bin/bash
# kernel-trace of a few layers: per-kernel duration and the gap to the previous kernel (steady state)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/lab/trace
rm -rf $OUT; mkdir -p $OUT
for only in "${@}"; do
  tag=$(echo "$only" | tr ' >@' '___')
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o t -- tools/convlab/convlab --libs ${LIB:-tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so} --only "$only" --warm_ms 3 --time_ms 3 > $OUT/$tag.log 2>&1
  python3 - "$OUT/$tag" "$only" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]  # steady state
from collections import defaultdict
dur = defaultdict(list); gap = defaultdict(list)
prev_end = None
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-60:]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[n].append(e - s)
    if prev_end is not None: gap[n].append(s - prev_end)
    prev_end = e
print('==', sys.argv[2])
for n in dur:
    d = sorted(dur[n]); g = sorted(gap[n]) or [0]
    print(f'   {n:62s} n={len(d):5d} dur median {d[len(d)//2]/1e3:7.2f} us  gap-before median {g[len(g)//2]/1e3:6.2f} us')
PY
done
