#!/bin/bash
# arbitrary counter groups over convlab, per-(kernel, grid) averages.
# Usage: GROUPS_="A B C;D E" LIB=... TAG=x LAB_ARGS="--only gru" [ENVS=...] tools/convlab/pmc_raw.sh
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/lab/${TAG:-raw}
mkdir -p $OUT
LIB=${LIB:-tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so}
i=0
IFS=';' read -ra GR <<< "$GROUPS_"
for grp in "${GR[@]}"; do
  i=$((i+1))
  env ${ENVS} timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o g$i -- \
    tools/convlab/convlab --libs $LIB --set ${SET:-big} --iters 2 --warm_ms 1 --time_ms 1 ${LAB_ARGS} > $OUT/g$i.log 2>&1
  echo "group $i ($grp) exit $?"
done
python - $OUT <<'PY'
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(path)):
        name = re.sub(r'^void ', '', r['Kernel_Name']).replace('deva::(anonymous namespace)::', '').replace('deva::', '')
        m = re.match(r'(\w+)<([^>]*)>', name)
        short = (m.group(1) + '<' + m.group(2).replace(' ', '') + '>') if m else re.sub(r'\(.*', '', name)
        if 'conv' not in short and 'splitk' not in short: continue
        c = acc[(short, int(r['Grid_Size']))][r['Counter_Name']]
        c[0] += float(r['Counter_Value']); c[1] += 1
for key in sorted(acc):
    print(key[0], 'grid', key[1], ' '.join(f'{k}={v[0] / v[1]:.0f}' for k, v in sorted(acc[key].items())))
PY
