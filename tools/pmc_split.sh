#!/bin/bash
# PMC counter passes over the 8-segment 1080p clip under --f16_split --f16_split_key_encoder (one rocprofv3 run per
# counter group, --kernel-trace only): HBM traffic of the convolution kernels per frame -> conv_traffic_split.json
# (the split `also` lines carried no traffic figure).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_split}
mkdir -p $OUT
export DEVA_LINE_STEPS=10 DEVA_LINE_WARMUP=6
i=0
for grp in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o split_g$i -- \
    python tools/bench_line.py 8seg split_all $OUT/line_g$i.json > $OUT/split_g$i.log 2>&1
  echo "pmc split group $i exit $?"
done
python tools/pmc_summary.py split $OUT/split $OUT/line_g1.json $OUT/conv_traffic_split.json
find $OUT -name "*.csv" -size +8M -delete
