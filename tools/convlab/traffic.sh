#!/bin/bash
# L2-fill / write-back bytes per layer of convlab on one library: FETCH_SIZE and WRITE_SIZE passes, per-dispatch averages
# by (kernel, grid).  Usage: LIB=... TAG=traffic_x SET=big [ENVS="DEVA_CONV_GROUP_M=2"] tools/convlab/traffic.sh
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/lab/${TAG:-traffic}
mkdir -p $OUT
LAB=tools/convlab/convlab
LIB=${LIB:-tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so}
i=0
for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT"; do
  i=$((i+1))
  env ${ENVS} timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o g$i -- \
    $LAB --libs $LIB --set ${SET:-big} --iters 2 --warm_ms 1 --time_ms 1 ${LAB_ARGS} > $OUT/g$i.log 2>&1
  echo "traffic group $i exit $?"
done
python tools/convlab/traffic_table.py $OUT > $OUT/table.txt 2>&1
cat $OUT/table.txt | head -${LINES_OUT:-60}
