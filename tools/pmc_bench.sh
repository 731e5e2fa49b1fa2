#!/bin/bash
# PMC counter passes over the bench command (one rocprofv3 run per counter group, --kernel-trace only):
# HBM traffic and MFMA utilisation of the convolution kernels per frame.  Output: gpurun_out/pmc/bench_g*.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for grp in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc -o bench_g$i -- \
    python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_extra --no_affinity > gpurun_out/pmc/bench_g$i.log 2>&1
  echo "pmc bench group $i exit $?"
done
python tools/pmc_summary.py conv gpurun_out/pmc/bench gpurun_out/pmc/conv_traffic.json
