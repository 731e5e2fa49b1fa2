#!/bin/bash
# PMC counter passes over the affinity microbench (one rocprofv3 run per counter group, --kernel-trace only).
# usage: tools/pmc_affinity.sh <tag> [env assignments for the microbench, e.g. DEVA_AFFINITY_SHAPE=2]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-aff}; shift
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  env "$@" SHAPES=10000x8160,83440x8160 ITERS=3 timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc -o ${tag}_g$i -- python tools/affinity_microbench.py > gpurun_out/pmc/${tag}_g$i.log 2>&1
  echo "pmc $tag group $i exit $?"
done
python tools/pmc_summary.py aff gpurun_out/pmc/$tag gpurun_out/pmc/${tag}_summary.json > gpurun_out/pmc/${tag}_summary.txt 2>&1
