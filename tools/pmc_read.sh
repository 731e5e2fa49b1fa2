#!/bin/bash
# PMC counter passes over deva_affinity_read at the BASELINE shape (one rocprofv3 run per counter group,
# --kernel-trace only): the fp16 pre-filter kernels and, from the same command, the fp32 kernels (pre-filter off).
# usage: tools/pmc_read.sh [tag]   -> gpurun_out/pmc/<tag>_summary.json (committed as profiles/pmc_r03/affinity_read.json)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-read}
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  SHAPES=10000x8160 ITERS=3 timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc -o ${tag}_g$i -- python tools/affinity_read_check.py > gpurun_out/pmc/${tag}_g$i.log 2>&1
  echo "pmc $tag group $i exit $?"
done
python tools/pmc_summary.py read gpurun_out/pmc/$tag gpurun_out/pmc/${tag}_summary.json > gpurun_out/pmc/${tag}_summary.txt 2>&1
cat gpurun_out/pmc/${tag}_summary.txt | head -80
