#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group, --kernel-trace only) over a microbench.
# usage: tools/prof_pmc.sh <tag> <command...>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc -o ${tag}_g$i -- "$@" > gpurun_out/pmc/${tag}_g$i.log 2>&1
  echo "pmc group $i exit $?"
done
ls gpurun_out/pmc | head -30
