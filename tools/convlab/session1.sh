#!/bin/bash
# convlab session 1: baseline timings of every conv layer of the 480p/5-object frame + PMC passes over the big layers
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/lab
LAB=tools/convlab/convlab
LIBS=${LIBS:-tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so}
timeout 120 $LAB --libs $LIBS --iters 20 > gpurun_out/lab/baseline.txt 2>&1; echo "baseline exit $?"
cat gpurun_out/lab/baseline.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/lab/pmc_g$i -o g$i -- \
    $LAB --libs $LIBS --set big --iters 3 > gpurun_out/lab/pmc_g$i.log 2>&1
  echo "pmc group $i exit $?"
done
python tools/convlab/pmc_table.py gpurun_out/lab/pmc_g1 gpurun_out/lab/pmc_g2 gpurun_out/lab/pmc_g3 > gpurun_out/lab/pmc_table.txt 2>&1
cat gpurun_out/lab/pmc_table.txt | head -150
