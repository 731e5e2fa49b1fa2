#!/bin/bash
# PMC passes over convlab on one library: wave-cycle breakdown, instruction mix, LDS.  Usage: LIB=... SET=big tools/convlab/pmc.sh
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/lab/${TAG:-pmc}
mkdir -p $OUT
LAB=tools/convlab/convlab
LIB=${LIB:-tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o g$i -- \
    $LAB --libs $LIB --set ${SET:-big} --iters 3 ${LAB_ARGS} > $OUT/g$i.log 2>&1
  echo "pmc group $i exit $?"
done
python tools/convlab/pmc_table.py $OUT/g1 $OUT/g2 $OUT/g3 > $OUT/table.txt 2>&1
grep -v "rocclr" $OUT/table.txt | head -${LINES_OUT:-140}
