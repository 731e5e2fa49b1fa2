cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_COUNT WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ONLY="up8_4 3x3 256->256 @120x216" ITERS=4 timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc -o rowconv_g$i -- python tools/conv_microbench.py > gpurun_out/pmc/rowconv_g$i.log 2>&1
  echo "pmc group $i exit $?"
done
ls gpurun_out/pmc
