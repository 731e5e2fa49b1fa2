#!/bin/bash
# policy sweep: forced tile x split-K target on the probes library, CSV per configuration -> gpurun_out/lab/sweep/
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/lab/sweep
LAB=tools/convlab/convlab
for tile in ${TILES:-0 128 64 12864 64128}; do
  for split in ${SPLITS:--1 0 256 384 512 768 1024}; do
    DEVA_CONV_TILE=$tile DEVA_CONV_SPLIT_TARGET=$split timeout 120 $LAB --libs tools/convlab/libconv_probes.so --iters ${ITERS:-10} --csv ${LAB_ARGS} \
      | grep "^csv" > gpurun_out/lab/sweep/t${tile}_s${split}.csv
  done
done
python tools/convlab/sweep_table.py gpurun_out/lab/sweep
