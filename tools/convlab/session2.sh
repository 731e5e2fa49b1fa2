#!/bin/bash
# convlab session: A/B of libraries on the 480p/5-object conv layers, with output comparison against the first library
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/lab
LAB=tools/convlab/convlab
LIBS=${LIBS:-tools/convlab/libdeva_r3.so,tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so}
timeout 300 $LAB --libs $LIBS --iters ${ITERS:-20} --check ${LAB_ARGS} > gpurun_out/lab/ab.txt 2>&1; echo "ab exit $?"
cat gpurun_out/lab/ab.txt
