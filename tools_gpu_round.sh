#!/bin/bash
# One GPU-box session: hardware facts, build check, the -m gpu tests (one process per file so a hung
# kernel only costs that file), the bench line and a rocprofv3 kernel trace.  Everything of interest
# is written under gpurun_out/.
mkdir -p gpurun_out
cd "$(dirname "$0")"
export TMPDIR=/tmp
( rocminfo | grep -E "Name:|Compute Unit|Max Clock|LDS|Wavefront|Size:" | head -60; rocm-smi --showmeminfo vram --showclocks 2>&1 | head -40; echo "nproc=$(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" ; free -g | head -2 ) > gpurun_out/hw.txt 2>&1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build exit $?"
for f in tests/test_gpu_a_conv.py tests/test_gpu_b_pointwise.py tests/test_gpu_c_bank.py tests/test_gpu_d_affinity.py tests/test_gpu_e_network.py; do
  timeout -k 10 ${TEST_TIMEOUT:-420} python -m pytest $f -m gpu -q -s -p no:cacheprovider > gpurun_out/$(basename $f .py).log 2>&1
  rc=$?
  echo "$f exit $rc : $(tail -1 gpurun_out/$(basename $f .py).log)"
  if [ $rc -ge 124 ]; then echo "crash/hang in $f -- stopping this session"; exit 1; fi
done
if [ "${RUN_BENCH:-1}" = "1" ]; then
  DEVA_BENCH_LAYERS=gpurun_out/conv_layers.json timeout -k 10 ${BENCH_TIMEOUT:-500} python bench.py --steps ${BENCH_STEPS:-20} --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; brc=$?; echo "bench exit $brc"; tail -c 3000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
  if [ $brc -ne 0 ]; then echo "bench failed -- skipping the rest"; exit 1; fi
fi
python tools/conv_microbench.py > gpurun_out/conv_microbench.txt 2>&1; cat gpurun_out/conv_microbench.txt
if [ "${LIST_COUNTERS:-0}" = "1" ]; then rocprofv3 -L > gpurun_out/counters.txt 2>&1; fi
if [ "${RUN_PROF:-1}" = "1" ]; then
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${PROF_TAG:-r01} -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_extra > gpurun_out/prof.log 2>&1; echo "prof exit $?"
  ls -R gpurun_out/prof | head -20
fi
