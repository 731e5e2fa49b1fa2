#!/bin/bash
# One GPU-box session (round 2): build check, the -m gpu test files one process each (a hung kernel
# only costs that file), affinity microbench in both kernel shapes, bench line, optional rocprof.
# Everything of interest is written under gpurun_out/.  Usage: tools/gpu_session.sh [sections...]
# sections: tests affinity bench prof pmc   (default: tests affinity bench)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
SECTIONS="${@:-tests affinity bench}"
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
( rocminfo | grep -E "Name:|Compute Unit|Max Clock" | head -12; echo "nproc=$(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)" ) > gpurun_out/hw.txt 2>&1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build exit $?"
run_test() {
  local f=$1; local to=${2:-420}
  timeout -k 10 $to python -m pytest tests/$f.py -m gpu -q -s -p no:cacheprovider > gpurun_out/$f.log 2>&1
  local rc=$?
  echo "$f exit $rc : $(tail -1 gpurun_out/$f.log)"
  return $rc
}
if has tests; then
  for f in ${TEST_FILES:-test_gpu_a_conv test_gpu_b_pointwise test_gpu_c_bank test_gpu_d_affinity test_gpu_f_memory_events test_gpu_e_network test_gpu_g_fullsize}; do
    run_test $f ${TEST_TIMEOUT:-600}
    rc=$?
    if [ $rc -ge 124 ]; then echo "hang/crash in $f -- stopping"; exit 1; fi
    if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/$f.log | head -12; fi
  done
fi
if has affshapes; then  # the affinity tests with every kernel shape forced in turn
  for shape in ${AFF_TEST_SHAPES:-1 2 3 4 5 6 7 8}; do
    DEVA_AFFINITY_SHAPE=$shape timeout -k 10 300 python -m pytest tests/test_gpu_d_affinity.py -m gpu -q -p no:cacheprovider > gpurun_out/test_gpu_d_affinity_shape$shape.log 2>&1
    echo "test_gpu_d_affinity with shape $shape exit $? : $(tail -1 gpurun_out/test_gpu_d_affinity_shape$shape.log)"
  done
fi
if has custom; then
  eval "$CUSTOM_CMD"
fi
if has affinity; then
  for shape in ${AFF_SHAPE_LIST:-2 4}; do
    DEVA_AFFINITY_SHAPE=$shape SHAPES=${AFF_SHAPES:-1620x1620,8100x1620,10000x1620,24580x1620,10000x8160,83440x8160,50000x32400} ITERS=10 \
      timeout -k 10 200 python tools/affinity_microbench.py > gpurun_out/affinity_shape$shape.txt 2>&1
    echo "--- affinity shape $shape"; cat gpurun_out/affinity_shape$shape.txt
  done
fi
if has bench; then
  timeout -k 10 ${BENCH_TIMEOUT:-600} python bench.py --steps ${BENCH_STEPS:-40} --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"
  tail -c 6000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
if has prof; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${PROF_TAG:-r02} -- python bench.py --steps 20 --warmup 3 --no_cpu_baseline > gpurun_out/prof.log 2>&1; echo "prof exit $?"
  ls -R gpurun_out/prof | head -20
fi
if has pmc; then
  bash tools/pmc_affinity.sh
fi
