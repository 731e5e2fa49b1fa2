#!/bin/bash
# Round-3 GPU-box session: like tools/gpu_session.sh with the round-3 sections.  Usage: tools/gpu_session_r03.sh [sections]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
SECTIONS="${@:-tests bench}"
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build exit $?"
run_test() {
  local f=$1; local to=${2:-900}; shift; shift
  local t0=$SECONDS
  timeout -k 10 $to python -m pytest tests/$f.py -m gpu -q -s -p no:cacheprovider "$@" > gpurun_out/$f.log 2>&1
  local rc=$?
  echo "$f exit $rc in $((SECONDS-t0)) s : $(tail -1 gpurun_out/$f.log)"
  if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/$f.log | head -20; fi
  return $rc
}
if has tests; then
  for f in ${TEST_FILES:-test_gpu_e_network test_gpu_g_fullsize}; do
    if [ -n "$PYTEST_K" ]; then run_test $f ${TEST_TIMEOUT:-900} -k "$PYTEST_K"; else run_test $f ${TEST_TIMEOUT:-900}; fi
  done
fi
if has alltests; then
  for f in test_gpu_a_conv test_gpu_b_pointwise test_gpu_c_bank test_gpu_d_affinity test_gpu_f_memory_events test_gpu_e_network test_gpu_g_fullsize; do run_test $f ${TEST_TIMEOUT:-900}; done
fi
if has custom; then eval "$CUSTOM_CMD"; fi
if has readcheck; then
  SHAPES=${READ_SHAPES:-} ITERS=10 timeout -k 10 300 python tools/affinity_read_check.py > gpurun_out/affinity_read_check.txt 2>&1
  cat gpurun_out/affinity_read_check.txt
fi
if has affinity; then
  SHAPES=${AFF_SHAPES:-1620x1620,8100x1620,10000x1620,24580x1620,10000x8160,83440x8160,50000x32400} ITERS=10 \
    timeout -k 10 200 python tools/affinity_microbench.py > gpurun_out/affinity_micro.txt 2>&1
  cat gpurun_out/affinity_micro.txt
fi
if has bench; then
  DEVA_BENCH_LAYERS=gpurun_out/conv_layers_480p5.json timeout -k 10 ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS:---steps 40 --warmup 5} > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"
  tail -c 9000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
if has prof; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${PROF_TAG:-r03} -- python bench.py --steps 20 --warmup 3 --no_cpu_baseline > gpurun_out/prof.log 2>&1; echo "prof exit $?"
  ls -R gpurun_out/prof | head -20
fi
if has readprof; then
  SHAPES=${READ_SHAPES:-10000x8160} ITERS=10 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/readprof -o read -- python tools/affinity_read_check.py > gpurun_out/readprof.log 2>&1
  python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/readprof/**/read_kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
fi
if has pmcread; then bash tools/pmc_read.sh read; fi
if has pmcbench; then bash tools/pmc_bench.sh; fi
