#!/bin/bash
# Round-4 evidence on the GPU box (sections: tests alltests pmc bench trace trace1080):  tools/evidence_r04.sh pmc bench trace
# Everything lands in gpurun_out/r04/; the summaries are copied into profiles/r04*/ by hand (tracked).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04
mkdir -p $OUT
SECTIONS="${@:-pmc bench trace}"
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
python __graft_entry__.py > $OUT/build.log 2>&1; echo "build exit $?"
CMD="python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_extra --no_affinity"
run_test() {
  local f=$1; local to=${2:-900}
  local t0=$SECONDS
  timeout -k 10 $to python -m pytest tests/$f.py -m gpu -q -s -p no:cacheprovider > $OUT/$f.log 2>&1
  local rc=$?
  echo "$f exit $rc in $((SECONDS-t0)) s : $(tail -1 $OUT/$f.log)"
  if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" $OUT/$f.log | head -20; fi
}
if has tests; then for f in ${TEST_FILES:-test_gpu_a_conv test_gpu_e_network}; do run_test $f ${TEST_TIMEOUT:-900}; done; fi
if has alltests; then
  for f in test_gpu_a_conv test_gpu_b_pointwise test_gpu_c_bank test_gpu_d_affinity test_gpu_f_memory_events test_gpu_e_network test_gpu_i_drivers test_gpu_g_fullsize; do
    [ -f tests/$f.py ] && run_test $f ${TEST_TIMEOUT:-1500}
  done
fi
if has pmc; then
  rm -rf gpurun_out/pmc; bash tools/pmc_bench.sh
  cp gpurun_out/pmc/conv_traffic.json $OUT/conv_traffic.json 2>/dev/null
  mkdir -p profiles/pmc_r04 && cp gpurun_out/pmc/conv_traffic.json profiles/pmc_r04/conv_traffic.json   # what bench.py reads (same build: sha checked)
  python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/conv_traffic.json'))
print({k: v for k, v in d.items() if k not in ('per_dispatch_averages',)})
PY
fi
if has bench; then
  DEVA_BENCH_LAYERS=$OUT/conv_layers_480p5.json timeout -k 10 ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS:---steps 40 --warmup 5} > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04/bench.json').read().strip().split('\n')[-1])
r = d['roofline']
print('headline %.1f FPS, conv %.1f TF = %.3f, ms in conv %.3f' % (d['value'], r['achieved'], r['frac'], r['ms_in_kernel_per_frame']))
print('eval_vos-style', d.get('timed_like_eval_vos'))
for e in d.get('also', []):
    print('  also: %-90s %s' % (e['metric'][:90], e.get('value')))
for e in d.get('also_kernels', []) if isinstance(d.get('also_kernels'), list) else []:
    print('  kernel: %-60s %-36s %8.1f us %7.0f GB/s %.3f' % (e['kernel'][:60], e['shape'][:36], e.get('us', 0), e.get('gbps', 0), e.get('frac_of_hbm_peak', 0)))
print('affinity', {k: v for k, v in d.get('affinity', {}).items() if k in ('us_read', 'f16_mfma_frac', 'hbm_algorithmic_frac')})
print('cpu', {k: d.get('cpu_baseline', {}).get(k) for k in ('value', 'cores', 'kind')})
PY
fi
if has trace; then
  rm -rf $OUT/trace
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1; echo "trace exit $?"
  python tools/kernel_stats_md.py $OUT/trace "rocprofv3 --kernel-trace -- $CMD" > $OUT/kernel_stats.md
  head -12 $OUT/kernel_stats.md
  find $OUT/trace -name "*.csv" -size +20M -delete   # the raw trace stays on the box
fi
if has trace1080; then
  # the 1080p / 1-object frame loop (working memory only): per-layer table + kernel trace
  rm -rf $OUT/trace1080
  DEVA_BENCH_LAYERS=$OUT/conv_layers_1080p1.json timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace1080 -o t -- \
    python bench.py --height 1080 --width 1920 --objects 1 --steps 12 --warmup 3 --no_cpu_baseline --no_extra --no_affinity > $OUT/bench_1080p1.json 2> $OUT/trace1080.log; echo "trace1080 exit $?"
  python tools/kernel_stats_md.py $OUT/trace1080 "rocprofv3 --kernel-trace -- python bench.py --height 1080 --width 1920 --objects 1 --steps 12 --warmup 3 --no_cpu_baseline --no_extra --no_affinity" "1080p / 1-object loop (working memory only)" $OUT/bench_1080p1.json > $OUT/kernel_stats_1080p1.md
  head -8 $OUT/kernel_stats_1080p1.md
  find $OUT/trace1080 -name "*.csv" -size +20M -delete
fi
