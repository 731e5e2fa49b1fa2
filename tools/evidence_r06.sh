#!/bin/bash
# Round-6 evidence on the GPU box:  tools/evidence_r06.sh <sections>
#   tests      the -m gpu suite, one log per file (gpurun_out/r06/tests/)
#   tests_split the network / full-size / driver files again with DEVA_TEST_F16_SPLIT=all (every parity gate under
#              --f16_split --f16_split_key_encoder, unchanged bounds; gpurun_out/r06/tests_split/)
#   pmc        PMC passes over the bench command (conv HBM traffic, MFMA utilisation, effective clock) and over the read
#   bench      the bench line (+ per-layer table of the headline)
#   trace      kernel trace of the headline loop        trace4k   kernel trace of the 4K line
#   trace8seg  kernel trace of the 8-segment 1080p clip under --f16_split --f16_split_key_encoder
#   pmcsplit   PMC passes (HBM traffic) over that clip -> conv_traffic_split.json
# Everything lands in gpurun_out/r06/; what is judged is copied into profiles/r06/ and profiles/pmc_r06/ (tracked).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06
mkdir -p $OUT
SECTIONS="${@:-pmc bench trace}"
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
python __graft_entry__.py > $OUT/build.log 2>&1; echo "build exit $?"
run_test() {
  local dir=$1 f=$2 to=${3:-1500}
  mkdir -p $OUT/$dir
  local t0=$SECONDS
  timeout -k 10 $to python -m pytest tests/$f.py -m gpu -q -s -p no:cacheprovider --durations=12 > $OUT/$dir/$f.log 2>&1
  local rc=$?
  echo "$dir/$f exit $rc in $((SECONDS-t0)) s : $(tail -1 $OUT/$dir/$f.log) [sources $(cat tracking-anything-with-deva_amd/csrc/*.hip tracking-anything-with-deva_amd/csrc/*.h include/*.h | sha1sum | cut -c1-12)]" | tee -a $OUT/$dir/summary.txt
  if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" $OUT/$dir/$f.log | head -20; fi
}
if has tests; then
  rm -f $OUT/tests/summary.txt
  for f in ${TEST_FILES:-test_gpu_a_conv test_gpu_b_pointwise test_gpu_c_bank test_gpu_d_affinity test_gpu_f_memory_events test_gpu_e_network test_gpu_i_drivers test_gpu_h_reference_drivers test_gpu_g_fullsize}; do
    [ -f tests/$f.py ] && run_test tests $f
  done
fi
if has tests_split; then
  rm -f $OUT/tests_split/summary.txt
  export DEVA_TEST_F16_SPLIT=all
  for f in ${SPLIT_TEST_FILES:-test_gpu_e_network test_gpu_i_drivers test_gpu_g_fullsize}; do run_test tests_split $f; done
  unset DEVA_TEST_F16_SPLIT
fi
if has pmc; then
  rm -rf gpurun_out/pmc; bash tools/pmc_bench.sh > $OUT/pmc_bench.log 2>&1
  mkdir -p profiles/pmc_r06 && cp gpurun_out/pmc/conv_traffic.json profiles/pmc_r06/conv_traffic.json && cp gpurun_out/pmc/conv_traffic.json $OUT/conv_traffic.json
  bash tools/pmc_read.sh read > $OUT/pmc_read.log 2>&1
  cp gpurun_out/pmc/read_summary.json profiles/pmc_r06/affinity_read.json && cp gpurun_out/pmc/read_summary.json $OUT/affinity_read.json
  # effective clock of a register-only fp32 MFMA stream (bench.py's sustained probe) from GRBM_GUI_ACTIVE / kernel time
  timeout -k 5 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc -o probe_g1 -- \
    python tools/probe_clock.py > $OUT/probe_clock.log 2>&1
  python tools/pmc_summary.py clock gpurun_out/pmc/probe gpurun_out/pmc/bench profiles/pmc_r06/effective_clock.json > $OUT/effective_clock.txt 2>&1
  cp profiles/pmc_r06/effective_clock.json $OUT/ 2>/dev/null
  tail -30 $OUT/effective_clock.txt
  python - <<'PY'
import json
d = json.load(open('gpurun_out/r06/conv_traffic.json'))
print({k: v for k, v in d.items() if k not in ('per_dispatch_averages',)})
r = json.load(open('gpurun_out/r06/affinity_read.json'))
print('read_total', r.get('read_total'))
PY
fi
if has bench; then
  DEVA_BENCH_LAYERS=$OUT/conv_layers_480p5.json timeout -k 10 ${BENCH_TIMEOUT:-1200} python bench.py ${BENCH_ARGS:---steps 40 --warmup 5} > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cp bench_extra.json $OUT/bench_extra.json 2>/dev/null
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06/bench.json').read().strip().split('\n')[-1])
r = d['roofline']
print('headline %.1f FPS, conv %.1f TF = %.3f, ms in conv %.3f' % (d['value'], r['achieved'], r['frac'], r['ms_in_kernel_per_frame']))
print({k: v for k, v in d['config'].items() if k.startswith('fps')})
print({k: v for k, v in r.items() if k.startswith(('affinity', 'f16_split', 'traffic'))})
print('cpu', {k: d.get('cpu_baseline', {}).get(k) for k in ('value', 'cores', 'kind', 'runs_fps')})
PY
fi
CMD="python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_extra --no_affinity"
if has trace; then
  rm -rf $OUT/trace
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1; echo "trace exit $?"
  python tools/kernel_stats_md.py $OUT/trace "rocprofv3 --kernel-trace -- $CMD" > $OUT/kernel_stats.md
  head -12 $OUT/kernel_stats.md
  find $OUT/trace -name "*.csv" -size +5M -delete   # the raw trace stays on the box
fi
if has trace4k; then
  rm -rf $OUT/trace4k
  CMD4="python bench.py --workload long4k --gpus 1 --steps 10 --warmup 3"
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace4k -o t -- $CMD4 > $OUT/bench_4k.json 2> $OUT/trace4k.log; echo "trace4k exit $?"
  python tools/trace_top.py $OUT/trace4k 40 "rocprofv3 --kernel-trace -- $CMD4 (4K, 1 object, 50 000-token long-term bank, one GPU)" > $OUT/kernel_stats_4k.md
  head -14 $OUT/kernel_stats_4k.md; tail -1 $OUT/bench_4k.json | cut -c1-300
  find $OUT/trace4k -name "*.csv" -size +5M -delete
fi
if has pmcsplit; then
  bash tools/pmc_split.sh $OUT/pmc_split > $OUT/pmc_split.log 2>&1; tail -16 $OUT/pmc_split.log
  cp $OUT/pmc_split/conv_traffic_split.json $OUT/conv_traffic_split.json 2>/dev/null
fi
if has trace8seg; then
  rm -rf $OUT/trace_8seg
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_8seg -o t -- python tools/bench_line.py 8seg split_all $OUT/line_8seg_split_all.json > $OUT/trace_8seg.log 2>&1; echo "trace8seg exit $?"
  python tools/trace_top.py $OUT/trace_8seg 45 "rocprofv3 --kernel-trace -- python tools/bench_line.py 8seg split_all (8-segment 1080p clip, --f16_split --f16_split_key_encoder; recording pass + warm-up + timed + event-timed replay)" > $OUT/kernel_stats_8seg_split_all.md
  head -14 $OUT/kernel_stats_8seg_split_all.md
  find $OUT/trace_8seg -name "*.csv" -size +5M -delete
fi
