#!/bin/bash
# gpurun_out/r05 (scratch, merged back from the GPU box) -> profiles/r05 (tracked): the summaries the docs quote
cd "$(dirname "$0")/.."
S=gpurun_out/r05; D=profiles/r05
mkdir -p $D/tests $D/tests_split $D/lab
for f in bench.json bench.err conv_layers_480p5.json kernel_stats.md kernel_stats_4k.md kernel_stats_8seg_split_all.md bench_4k.json \
         conv_traffic.json affinity_read.json effective_clock.json effective_clock.txt line_8seg_split.txt line_8seg_split_all.txt; do
  [ -f $S/$f ] && cp $S/$f $D/$f
done
for t in tests tests_split; do
  [ -f $S/$t/summary.txt ] && cp $S/$t/summary.txt $D/$t/summary.txt
  for f in $S/$t/*.log; do
    [ -f "$f" ] || continue
    b=$(basename $f .log)
    if [ $(stat -c %s $f) -gt 200000 ]; then tail -c 150000 $f > $D/$t/$b.tail.log; else cp $f $D/$t/$b.log; fi
  done
done
for f in gpurun_out/lab/split_*.txt gpurun_out/lab/amp_v4.txt gpurun_out/lab/pmc_split/table.txt gpurun_out/lab/pmc_split_final/table.txt gpurun_out/lab/pmc_split_final_zero/table.txt; do
  [ -f $f ] && cp $f $D/lab/$(basename $(dirname $f))_$(basename $f)
done
ls -la $D $D/tests $D/tests_split $D/lab | head -80
du -sh $D
