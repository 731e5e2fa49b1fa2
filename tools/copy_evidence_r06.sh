#!/bin/bash
# copy what is judged from gpurun_out/r06 (scratch) into profiles/r06 and profiles/pmc_r06 (tracked)
cd "$(dirname "$0")/.."
S=gpurun_out/r06
mkdir -p profiles/r06/tests profiles/pmc_r06
for f in bench.err bench.json bench_4k.json bench_extra.json conv_layers_480p5.json effective_clock.txt kernel_stats.md kernel_stats_4k.md kernel_stats_8seg_split_all.md line_8seg_split_all.json; do cp $S/$f profiles/r06/$f; done
cp $S/tests/*.log $S/tests/summary.txt profiles/r06/tests/
for f in affinity_read.json conv_traffic.json conv_traffic_split.json effective_clock.json; do cp $S/$f profiles/pmc_r06/$f; done
git status --short profiles | head -40
