cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o p1080 -- python bench.py --height 1080 --width 1920 --objects 1 --steps 10 --warmup 2 --no_cpu_baseline --no_extra > gpurun_out/p1080.log 2>&1
head -22 gpurun_out/prof/p1080_kernel_stats.csv | cut -c1-200
grep -o '"value": [0-9.]*' gpurun_out/p1080.log | head -1
