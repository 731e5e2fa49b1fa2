cd /root/repo
mkdir -p gpurun_out/r04
free -g | head -2; nproc
python __graft_entry__.py > /dev/null 2>&1
DEVA_TEST_EVERY_QUERY=1 timeout 900 python -m pytest tests/test_gpu_g_fullsize.py -m gpu -q -s -k "affinity_at_bench_shapes and 83440" > gpurun_out/r04/test_affinity_every_query_83440x8160.log 2>&1; tail -4 gpurun_out/r04/test_affinity_every_query_83440x8160.log
