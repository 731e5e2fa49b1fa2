cd /root/repo
mkdir -p gpurun_out/r04
python __graft_entry__.py > /dev/null 2>&1
timeout 1700 python -m pytest tests/test_gpu_g_fullsize.py -m gpu -q -s -k "free_running or (affinity and 50000)" > gpurun_out/r04/test_gpu_g_4k.log 2>&1; tail -3 gpurun_out/r04/test_gpu_g_4k.log; grep "clip:\|FAILED\|Error\|affinity N" gpurun_out/r04/test_gpu_g_4k.log | cut -c1-500
