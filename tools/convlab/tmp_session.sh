cd /root/repo
timeout 600 python -m pytest tests/test_gpu_e_network.py -m gpu -x -q -k "prefetch" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_a_conv.py tests/test_gpu_b_pointwise.py tests/test_gpu_d_affinity.py -m gpu -x -q 2>&1 | tail -2
python tools/host_profile_480p1.py 80 2>&1 | grep "free-running"
