cd /root/repo
timeout 600 python -m pytest tests/test_gpu_d_affinity.py -m gpu -x -q -k "prefilter" 2>&1 | tail -5
