cd /root/repo
timeout 900 python -m pytest tests/test_gpu_d_affinity.py tests/test_gpu_f_memory_events.py tests/test_gpu_c_bank.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tracking-anything-with-deva_amd')
import torch, bench
dev = torch.device('cuda:0')
for r in bench.readout_roofline(dev, '480p5', 5, 1620, 16200) + bench.readout_roofline(dev, '1080p', 1, 8160, 18160)+ bench.readout_roofline(dev, '1080p full bank', 1, 8160, 83440):
    print('%-60s %-8s %8.1f us %7.0f GB/s %.3f' % (r['kernel'][:60], r['shape'], r.get('us', 0), r.get('gbps', 0), r.get('frac_of_hbm_peak', 0)), r.get('error',''))
PY
