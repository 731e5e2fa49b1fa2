cd /root/repo
mkdir -p gpurun_out/r04
python __graft_entry__.py > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_a_conv.py tests/test_gpu_e_network.py -m gpu -q -s -k "amp" > gpurun_out/r04/test_amp.log 2>&1; tail -3 gpurun_out/r04/test_amp.log; grep "amp lockstep\|amp vs\|FAILED\|Error" gpurun_out/r04/test_amp.log | cut -c1-700
python - <<'PY'
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tracking-anything-with-deva_amd')
import torch, bench
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
net, _ = bench.build_network(dev, amp=True)
fps, state = bench.run_1080p_segments(net, dev, steps=25, warmup=6, segments=8, conv_roofline=True)
print('amp 8-seg 1080p FPS', fps, json.dumps(state['conv_roofline']), state['objects_per_timed_frame'])
net32, _ = bench.build_network(dev)
fps32, state32 = bench.run_1080p_segments(net32, dev, steps=25, warmup=6, segments=8, conv_roofline=True)
print('fp32 8-seg 1080p FPS', fps32, json.dumps(state32['conv_roofline']))
PY
