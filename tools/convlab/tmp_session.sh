cd /root/repo
timeout 900 python -m pytest tests/test_gpu_a_conv.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --no_cpu_baseline --no_extra --no_affinity > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_quick.json').read().strip().split('\n')[-1])
r=d['roofline']
print('FPS',d['value'],'conv TF',r['achieved'],'frac',r['frac'],'ms in kernel',r['ms_in_kernel_per_frame'],'probe',r['sustained_mfma_probe'].get('random_operands_tflops'), 'eval_vos-style', d['timed_like_eval_vos']['fps'])
PY
