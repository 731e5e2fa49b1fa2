#!/usr/bin/env python
"""the figures DESIGN.md section 7 / README quote, read from the round's evidence files:  python tools/r05_numbers.py [dir]"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r05'
b = json.loads(open(os.path.join(d, 'bench.json')).read().strip().split('\n')[-1])
r, c = b['roofline'], b['config']
print('headline %.1f FPS (%.2f ms), conv %.1f TF = %.3f of peak, ms in conv %.3f, traffic/alg %s (this build: %s)' % (
    b['value'], b['ms_per_step'], r['achieved'], r['frac'], r['ms_in_kernel_per_frame'], r.get('traffic_over_algorithmic'), r.get('traffic_measured_on_this_build')))
print('probe', r.get('sustained_mfma_probe', {}).get('random_operands_tflops'), 'frac_of_probe', r.get('frac_of_sustained_probe'))
print('eval_vos-style', b.get('timed_like_eval_vos', {}).get('fps'))
for k, v in c.items():
    if k.startswith('fps'):
        print('  ', k, v)
for k, v in r.items():
    if k.startswith(('affinity', 'f16_split')):
        print('  ', k, v)
for e in b.get('also', []):
    st = (e.get('config', {}).get('state_at_end') or {})
    cr = st.get('conv_roofline')
    print('also: %-100s %s' % (e['metric'][:100], None if e.get('value') is None else round(e['value'], 2)))
    if cr:
        for kk in ('split_kernels', 'f16_kernels', 'f32_kernels'):
            if kk in cr:
                x = cr[kk]
                print('        %-14s %.1f GF/frame %.2f ms/frame %s launches  %s' % (kk, x['gflop_per_frame'], x['ms_per_frame'], round(x['launches_per_frame'], 1),
                      {a: round(v, 3) for a, v in x.items() if a.startswith(('frac', 'fp32_equiv', 'tflops', 'vs_')) and v is not None}))
        print('        split_fallbacks', st.get('split_fallbacks'), 'objects', st.get('objects_per_timed_frame'))
a = b.get('affinity', {})
print('affinity', {k: a.get(k) for k in ('us_read', 'us_read_bank_operands_cached', 'us_read_fp32_kernels_only', 'f16_mfma_frac', 'hbm_algorithmic_frac', 'hbm_counter_traffic_over_algorithmic')})
cb = b.get('cpu_baseline', {})
print('cpu', {k: cb.get(k) for k in ('value', 'runs_fps', 'spread_frac', 'cores', 'kind')})
print('cpu stages', cb.get('stage_ms_per_frame'))
for row in cb.get('affinity_kernels', []) if isinstance(cb.get('affinity_kernels'), list) else []:
    print('   ', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items() if not k.endswith('range')})
for e in b.get('also_kernels', []) if isinstance(b.get('also_kernels'), list) else []:
    print('  kernel: %-60s %-40s %8.1f us %.3f' % (e['kernel'][:60], e['shape'][:40], e.get('us', 0), e.get('frac_of_hbm_peak', 0)))
