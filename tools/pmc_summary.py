#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (one run per counter group, tools/pmc_bench.sh / pmc_affinity.sh) into
the per-launch / per-frame JSON files committed under profiles/pmc_rNN/.

    python tools/pmc_summary.py conv  gpurun_out/pmc/bench  profiles/pmc_r02/conv_traffic.json
    python tools/pmc_summary.py aff   gpurun_out/pmc/aff    profiles/pmc_r02/affinity_per_launch.json

Counters are averaged per dispatch of a kernel over all its dispatches in a pass.  FETCH_SIZE / WRITE_SIZE
are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so read bytes =
FETCH_SIZE x 1024 x 2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE x 1024 is taken as is (uncalibrated)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def load(prefix):
    """-> {kernel name: {counter: [sum, dispatches]}}, {kernel name: [duration sum ns, dispatches]}"""
    counters = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    durations = defaultdict(lambda: [0.0, set()])
    for path in sorted(glob.glob(prefix + '_g*_counter_collection.csv')):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = re.sub(r'^void ', '', row['Kernel_Name']).replace('deva::(anonymous namespace)::', '')
                name = re.sub(r'[<(].*', '', name)
                c = counters[name][row['Counter_Name']]
                c[0] += float(row['Counter_Value'])
                c[1] += 1
                key = (path, row['Dispatch_Id'])
                if key not in durations[name][1]:
                    durations[name][1].add(key)
                    durations[name][0] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    return counters, {k: (v[0], len(v[1])) for k, v in durations.items()}


def _conv_sha1():
    """the same digest as bench.py:conv_source_sha1 (bench.py imports torch at module level: hashed here directly)"""
    import hashlib
    import os
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tracking-anything-with-deva_amd', 'csrc')
    for name in ('conv_args.h', 'conv_epilogue.h', 'conv_igemm.hip', 'conv_mfma.hip', 'conv_wino.hip', 'conv_cout1.hip', 'common.h'):
        with open(os.path.join(d, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def split(prefix, line_json, out):
    """HBM traffic per frame of the convolution kernels of the 8-segment 1080p clip under --f16_split --f16_split_key_encoder
    (tools/pmc_split.sh) next to the algorithmic bytes of the same clip (tools/bench_line.py: every operand of every
    deva_conv2d launch once)"""
    counters, durations = load(prefix)
    is_conv = lambda n: n.startswith(('conv_', 'splitk_reduce', 'stem7x7'))
    frames = max(counters.get('upsample4x_softmax_kernel', {}).get('FETCH_SIZE', [0, 0])[1], 1)  # one decoder pass per propagated frame
    rd = sum(c['FETCH_SIZE'][0] for n, c in counters.items() if is_conv(n) and 'FETCH_SIZE' in c) * 1024 * 2
    wr = sum(c['WRITE_SIZE'][0] for n, c in counters.items() if is_conv(n) and 'WRITE_SIZE' in c) * 1024
    with open(line_json) as f:
        line = json.load(f)
    alg = (line.get('state', {}).get('conv_roofline') or {}).get('algorithmic_bytes_per_frame')
    per_family = {}
    for n, c in counters.items():
        if is_conv(n):
            per_family[n] = {'read_bytes_per_frame': c.get('FETCH_SIZE', [0.0])[0] * 2048 / frames,
                             'write_bytes_per_frame': c.get('WRITE_SIZE', [0.0])[0] * 1024 / frames,
                             'dispatches_per_frame': max(v[1] for v in c.values()) / frames}
    res = {
        'command': 'DEVA_LINE_STEPS=10 DEVA_LINE_WARMUP=6 python tools/bench_line.py 8seg split_all (8-segment 1080p clip, '
                   '--f16_split --f16_split_key_encoder; recording pass + warm-up + timed + event-timed replay)',
        'conv_source_sha1': _conv_sha1(),
        'frames_in_a_pass': frames,
        'hbm_read_bytes_per_frame (FETCH_SIZE KiB x 1024 x 2, gfx950 correction)': rd / frames,
        'hbm_write_bytes_per_frame (WRITE_SIZE KiB x 1024)': wr / frames,
        'hbm_bytes_per_frame': (rd + wr) / frames,
        'algorithmic_bytes_per_frame': alg,
        'traffic_over_algorithmic': ((rd + wr) / frames / alg) if alg else None,
        'kernels': 'conv_f16_kernel*, stem7x7_kernel*, conv_mfma_kernel* (fp32 layers + the gated re-runs), splitk_reduce_kernel, '
                   'conv_cout1 kernels',
        'per_kernel_family': per_family,
    }
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != 'per_kernel_family'}, indent=1))


def conv(prefix, out):
    counters, durations = load(prefix)
    is_conv = lambda n: n.startswith('conv_') or n.startswith('splitk_reduce')  # conv_mfma, conv_igemm, conv_cout1, conv3x3_cout1_rows
    frames = counters.get('upsample4x_softmax_kernel', {}).get('FETCH_SIZE', [0, 0])[1]
    frames = max(frames, 1) + 1  # one decoder pass per propagated frame + the annotated first frame
    rd = sum(c['FETCH_SIZE'][0] for n, c in counters.items() if is_conv(n) and 'FETCH_SIZE' in c) * 1024 * 2
    wr = sum(c['WRITE_SIZE'][0] for n, c in counters.items() if is_conv(n) and 'WRITE_SIZE' in c) * 1024
    per_kernel = {}
    for n, c in counters.items():
        if not is_conv(n):
            continue
        per_kernel[n] = {k: v[0] / v[1] for k, v in c.items()}
        per_kernel[n]['dispatches_per_pass'] = max(v[1] for v in c.values())
    res = {
        'command': 'python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_extra --no_affinity (480p, 5 objects)',
        'conv_source_sha1': _conv_sha1(),
        'frames_in_a_pass': frames,
        'hbm_read_bytes_per_frame (FETCH_SIZE KiB x 1024 x 2, gfx950 correction)': rd / frames,
        'hbm_write_bytes_per_frame (WRITE_SIZE KiB x 1024)': wr / frames,
        'hbm_bytes_per_frame': (rd + wr) / frames,
        'kernels': 'conv_wino_kernel*, conv_mfma_kernel*, conv_igemm_kernel*, splitk_reduce_kernel, conv_cout1 kernels (everything deva_conv2d launches)',
        'per_dispatch_averages': per_kernel,
    }
    g = counters.get('conv_mfma_kernel', None) or next((c for n, c in counters.items() if n.startswith('conv_mfma')), {})
    if 'SQ_INSTS_MFMA' in g and 'SQ_VALU_MFMA_BUSY_CYCLES' in g and 'GRBM_GUI_ACTIVE' in g:
        # busy cycles are summed over the SIMDs: 1024 SIMDs x active cycles = 100 %
        res['conv_mfma_mfma_util_frac'] = (g['SQ_VALU_MFMA_BUSY_CYCLES'][0] / g['SQ_VALU_MFMA_BUSY_CYCLES'][1]) / (
            g['GRBM_GUI_ACTIVE'][0] / g['GRBM_GUI_ACTIVE'][1] / 8.0 * 1024)
    w = next((c for n, c in counters.items() if n.startswith('conv_wino')), {})
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in w and 'GRBM_GUI_ACTIVE' in w:
        res['conv_wino_mfma_util_frac'] = (w['SQ_VALU_MFMA_BUSY_CYCLES'][0] / w['SQ_VALU_MFMA_BUSY_CYCLES'][1]) / (
            w['GRBM_GUI_ACTIVE'][0] / w['GRBM_GUI_ACTIVE'][1] / 8.0 * 1024)
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != 'per_dispatch_averages'}, indent=1))


def aff(prefix, out):
    counters, durations = load(prefix)
    res = {}
    topk = [n for n in counters if n.startswith('affinity_topk_kernel')]
    # the microbench runs the two shapes one after the other with the same number of launches each (and
    # possibly the same grid size): split the kernel's dispatches of every pass into first / second half
    by_grid = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for path in sorted(glob.glob(prefix + '_g*_counter_collection.csv')):
        with open(path) as f:
            rows = [r for r in csv.DictReader(f) if 'affinity_topk' in r['Kernel_Name']]
        ids = sorted({int(r['Dispatch_Id']) for r in rows})
        first = set(ids[:len(ids) // 2])
        seen = set()
        for row in rows:
            g = 0 if int(row['Dispatch_Id']) in first else 1
            c = by_grid[g][row['Counter_Name']]
            c[0] += float(row['Counter_Value'])
            c[1] += 1
            if row['Dispatch_Id'] not in seen:
                seen.add(row['Dispatch_Id'])
                dur[g][0] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
                dur[g][1] += 1
    shapes = {'10k': (10000, 8160), '83k': (83440, 8160)}
    grids = sorted(by_grid)
    for tag, g in zip(shapes, grids):
        n, hw = shapes[tag]
        c = {k: v[0] / v[1] for k, v in by_grid[g].items()}
        tiles_per_wave = (n / 32.0) * (hw / 32.0) / max(c.get('SQ_WAVES', 1), 1)
        us = dur[g][0] / dur[g][1] / 1e3
        entry = {'shape': f'N={n} x HW={hw}', 'avg_duration_us': us, 'raw_per_dispatch': c}
        if 'GRBM_GUI_ACTIVE' in c:
            active = c['GRBM_GUI_ACTIVE'] / 8.0  # summed over the 8 XCDs
            entry['clock_GHz'] = active / (us * 1e3)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:  # busy cycles are summed over the 1024 SIMDs
                entry['mfma_util_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (active * 1024)
        if 'SQ_WAVES' in c:
            entry['per_wave_per_tile'] = {k: c[k] / c['SQ_WAVES'] / tiles_per_wave for k in
                                          ('SQ_INSTS_MFMA', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM')
                                          if k in c}
            if 'SQ_WAVE_CYCLES' in c:
                entry['wave_cycle_shares'] = {k: c[k] / c['SQ_WAVE_CYCLES'] for k in
                                              ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU')
                                              if k in c}
        if 'FETCH_SIZE' in c:
            entry['hbm_read_bytes (FETCH_SIZE KiB x 1024 x 2)'] = c['FETCH_SIZE'] * 2048
        if 'WRITE_SIZE' in c:
            entry['hbm_write_bytes (WRITE_SIZE KiB x 1024)'] = c['WRITE_SIZE'] * 1024
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            entry['hbm_bytes'] = c['FETCH_SIZE'] * 2048 + c['WRITE_SIZE'] * 1024
            entry['algorithmic_bytes'] = 4.0 * (64 * n + n + 2 * 64 * hw) + 8.0 * 30 * hw + 4.0 * n
        if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c:
            entry['l2_hit_rate'] = c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
        res[tag] = entry
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'raw_per_dispatch'} for k, v in res.items()}, indent=1))


def read(prefix, out):
    """per-dispatch averages of every kernel of deva_affinity_read at N = 10 000 x HW = 8 160 (tools/pmc_read.sh)"""
    counters, durations = load(prefix)
    n, hw = 10000, 8160
    res = {'shape': f'N={n} x HW={hw}, k=30', 'algorithmic_bytes': 4.0 * (64 * n + n + 2 * 64 * hw) + 8.0 * 30 * hw + 4.0 * n,
           'kernels': {}}
    for name in sorted(counters):
        if not name.startswith('affinity_'):
            continue
        c = {k: v[0] / v[1] for k, v in counters[name].items()}
        d_ns, d_n = durations[name]
        entry = {'avg_duration_us_under_counters': d_ns / d_n / 1e3, 'dispatches': d_n}
        if 'GRBM_GUI_ACTIVE' in c:
            active = c['GRBM_GUI_ACTIVE'] / 8.0  # summed over the 8 XCDs
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and active > 0:
                entry['mfma_util_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (active * 1024)
        if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES'] > 0:
            entry['wave_cycle_shares'] = {k: c[k] / c['SQ_WAVE_CYCLES'] for k in
                                          ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU')
                                          if k in c}
        if c.get('SQ_INSTS_MFMA', 0) > 0:
            entry['instructions_per_mfma'] = {k: c[k] / c['SQ_INSTS_MFMA'] for k in
                                              ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM') if k in c}
        if 'FETCH_SIZE' in c:
            entry['hbm_read_bytes (FETCH_SIZE KiB x 1024 x 2)'] = c['FETCH_SIZE'] * 2048
        if 'WRITE_SIZE' in c:
            entry['hbm_write_bytes (WRITE_SIZE KiB x 1024)'] = c['WRITE_SIZE'] * 1024
        if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
            entry['l2_hit_rate'] = c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
        entry['raw_per_dispatch'] = c
        res['kernels'][name] = entry
    # one read = one dispatch of every kernel of the pre-filter path + the two gated fall-back kernels; the fp32-only passes
    # of the same command (pre-filter forced off) dispatch affinity_topk* / affinity_finalize* with real work, so the sum
    # below takes the pre-filter kernels only and adds the cheapest observed fall-back dispatches separately
    pf = [k for k in res['kernels'] if k.startswith('affinity_pf_')]
    tot = lambda key: sum(res['kernels'][k].get(key, 0.0) for k in pf)
    res['prefilter_total'] = {'hbm_read_bytes': tot('hbm_read_bytes (FETCH_SIZE KiB x 1024 x 2)'),
                              'hbm_write_bytes': tot('hbm_write_bytes (WRITE_SIZE KiB x 1024)')}
    res['prefilter_total']['hbm_bytes'] = res['prefilter_total']['hbm_read_bytes'] + res['prefilter_total']['hbm_write_bytes']
    res['prefilter_total']['traffic_over_algorithmic'] = res['prefilter_total']['hbm_bytes'] / res['algorithmic_bytes']
    res['read_total'] = dict(res['prefilter_total'], note='FETCH_SIZE x 2 (gfx950) + WRITE_SIZE per dispatch, summed over the '
                             'affinity_pf_* kernels of one deva_affinity_read at this shape (what bench.py divides by the '
                             'algorithmic bytes)')
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    slim = {k: {kk: vv for kk, vv in v.items() if kk != 'raw_per_dispatch'} for k, v in res['kernels'].items()}
    print(json.dumps({'prefilter_total': res['prefilter_total'], 'kernels': slim}, indent=1))


def clock(probe_prefix, bench_prefix, out):
    """effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time, per kernel family: the register-only fp32 MFMA
    probe and the convolution kernels of the bench command; MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE / 8 x
    1024 SIMDs).  Every counter is averaged per dispatch over the passes that collected it (counters of one kernel may come
    from different passes of the same command)."""
    res = {'method': 'rocprofv3 --pmc GRBM_GUI_ACTIVE (+ SQ_VALU_MFMA_BUSY_CYCLES) --kernel-trace; clock = GUI_ACTIVE / 8 / (End - Start), '
                     'per-dispatch averages; the bench command is `python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_extra --no_affinity`'}
    for tag, prefix in (('probe', probe_prefix), ('bench', bench_prefix)):
        counters, durations = load(prefix)
        fam = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        dur = defaultdict(lambda: [0.0, 0])
        for name, c in counters.items():
            key = ('mfma_probe_kernel' if 'probe' in name else 'conv_mfma_kernel' if name.startswith('conv_mfma') else
                   'conv_wino_kernel' if name.startswith('conv_wino') else 'conv_f16_kernel' if name.startswith('conv_f16') else None)
            if key is None or 'GRBM_GUI_ACTIVE' not in c:
                continue
            for cn in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
                if cn in c:
                    fam[key][cn][0] += c[cn][0]
                    fam[key][cn][1] += c[cn][1]
            dur[key][0] += durations[name][0]
            dur[key][1] += durations[name][1]
        for key, cs in fam.items():
            gui = cs['GRBM_GUI_ACTIVE'][0] / max(cs['GRBM_GUI_ACTIVE'][1], 1) / 8.0  # cycles per dispatch
            ns = dur[key][0] / max(dur[key][1], 1)
            e = {'dispatches_per_pass': dur[key][1], 'avg_kernel_us': ns / 1e3, 'effective_clock_ghz': gui / ns}
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and cs['SQ_VALU_MFMA_BUSY_CYCLES'][1]:
                e['mfma_busy_frac'] = cs['SQ_VALU_MFMA_BUSY_CYCLES'][0] / cs['SQ_VALU_MFMA_BUSY_CYCLES'][1] / (gui * 1024)
                e['tflops_fp32_mfma_at_this_clock_and_busy_frac'] = 64.0 * 1024 * e['effective_clock_ghz'] * e['mfma_busy_frac'] / 1e3
            res[f'{tag}:{key}'] = e
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'clock':
        clock(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == 'split':
        split(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        {'conv': conv, 'aff': aff, 'read': read}[sys.argv[1]](sys.argv[2], sys.argv[3])
