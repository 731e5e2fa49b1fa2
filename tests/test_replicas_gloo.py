"""N>1 path of bench.py (BASELINE config 4: independent clips, one per GPU -- replicas only, the
collective is used for the barrier and the max-over-ranks time).  Exercised here with 2 CPU
processes over gloo; on the GPU box the same code runs over RCCL."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    # rank r "propagates" for 0.05*(r+1) s: the job time must be the slowest rank's on every rank
    elapsed = bench.timed_region(lambda: time.sleep(0.05 * (rank + 1)), dist, 'cpu')
    fps = bench.whole_job_fps(10, world, elapsed)
    out.put((rank, elapsed, fps))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_report_the_slowest_rank():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, e0, f0), (_, e1, f1) = res
    assert abs(e0 - e1) < 1e-9, 'both ranks must agree on the job time (max over ranks)'
    assert 0.09 <= e0 < 0.5, e0
    assert abs(f0 - world * 10 / e0) < 1e-9 and f0 == f1


def test_single_process_path_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import bench
    e = bench.timed_region(lambda: time.sleep(0.01), None, 'cpu')
    assert 0.009 < e < 0.2


def bench_contract_keys():
    sys.path.insert(0, ROOT)
    import bench
    return bench.CONTRACT_KEYS


def test_compact_line_of_a_full_size_record_fits_the_limit():
    """the record of round 5 (26.7 KB as ONE line, which the driver could not parse) through compact_line: <= 4 KB,
    valid JSON, headline + roofline + cpu_baseline kept"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, 'profiles', 'r05', 'bench.json')) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    text = bench.compact_line(full)
    assert len(text) <= bench.LINE_LIMIT == 4096 and '\n' not in text
    line = json.loads(text)
    assert line['value'] == float(f"{full['value']:.6g}") and line['metric'] == full['metric']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in line['roofline'], k
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in line['cpu_baseline'], k
    assert all(not isinstance(v, (dict, list)) for v in line['roofline'].values())
    # a pathological record still fits: the optional scalars are dropped, never the contract keys
    full['config'].update({f'fps_extra_{i}': 1.0 * i for i in range(400)})
    text = bench.compact_line(full)
    assert len(text) <= 4096 and json.loads(text)['roofline']['frac'] == line['roofline']['frac']


def test_bench_gpus_2_launches_two_ranks(tmp_path):
    """`python bench.py --gpus 2` -- the driver's form, no torchrun environment -- must become two ranks
    (VERDICT r2 missing 1: the flag used to be parsed and ignored).  Here on CPU: gloo + the emulated ops
    (DEVA_BENCH_EMULATED=1), a tiny frame; on the GPU box the same launch path runs one rank per GPU on RCCL."""
    import json
    import subprocess
    env = dict(os.environ, DEVA_BENCH_EMULATED='1', OMP_NUM_THREADS='2', DEVA_BENCH_EXTRA_DIR=str(tmp_path))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                          '--height', '96', '--width', '128', '--objects', '2', '--no_cpu_baseline'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # the contract line is the LAST line of stdout, one compact JSON object (a 27 KB line was not parsed)
    last = out.stdout.rstrip('\n').splitlines()[-1]
    assert len(last) <= 4096, len(last)
    line = json.loads(last)
    assert set(line) <= set(bench_contract_keys()) | {'extra'}, sorted(line)
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['scaling'] == 'weak'
    assert line['value'] > 0 and abs(line['value'] - 2 * 2 / (line['ms_per_step'] * 2 * 1e-3)) < 1e-4 * line['value']
    assert line['config']['fps_4k_one_clip_on_all_gpus'] > 0 and line['config']['collective_bytes_per_frame_rank0'] > 0
    # the full record is in the side file
    with open(tmp_path / 'bench_extra.json') as f:
        full = json.load(f)
    one_clip = full['also_multi_gpu'][0]
    assert one_clip['scaling'] == 'strong' and one_clip['config']['collective_bytes_per_frame_rank0'] > 0
    assert abs(full['value'] - line['value']) < 1e-4 * full['value']


def test_bench_long4k_bank_mode_on_two_ranks():
    """`bench.py --workload long4k --long4k_mode bank --gpus 2`: ONE clip on two ranks, token-sharded read over a
    value-sharded bank (BASELINE configs[4]); launch path, collectives and the JSON line on CPU / gloo, tiny frame"""
    import json
    import subprocess
    env = dict(os.environ, DEVA_BENCH_EMULATED='1', OMP_NUM_THREADS='2')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '12', '--warmup', '2',
                          '--workload', 'long4k', '--long4k_mode', 'bank'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] > 0
    assert line['config']['collective_bytes_per_frame_rank0'] > 0
    assert 'token-sharded' in line['config']['parallelism']
