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
