"""One clip on several GPUs (BASELINE config 5, SURVEY.md §8e "replicate bank, shard queries"):
`MemoryManager.shard_queries` partitions the memory read by query column, all-gathers the read-out
columns and all-reduces the fixed-point usage counters.  Exercised here with 2 and 3 CPU processes
over gloo (3 ranks -> ragged column split), the HIP ops replaced by their PyTorch emulation; on a GPU
node the same code runs over RCCL.  The sharded run must be bit-identical to the unsharded one --
outputs of every frame AND the usage statistics that drive consolidation / eviction."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

# The HIP kernels compute every query column independently of the others (bit-identical under any
# column split, tests/test_gpu_d_affinity.py); the CPU emulation goes through BLAS, whose blocking --
# and with it the last bits of a column -- depends on how many columns a call has.
TOL = 2e-5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _Patch:
    """the part of pytest's monkeypatch that emu_ops.install uses"""

    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


def _worker(rank, world, port, name, out):
    try:
        _run(rank, world, port, name, out)
    except Exception:  # report instead of leaving the parent waiting on the queue
        import traceback
        out.put((rank, 'error', traceback.format_exc(), 0))


def _run(rank, world, port, name, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, 'tracking-anything-with-deva_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    torch.set_grad_enabled(False)
    torch.set_num_threads(2)
    import emu_ops
    import scenarios
    from workload import synth, weights
    emu_ops.install(_Patch)
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.model.network import DEVA
    with open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_spec.json')) as f:
        spec = json.load(f)
    sd = weights.make_state_dict([(k, tuple(s), getattr(torch, d)) for k, s, d in spec['tensors']], seed=0)
    net = DEVA(synth.base_config())
    net.load_weights(sd)
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    sc = dict(scenarios.E2E[name])
    sc['frames'] = min(sc['frames'], 22)

    def state(core):
        mem = core.memory
        st = {}
        for b in mem.work_mem.buckets:
            st[f'work{b}'] = mem.work_mem.size(b)
            if mem.use_long_term:
                st[f'use{b}'] = mem.work_mem.get_usage(b).clone()
                if mem.long_mem.engaged(b):
                    st[f'long{b}'] = mem.long_mem.size(b)
                    st[f'lkey{b}'] = mem.long_mem.key[b].clone()
        return st

    plain, core = scenarios.run_scenario(lambda cfg: DEVAInferenceCore(net, cfg), sc)
    want = state(core)

    def make_sharded(cfg):
        c = DEVAInferenceCore(net, cfg)
        c.memory.shard_queries()
        return c

    got_out, core = scenarios.run_scenario(make_sharded, sc)
    got = state(core)
    d_out = max((a - b).abs().max().item() for a, b in zip(plain, got_out))
    d_state = 0.0
    same_sizes = want.keys() == got.keys()
    for k in want:
        if not same_sizes:
            break
        if torch.is_tensor(want[k]):
            same_sizes = same_sizes and want[k].shape == got[k].shape
            if same_sizes:
                d_state = max(d_state, ((want[k] - got[k]).abs() / (1 + want[k].abs())).max().item())
        else:
            same_sizes = same_sizes and want[k] == got[k]
    out.put((rank, d_out, d_state if same_sizes else float('inf'), len(plain)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,name', [(2, 'two_buckets'), (3, 'no_lt')])
def test_query_sharded_read_is_bit_identical(world, name):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        r = q.get(timeout=900)
        if r[1] == 'error':
            for p in procs:
                p.terminate()
            pytest.fail(f'rank {r[0]} failed:\n{r[2]}')
        res.append(r)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    print(sorted(res))
    for rank, d_out, d_state, frames in sorted(res):
        assert frames > 0
        assert d_out <= TOL, f'rank {rank}: sharded outputs differ from the unsharded run by {d_out:.3e}'
        assert d_state <= TOL, f'rank {rank}: memory state (sizes / usage / long-term keys) differs by {d_state:.3e}'
    # every rank holds the same replica
    assert len({(r[1], r[2]) for r in res}) == 1
