#!/usr/bin/env python
"""Propagation-FPS benchmark of the MI355X-native DEVA hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one propagated frame: `DEVAInferenceCore.step(image)` = key encoder -> memory affinity
+ readout -> mask decoder (-> value encoder + memory append on every `mem_every`-th frame), timed
exactly like evaluation/eval_vos.py:150-186 but over the whole K-frame region.

Workload at N=1 (config.workload): BASELINE.json configs[1] -- DAVIS-2017-style 480p (854x480 ->
padded 480x864), 5 objects, working memory only, synthetic temporally-coherent frames, recipe
weights (the checkpoint is a download; workload/weights.py).  Frames are resident in HBM before the
timed region.  N>1: independent clips, one per GPU (replicas; RCCL is used only for the barrier
and the max-over-ranks reduction) -> "scaling": "weak".

Extra objects on the JSON line:
  roofline      dominant kernel (conv implicit GEMM, fp32 MFMA): algorithmic FLOPs of every launch
                / its HIP-event duration, measured in an instrumented pass right after the timed
                region (so event overhead never touches `value`); peak = 157.3 TFLOP/s fp32 matrix.
  affinity      the north-star kernel (fused similarity/top-k/softmax): event-timed at the
                BASELINE shape (N=10 000 bank, 1080p queries), reported against the fp32-MFMA roof
                that binds it and as HBM GB/s on algorithmic and on materialised-equivalent bytes
                (SURVEY.md §8d asks for all three).
  cpu_baseline  the CPU oracle (port of the reference's PyTorch path) on the same workload, on this
                box's host cores, for a bounded sample of frames.
  extra         1080p / 10k-token long-term bank propagation FPS (BASELINE target line) and the
                4K / 50k-token bank FPS of configs[4] on this GPU.

`--workload long4k` runs BASELINE configs[4] instead: ONE 2160x3840 clip with a 50 000-token
long-term bank on N GPUs -- every rank steps the same clip (replicated bank), the memory read is
sharded by query column with an RCCL all-gather of the read-out columns and an all-reduce of the
usage counters (MemoryManager.shard_queries) -> "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tracking-anything-with-deva_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md, dense fp32-in MFMA
PEAK_HBM_GBPS = 8000.0


def build_network(device):
    from workload import synth, weights
    from deva.model.network import DEVA
    with open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_spec.json')) as f:
        spec = json.load(f)['tensors']
    sd = weights.make_state_dict([(k, tuple(s), getattr(torch, d)) for k, s, d in spec], seed=0)
    net = DEVA(synth.base_config())
    net.load_weights(sd)
    return net.to(device).eval(), sd


def make_clip(height, width, n_frames, seed, device):
    from workload import synth
    stream = synth.FrameStream(height, width, seed=seed)
    return [stream.next().to(device) for _ in range(n_frames)]


def start_clip(net, cfg, frames, num_objects, device, lt_prefill=0, shard=False):
    """annotated first frame (+ optional pre-filled long-term bank)"""
    from workload import synth
    from deva.inference.inference_core import DEVAInferenceCore
    core = DEVAInferenceCore(net, cfg)
    if shard:
        core.memory.shard_queries()
    h, w = frames[0].shape[-2:]
    mask = synth.box_mask(h, w, num_objects).to(device)
    core.step(frames[0], mask, list(range(1, num_objects + 1)))
    if lt_prefill:
        g = torch.Generator().manual_seed(1)
        key = torch.randn(64, lt_prefill, generator=g).to(device)
        shr = (torch.rand(1, lt_prefill, generator=g) + 1).to(device)
        vals = {o: torch.randn(512, lt_prefill, generator=g).to(device) for o in range(1, num_objects + 1)}
        core.memory.long_mem.add(key, vals, shr, selection=None, supposed_bucket_id=0)
    return core


class ConvTimer:
    """HIP-event timing of every deva_conv2d launch.  The events are recorded on torch's current
    stream (the stream the kernels are launched on) immediately around the C call, with a device
    sync before each launch so that the interval is the kernel(s) of that launch and not host-side
    gaps (the un-instrumented frame loop is GPU-bound, the instrumented one would not be)."""

    def __init__(self):
        from deva.hip import lib
        self.handle = lib()
        self.real = self.handle.deva_conv2d
        self.records = []

    def __enter__(self):
        def timed(desc_ref, stream):
            d = desc_ref._obj if hasattr(desc_ref, '_obj') else desc_ref
            oh = (d.height + 2 * d.pad - d.kh) // d.stride + 1
            ow = (d.width + 2 * d.pad - d.kw) // d.stride + 1
            flops = 2.0 * d.cout * (d.c0 + d.c1) * d.kh * d.kw * d.batch * oh * ow
            sig = (d.c0 + d.c1, d.cout, d.kh, d.stride, d.batch, oh, ow)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = self.real(desc_ref, stream)
            e.record()
            self.records.append((flops, s, e, sig))
            return rc

        self.handle.deva_conv2d = timed
        return self

    def __exit__(self, *a):
        self.handle.deva_conv2d = self.real
        torch.cuda.synchronize()

    def summary(self):
        flops = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return flops, ms, len(self.records)

    def per_layer(self, frames):
        """time and achieved TFLOP/s per distinct (cin, cout, k, stride, batch, OH, OW)"""
        agg = {}
        for fl, s, e, sig in self.records:
            a = agg.setdefault(sig, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += fl
            a[2] += s.elapsed_time(e)
        rows = [dict(cin=k[0], cout=k[1], k=k[2], stride=k[3], batch=k[4], oh=k[5], ow=k[6],
                     calls_per_frame=v[0] / frames, ms_per_frame=v[2] / frames, gflop_per_call=v[1] / v[0] / 1e9,
                     tflops=v[1] / (v[2] * 1e-3) / 1e12) for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r['ms_per_frame'])


def affinity_microbench(device, n=10000, hw=8160, k=30, iters=20):
    from deva.hip import ops
    g = torch.Generator().manual_seed(0)
    key = torch.randn(n, 64, generator=g).to(device)
    shr = (torch.rand(n, generator=g) + 1).to(device)
    qk = torch.randn(64, hw, generator=g).to(device)
    qe = torch.rand(64, hw, generator=g).to(device)
    fix = torch.zeros(n, dtype=torch.int64, device=device)
    L = __import__('deva.hip', fromlist=['lib']).lib()
    splits = L.deva_affinity_default_splits(n, hw)
    part = torch.empty((L.deva_affinity_workspace(hw, k, splits),), dtype=torch.int64, device=device)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        ops.affinity_topk(None, None, 0, key, shr, n, qk, qe, k, fix)
    torch.cuda.synchronize()
    t_main = t_fin = 0.0
    idx = torch.empty((hw, k), dtype=torch.int32, device=device)
    wgt = torch.empty((hw, k), dtype=torch.float32, device=device)
    for _ in range(iters):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        L.deva_affinity_topk(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(), hw, k,
                             splits, part.data_ptr(), st)
        e1.record()
        L.deva_affinity_finalize(part.data_ptr(), hw, k, splits, idx.data_ptr(), wgt.data_ptr(), fix.data_ptr(), st)
        e2.record()
        torch.cuda.synchronize()
        t_main += e0.elapsed_time(e1)
        t_fin += e1.elapsed_time(e2)
    t_main, t_fin = t_main / iters * 1e-3, t_fin / iters * 1e-3
    flops = 4.0 * 64 * n * hw
    b_alg = 4.0 * (64 * n + n + 2 * 64 * hw) + 8.0 * k * hw + 4.0 * n
    b_mat = 12.0 * 4 * n * hw
    t = t_main + t_fin
    return dict(shape=dict(n=n, hw=hw, k=k, splits=splits), us_topk=t_main * 1e6, us_finalize=t_fin * 1e6,
                bound='mfma_fp32', achieved_tflops=flops / t / 1e12, peak_tflops=PEAK_FP32_MATRIX_TFLOPS,
                frac=flops / t / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                hbm_algorithmic_gbps=b_alg / t / 1e9, hbm_algorithmic_frac=b_alg / t / 1e9 / PEAK_HBM_GBPS,
                hbm_materialised_equiv_gbps=b_mat / t / 1e9)


def cpu_baseline(sd, cfg, height, width, num_objects, frames_cpu):
    from oracle import deva_oracle as O
    from workload import synth
    core = O.OracleCore(sd, cfg)
    mask = synth.box_mask(height, width, num_objects)
    core.step(frames_cpu[0], mask, list(range(1, num_objects + 1)))
    t0 = time.perf_counter()
    for f in frames_cpu[1:]:
        core.step(f)
    dt = time.perf_counter() - t0
    n = len(frames_cpu) - 1
    return dict(value=n / dt, unit='frames/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{n} propagated frames after the annotated one, same workload (CPU oracle, fp32)')


def timed_region(fn, dist=None, device=None):
    """barrier + device sync, run fn, barrier + device sync; returns the MAX elapsed seconds over all
    ranks (replicas: the job is as slow as its slowest clip).  `dist` is torch.distributed when
    world_size > 1 (RCCL on the GPU box, gloo in the CPU test), else None."""
    def fence():
        if dist is not None:
            dist.barrier()
        if device is not None and torch.device(device).type == 'cuda':
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    fn()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def whole_job_fps(steps_per_rank: int, world: int, elapsed: float) -> float:
    """weak scaling over independent clips: every rank propagates `steps_per_rank` frames"""
    return world * steps_per_rank / elapsed


def run_long4k(net, device, steps, warmup, seed, shard, dist):
    """BASELINE configs[4]: one 4K clip, 1 object, 50 000-token long-term bank.  With `shard` the
    memory read is partitioned by query column over the process group (every rank steps the clip)."""
    from workload import synth
    cfg = synth.base_config(max_long_term_elements=50000)
    n_frames = 1 + warmup + steps
    frames = make_clip(2160, 3840, n_frames, seed=seed, device=device)
    core = start_clip(net, cfg, frames, 1, device, lt_prefill=50000 - cfg['num_prototypes'], shard=shard)
    for t in range(1, 1 + warmup):
        core.step(frames[t])

    def timed_steps():
        for t in range(1 + warmup, n_frames):
            core.step(frames[t])

    elapsed = timed_region(timed_steps, dist, device)
    mem = core.memory
    bank = {'long': {b: mem.long_mem.size(b) for b in mem.long_mem.buckets},
            'work': {b: mem.work_mem.size(b) for b in mem.work_mem.buckets}}
    return steps / elapsed, bank


def long4k(args, net, rank, world, device, dist):
    """`--workload long4k`: strong scaling of ONE clip over the GPUs of the node"""
    fps, bank = run_long4k(net, device, args.steps, args.warmup, seed=11, shard=dist is not None, dist=dist)
    if rank == 0:
        print(json.dumps({
            'metric': 'propagation FPS @4K (1 object, 50k-token long-term bank)',
            'value': fps, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 / fps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[4]: one synthetic 3840x2160 clip, 1 object, long-term memory '
                                   'pre-filled to 50 000 tokens; bank replicated, memory read sharded by query column',
                       'bank_tokens_at_end': bank,
                       'parallelism': f'query-sharded memory read x{world} (all-gather read-out, all-reduce usage)'},
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    ap.add_argument('--objects', type=int, default=5)
    ap.add_argument('--workload', choices=['clips', 'long4k'], default='clips')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_extra', action='store_true')
    args = ap.parse_args()

    torch.set_grad_enabled(False)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    distributed = world > 1
    device = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(device)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl')  # RCCL on ROCm

    from workload import synth
    net, sd = build_network(device)
    if args.workload == 'long4k':
        return long4k(args, net, rank, world, device, dist if distributed else None)
    cfg = synth.base_config(enable_long_term=False, enable_long_term_count_usage=False)
    n_frames = 1 + args.warmup + args.steps
    frames = make_clip(args.height, args.width, n_frames, seed=100 + rank, device=device)  # HBM resident

    core = start_clip(net, cfg, frames, args.objects, device)
    for t in range(1, 1 + args.warmup):
        core.step(frames[t])

    def timed_steps():
        for t in range(1 + args.warmup, n_frames):
            core.step(frames[t])

    elapsed = timed_region(timed_steps, dist if distributed else None, device)
    bank = {b: core.memory.work_mem.size(b) for b in core.memory.work_mem.buckets}

    result = {
        'metric': 'propagation FPS @480p (5 objects, working memory only)',
        'value': whole_job_fps(args.steps, world, elapsed),
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': f'BASELINE configs[1]: DAVIS-2017-style {args.width}x{args.height} clip, {args.objects} objects, '
                        f'working memory only (mem_every=5, top_k=30), recipe weights, one clip per GPU',
            'frame_padded': [frames[0].shape[-2] + (-frames[0].shape[-2]) % 16, frames[0].shape[-1] + (-frames[0].shape[-1]) % 16],
            'bank_tokens_at_end': bank,
            'parallelism': f'replicas x{world}',
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, instrumented pass continuing the same clip
        extra_frames = make_clip(args.height, args.width, 5, seed=999, device=device)
        with ConvTimer() as ct:
            for f in extra_frames:
                core.step(f)
        flops, ms, launches = ct.summary()
        if os.environ.get('DEVA_BENCH_LAYERS'):
            with open(os.environ['DEVA_BENCH_LAYERS'], 'w') as f:
                json.dump(ct.per_layer(len(extra_frames)), f, indent=1)
        ach = flops / (ms * 1e-3) / 1e12
        result['roofline'] = {
            'kernel': 'conv_igemm_kernel (deva_conv2d, fp32 MFMA implicit GEMM)',
            'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP32_MATRIX_TFLOPS, 'unit': 'TFLOP/s',
            'frac': ach / PEAK_FP32_MATRIX_TFLOPS, 'traffic': None,
            'launches_per_frame': launches / len(extra_frames),
            'gflop_per_frame': flops / len(extra_frames) / 1e9,
            'ms_in_kernel_per_frame': ms / len(extra_frames),
        }
        # HBM traffic of the heaviest launch of that kernel (3x3 256->256 on the 1/4-resolution map), from
        # the committed rocprofv3 --pmc passes (separate runs; FETCH_SIZE doubled per the gfx950 note of
        # MI355X_MICROARCH.md).  `traffic` stays null: bench.py cannot collect PMC counters itself.
        pmc = os.path.join(ROOT, 'profiles', 'pmc_r01', 'rowconv_per_launch.json')
        if os.path.exists(pmc):
            with open(pmc) as f:
                d = json.load(f)
            rd = [v for k, v in d.items() if k.startswith('hbm_read_bytes')][0]
            wr = [v for k, v in d.items() if k.startswith('hbm_write_bytes')][0]
            alg = [v for k, v in d.items() if k.startswith('algorithmic_bytes')][0]
            result['roofline']['pmc_heaviest_launch'] = {
                'source': 'profiles/pmc_r01/rowconv_per_launch.json', 'hbm_bytes': rd + wr, 'algorithmic_bytes': alg,
                'mfma_util': d['mfma_util_frac'], 'l2_hit_rate': d['l2_hit_rate'],
                'clock_adjusted_peak_tflops': d['clock_adjusted_peak_tflops']}
        result['affinity'] = affinity_microbench(device)
        pmc_aff = os.path.join(ROOT, 'profiles', 'pmc_r01', 'affinity_per_launch.json')
        if os.path.exists(pmc_aff):
            with open(pmc_aff) as f:
                d = json.load(f)['10k']
            result['affinity']['pmc'] = {'source': 'profiles/pmc_r01/affinity_per_launch.json', 'mfma_util': d['mfma_util_frac'],
                                         'valu_instructions_per_tile': d['per_wave_per_tile']['SQ_INSTS_VALU'],
                                         'mfma_instructions_per_tile': d['per_wave_per_tile']['SQ_INSTS_MFMA'],
                                         'l2_hit_rate': d['l2_hit_rate']}
        if not args.no_extra:
            cfg_lt = synth.base_config()
            f1080 = make_clip(1080, 1920, 12, seed=7, device=device)
            core2 = start_clip(net, cfg_lt, f1080, 1, device, lt_prefill=10000)
            core2.step(f1080[1])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for f in f1080[2:]:
                core2.step(f)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            result['extra'] = {
                'fps_1080p_1obj_10k_longterm_bank': (len(f1080) - 2) / dt,
                'note': 'BASELINE target line: 1920x1080 (padded 1088x1920), 1 object, long-term memory '
                        'pre-filled with 10 000 tokens + working memory, 10 propagated frames',
            }
            del core2, f1080
            fps4k, bank4k = run_long4k(net, device, steps=6, warmup=2, seed=11, shard=False, dist=None)
            result['extra']['fps_4k_1obj_50k_longterm_bank'] = fps4k
            result['extra']['note_4k'] = ('BASELINE configs[4] on one GPU: 3840x2160, 1 object, long-term memory '
                                          f'pre-filled with 50 000 tokens, bank at end {bank4k}; 6 propagated frames')
        if not args.no_cpu_baseline and world == 1:
            n_cpu = 4
            frames_cpu = [f.cpu() for f in frames[:1 + n_cpu]]
            result['cpu_baseline'] = cpu_baseline(sd, cfg, args.height, args.width, args.objects, frames_cpu)
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
