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
                / its HIP-event duration.  The events are recorded on the launch stream around every
                deva_conv2d call WITHOUT any synchronisation, in a pass that replays the same number of
                frames right after the timed region (event overhead never touches `value`); the stream
                stays busy back to back, so an event pair brackets exactly the kernel(s) of its launch
                at the clocks of the real frame loop.  peak = 157.3 TFLOP/s fp32 matrix (data sheet);
                `sustained_mfma_probe` = what a register-only fp32 MFMA loop reaches on this box, timed right
                here (the chip clocks to its power budget: ~0.78 of the data sheet).  `traffic` =
                HBM bytes per frame of those kernels from the committed rocprofv3 --pmc passes over
                this same command (profiles/pmc_r06/conv_traffic.json, tools/pmc_bench.sh; bench.py cannot collect PMC
                counters itself), next to the algorithmic bytes per frame computed here.
  affinity      the north-star read (similarity -> exact top-k -> softmax -> usage): event-timed at the
                BASELINE shape (N=10 000 bank, 1080p queries) through deva_affinity_read (fp16 MFMA
                pre-filter + exact fp32 re-scoring) and through the fp32 kernels alone; reported as
                fp32-equivalent TFLOP/s against the fp32-MFMA roof (which the pre-filter is not bound
                by), as f16 MFMA TFLOP/s, and as HBM GB/s on algorithmic and on materialised-equivalent
                bytes (SURVEY.md §8d asks for all three).
  cpu_baseline  the CPU oracle (port of the reference's PyTorch path) on the same workload, on this
                box's host cores: median of three runs (warm-up frames first) with the spread, per-stage ms; the five
                kernel-only affinity shapes: one warm-up call + median of three.
                `config` / `roofline` also carry, as plain scalars, the lines a reader looks for first (`fps_1080p_1obj_10k_bank`,
                `fps_1080p_8seg*`, `fps_4k_*`, `affinity_read_us_*`, `f16_split_*`): a record that keeps only the keys of
                the contract still holds them; the full entries are in `also`.
  also          further lines of BASELINE.json's metric, each with its own warm-up and timed frames (incl. the headline,
                the north-star target line and the 8-segment clip with --f16_split: fp32-accurate convolutions on the f16
                matrix pipes, `dtype: "f32 via 3x f16 split, f32 acc"`, conv roofline against the f16 peak counting 3 MFMAs):
                1080p / detections every 5th frame / 10k-token long-term bank (BASELINE configs[2] and
                the north-star target line) and 4K / 50k-token bank (configs[4] on this one GPU).  Each
                names the -m gpu test that gates its parity.

`--gpus N` without a torchrun environment re-executes this script under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL); the line then carries
`rccl_ranks: N` and, in `also_multi_gpu`, the one-clip-on-N-GPUs line of BASELINE configs[4] (frame-owner mode) with
the bytes rank 0 moved per frame.  No scaling curve has been measured by the builder (one GPU per box).

`--workload long4k` runs BASELINE configs[4] instead: ONE 2160x3840 clip with a 50 000-token
long-term bank on N GPUs, `--long4k_mode`:
  owner    (default) rank 0 alone encodes / decodes; it broadcasts the query key / selection, every rank
           matches and reads out its share of the query columns against its replica of the bank, the
           read-out columns are gathered to rank 0, usage counters all-reduced, new memory rows broadcast;
  queries  every rank steps the whole clip, only the memory read is sharded by query column (all-gather);
  bank     memory read sharded by token range: per-shard top-k keys all-gathered and merged exactly,
           partial read-outs all-reduced; each rank stores 1/N of the value rows;
  owner_bank  the same sharded bank with rank 0 as the only encoder / decoder: partial read-outs reduced to it.
-> "scaling": "strong"; the line reports the bytes every rank moved per frame next to the FPS.
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


def conv_source_sha1():
    """identity of the convolution kernels' build: the PMC traffic file of a round records it"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, 'tracking-anything-with-deva_amd', 'csrc')
    for name in ('conv_args.h', 'conv_epilogue.h', 'conv_igemm.hip', 'conv_mfma.hip', 'conv_wino.hip', 'conv_cout1.hip', 'common.h'):
        with open(os.path.join(d, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()
PEAK_HBM_GBPS = 8000.0


def mfma_probe(device, iters=6000, reps=4):
    """what the fp32 matrix pipes sustain on THIS box (include/deva_hip.h: deva_probe_mfma_f32): a register-only
    v_mfma_f32_32x32x2_f32 loop, 4 waves per SIMD on every CU, timed with events after a warm-up; random operands (the
    switching activity of real data) and zeros.  The chip clocks to its power budget under dense MFMA work
    (MI355X_MICROARCH.md, DVFS give-back), so this -- not the data-sheet 157.3 -- is what a perfect kernel reaches."""
    from deva.hip import lib
    L = lib()
    out = {}
    sink = torch.empty(1 << 16, device=device)
    for name, ops_ in (('random_operands', torch.rand(1 << 16, device=device) * 2 - 1), ('zero_operands', torch.zeros(1 << 16, device=device))):
        st = torch.cuda.current_stream().cuda_stream
        flop = 0
        for _ in range(3):  # ~40 ms of warm-up: the clock settles within milliseconds
            flop = L.deva_probe_mfma_f32(ops_.data_ptr(), ops_.numel(), sink.data_ptr(), iters, st)
        if flop <= 0:
            raise RuntimeError('deva_probe_mfma_f32 failed')
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            L.deva_probe_mfma_f32(ops_.data_ptr(), ops_.numel(), sink.data_ptr(), iters, st)
        b.record()
        torch.cuda.synchronize()
        out[name + '_tflops'] = flop * reps / (a.elapsed_time(b) * 1e-3) / 1e12
    out['frac_of_peak'] = out['random_operands_tflops'] / PEAK_FP32_MATRIX_TFLOPS
    out['method'] = (f'deva_probe_mfma_f32: {reps} launches of {iters} x 16 back-to-back v_mfma_f32_32x32x2_f32 per wave, 4 waves per '
                     'SIMD, no memory traffic in the loop, events on the launch stream after 3 warm-up launches')
    return out


def build_network(device, amp=False, split=False, split_key_encoder=False):
    from workload import synth, weights
    from deva.model.network import DEVA
    with open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_spec.json')) as f:
        spec = json.load(f)['tensors']
    sd = weights.make_state_dict([(k, tuple(s), getattr(torch, d)) for k, s, d in spec], seed=0)
    net = DEVA(dict(synth.base_config(), amp=amp, f16_split=split, f16_split_key_encoder=split_key_encoder))
    net.load_weights(sd)
    return net.to(device).eval(), sd


def make_clip(height, width, n_frames, seed, device):
    from workload import synth
    stream = synth.FrameStream(height, width, seed=seed)
    return [stream.next().to(device) for _ in range(n_frames)]


def start_clip(net, cfg, frames, num_objects, device, lt_prefill=0, shard=None):
    """annotated first frame (+ optional pre-filled long-term bank); shard =
    None | 'owner' | 'queries' | 'bank' (one clip on all ranks of the process group)"""
    from workload import synth
    from deva.inference.inference_core import DEVAInferenceCore
    core = DEVAInferenceCore(net, cfg)
    if shard == 'bank':
        core.memory.shard_bank()
    elif shard == 'owner_bank':
        core.memory.shard_bank(owner=0)
    elif shard is not None:
        core.memory.shard_queries(owner=0 if shard == 'owner' else None)
    h, w = frames[0].shape[-2:]
    mask = synth.box_mask(h, w, num_objects).to(device)
    objects = list(range(1, num_objects + 1))
    core.step(frames[0], mask, objects)
    if lt_prefill:
        key, shr, vals = synth.prefill_bank(lt_prefill, objects, seed=1)
        core.memory.long_mem.add(key.to(device), {o: v.to(device) for o, v in vals.items()}, shr.to(device),
                                 selection=None, supposed_bucket_id=0)
    return core


class ConvTimer:
    """HIP-event timing of every deva_conv2d launch WITHOUT synchronisation: the two events are recorded
    on torch's current stream (the stream the kernels are launched on) immediately around the C call.
    The frame loop is GPU-bound (the stream stays busy back to back), so an event pair brackets exactly
    the kernel(s) of its launch -- the implicit-GEMM kernel and, for split-K layers, its reduction -- at the
    clocks of the real frame loop.  (Round 1 synchronised before every launch, which let the chip boost
    on big layers and charged launch latency to small ones.)"""

    def __init__(self, spacer=False):
        from deva.hip import lib
        self.spacer = spacer  # the `also` lines (frames down to 4 ms); the headline loop is GPU-bound and keeps the plain pairs
        self.handle = lib()
        self.real = self.handle.deva_conv2d
        self.records = []
        # A frame shorter than the host needs to issue it WITH this instrumentation (the split lines at 480p: 4.3 ms) leaves
        # the stream idle now and then, and an event recorded on an idle stream is stamped with the END of the last kernel
        # before the gap: the idle time landed in the next convolution's pair (round 6: 0.9 ms "convolutions" of 20 us).
        # A one-thread spacer launch right before the start event gives it a completion to be stamped with.
        self._spacer = torch.zeros(2, device='cuda')

    def _space(self, stream):
        if self.spacer:
            self.handle.deva_usage_init(self._spacer.data_ptr(), self._spacer.data_ptr() + 4, 1, stream)

    def __enter__(self):
        def timed(desc_ref, stream):
            d = desc_ref._obj if hasattr(desc_ref, '_obj') else desc_ref
            oh = (d.height + 2 * d.pad - d.kh) // d.stride + 1
            ow = (d.width + 2 * d.pad - d.kw) // d.stride + 1
            cin = d.c0 + d.c1
            flops = 2.0 * d.cout * cin * d.kh * d.kw * d.batch * oh * ow
            # algorithmic bytes: every operand once (a broadcast operand has batch stride 0)
            b0 = d.batch if d.in0_batch_stride else 1
            b1 = d.batch if d.in1_batch_stride else 1
            br = d.batch if d.residual_batch_stride else 1
            nbytes = 4.0 * (d.c0 * b0 * d.height * d.width + d.c1 * b1 * d.height * d.width
                            + cin * d.kh * d.kw * d.cout + d.cout * oh * ow * d.batch
                            + (d.cout * oh * ow * br if d.residual else 0) + (d.cout if d.bias else 0))
            sig = (cin, d.cout, d.kh, d.stride, d.batch, oh, ow)
            # the shapes csrc/conv_f16.hip takes when amp is requested (launch_conv_f16 + the vector-gather geometry)
            # (d.amp: 1 = fp16 operands, K steps of 64; 2 = hi/lo split, K steps of 32)
            gran = 64 if d.amp == 1 else 32
            tail_ok = d.amp == 2 and d.kh == 1 and (d.c0 % gran == 0 if d.c1 else True)  # split 1x1: partial last K step
            f16 = bool(d.amp and d.weight_f16 and d.stride == 1 and d.cout >= 64 and ((d.c0 % gran == 0 and d.c1 % gran == 0) or tail_ok)
                       and ((d.kh == 1 and d.pad == 0) or (d.kh == 3 and d.pad == 1)) and (oh * ow) % 4 == 0 and ow >= 4)
            f16 = (d.amp if f16 else 0)
            # fp32 Winograd F(2x2, 3x3) (csrc/conv_wino.hip: launch_conv_wino's rule): 16 instead of 36 multiply-adds per input
            # channel and 2x2 outputs -- class 3; its ALGORITHMIC flops (the direct count) stay what `achieved` is made of
            if (not d.amp and d.weight_wino and d.kh == 3 and d.kw == 3 and d.stride == 1 and d.pad == 1
                    and d.width % 2 == 0 and d.width >= 4 and d.c0 % 8 == 0 and cin % 8 == 0 and d.cout >= 32
                    and (d.height * d.width) % 4 == 0
                    and ((d.cout + 63) // 64) * ((d.batch * ((d.height + 1) // 2) * (d.width // 2) + 63) // 64) >= 160):
                f16 = 3
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._space(stream)
            s.record()
            rc = self.real(desc_ref, stream)
            e.record()
            self.records.append((flops, s, e, sig, nbytes, f16))
            return rc

        def timed_stem(in0, bs0, c0, in1, bs1, c1, batch, height, width, *rest):
            # deva_stem7x7 (--f16_split*: the 7x7 stride-2 stems on the f16 pipes): counted with the split kernels
            cin = c0 + c1
            oh, ow = height // 2, width // 2
            flops = 2.0 * 64 * cin * 49 * batch * oh * ow
            nbytes = 4.0 * (c0 * (batch if bs0 else 1) * height * width + c1 * batch * height * width + cin * 49 * 64
                            + 64 * oh * ow * batch)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._space(rest[-1])
            s.record()
            rc = self.real_stem(in0, bs0, c0, in1, bs1, c1, batch, height, width, *rest)
            e.record()
            self.records.append((flops, s, e, (cin, 64, 7, 2, batch, oh, ow), nbytes, 2))
            return rc

        self.handle.deva_conv2d = timed
        self.real_stem = self.handle.deva_stem7x7
        self.handle.deva_stem7x7 = timed_stem
        return self

    def __exit__(self, *a):
        self.handle.deva_conv2d = self.real
        self.handle.deva_stem7x7 = self.real_stem
        torch.cuda.synchronize()

    def summary(self):
        flops = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return flops, ms, len(self.records), sum(r[4] for r in self.records)

    def split_by_precision(self):
        """-> {'f16': (flops, ms, launches), 'split': (...), 'f32': (...)}: launches the fp16-operand kernels took, the
        hi/lo split kernels took (with their gated fp32 launch behind them), and the plain fp32 ones"""
        out = {'f16': [0.0, 0.0, 0], 'split': [0.0, 0.0, 0], 'f32': [0.0, 0.0, 0], 'wino': [0.0, 0.0, 0]}
        for r in self.records:
            o = out[{0: 'f32', 1: 'f16', 2: 'split', 3: 'wino'}[int(r[5])]]
            o[0] += r[0]
            o[1] += r[1].elapsed_time(r[2])
            o[2] += 1
        return out

    def per_layer(self, frames):
        """time and achieved TFLOP/s per distinct (cin, cout, k, stride, batch, OH, OW)"""
        agg = {}
        for fl, s, e, sig, _, _ in self.records:
            a = agg.setdefault(sig, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += fl
            a[2] += s.elapsed_time(e)
        rows = [dict(cin=k[0], cout=k[1], k=k[2], stride=k[3], batch=k[4], oh=k[5], ow=k[6],
                     calls_per_frame=v[0] / frames, ms_per_frame=v[2] / frames, gflop_per_call=v[1] / v[0] / 1e9,
                     tflops=v[1] / (v[2] * 1e-3) / 1e12) for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r['ms_per_frame'])


PEAK_F16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense f16 / bf16 MFMA


def conv_roofline_report(ct, frames):
    """convolution FLOPs / event time of a ConvTimer pass, by the kernels that took the launches: fp16-operand kernels
    (--amp) against the f16 MFMA peak; hi/lo split kernels (--f16_split) as fp32-EQUIVALENT TFLOP/s (the convolution's
    2*K*cout*pixels) and as f16 MFMA work -- three v_mfma_f32_32x32x16_f16 per operand block, so 3x the flops -- against
    the f16 peak (their event pairs include the gated fp32 launch that follows every split launch); fp32 kernels
    against the fp32 matrix peak"""
    sp = ct.split_by_precision()
    out = {'method': 'HIP events around every deva_conv2d launch of an un-synchronised replay (bench.py:ConvTimer)',
           # every operand of every deva_conv2d / deva_stem7x7 launch once
           'algorithmic_bytes_per_frame': sum(r[4] for r in ct.records) / frames}
    for name, (fl, ms, n) in sp.items():
        if n == 0:
            continue
        tf = fl / max(ms, 1e-9) / 1e9
        e = {'tflops': tf, 'gflop_per_frame': fl / frames / 1e9, 'ms_per_frame': ms / frames, 'launches_per_frame': n / frames}
        if name == 'f16':
            e['frac_of_f16_mfma_peak'] = tf / PEAK_F16_MATRIX_TFLOPS
        elif name == 'split':
            e['tflops'] = None
            e['fp32_equivalent_tflops'] = tf
            e['f16_mfma_tflops_issued (3 MFMAs per block)'] = 3 * tf
            e['frac_of_f16_mfma_peak'] = 3 * tf / PEAK_F16_MATRIX_TFLOPS
            e['vs_fp32_matrix_peak'] = tf / PEAK_FP32_MATRIX_TFLOPS
        elif name == 'wino':
            e['algorithmic_tflops'] = tf
            e['mfma_tflops_executed (16 of 36 multiply-adds)'] = tf / 2.25
            e['algorithmic_over_fp32_mfma_peak'] = tf / PEAK_FP32_MATRIX_TFLOPS
            e['executed_frac_of_fp32_mfma_peak'] = tf / 2.25 / PEAK_FP32_MATRIX_TFLOPS
        else:
            e['frac_of_fp32_mfma_peak'] = tf / PEAK_FP32_MATRIX_TFLOPS
        out[name + '_kernels'] = e
    return out


def affinity_microbench(device, n=10000, hw=8160, k=30, iters=20):
    """the whole memory read (similarity -> exact top-k -> softmax -> usage counters) at the BASELINE shape, event-timed:
    `deva_affinity_read` as the frame loop calls it (fp16 pre-filter + exact fp32 re-scoring, bit-identical to the
    fp32 kernels: tests/test_gpu_d_affinity.py) and, beside it, the fp32 kernels alone (pre-filter forced off)"""
    from deva.hip import lib
    from workload import synth
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=0)  # SURVEY.md §8d kernel-only inputs
    key = mk.t().contiguous().to(device)
    shr = ms.reshape(-1).contiguous().to(device)
    qk, qe = qk.to(device), qe.to(device)
    fix = torch.zeros(n, dtype=torch.int64, device=device)
    L = lib()
    scratch = torch.empty((L.deva_affinity_read_scratch(n, hw, k),), dtype=torch.int64, device=device)
    st = torch.cuda.current_stream().cuda_stream
    idx = torch.empty((hw, k), dtype=torch.int32, device=device)
    wgt = torch.empty((hw, k), dtype=torch.float32, device=device)

    def read():
        rc = L.deva_affinity_read(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(), hw, k,
                                  scratch.data_ptr(), idx.data_ptr(), wgt.data_ptr(), fix.data_ptr(), None, None, 0, st)
        assert rc == 0, L.deva_hip_last_error()

    us = {}
    try:
        for mode in (0, 1):
            L.deva_affinity_force_prefilter(mode)
            for _ in range(3):
                read()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                read()
            e1.record()
            torch.cuda.synchronize()
            us[mode] = e0.elapsed_time(e1) / iters * 1e3
        flag = L.deva_affinity_read_flag(scratch.data_ptr(), st)
        # the read as the frames BETWEEN two memory frames run it: bank operands of the pre-filter kept in a prepared-bank
        # buffer (deva_affinity_read_prepared, MemoryManager._prep_of), the three bank kernels skipped
        prep = torch.empty((L.deva_affinity_bank_prep_bytes(n) // 8 + 8,), dtype=torch.int64, device=device)

        def read_cached(valid):
            rc = L.deva_affinity_read_prepared(None, None, 0, key.data_ptr(), shr.data_ptr(), n, qk.data_ptr(), qe.data_ptr(), hw, k,
                                               scratch.data_ptr(), idx.data_ptr(), wgt.data_ptr(), fix.data_ptr(), None, None, 0,
                                               prep.data_ptr(), valid, st)
            assert rc == 0, L.deva_hip_last_error()

        read_cached(0)
        for _ in range(3):
            read_cached(1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            read_cached(1)
        e1.record()
        torch.cuda.synchronize()
        us_cached = e0.elapsed_time(e1) / iters * 1e3
    finally:
        L.deva_affinity_force_prefilter(1)
    t = us[1] * 1e-6
    flops = 4.0 * 64 * n * hw                       # the reference's two K=64 contractions (memory_utils.py:29-43)
    f16_flops = 2 * 2.0 * 144 * n * hw              # what the pre-filter issues: two passes of K = 144 (P, m x bsq, Q chains)
    b_alg = 4.0 * (64 * n + n + 2 * 64 * hw) + 8.0 * k * hw + 4.0 * n
    b_mat = 12.0 * 4 * n * hw
    return dict(shape=dict(n=n, hw=hw, k=k), us_read=us[1], us_read_bank_operands_cached=us_cached,
                us_read_fp32_kernels_only=us[0], speedup_over_fp32_kernels=us[0] / us[1],
                prefilter_fell_back=bool(flag),
                bound='f16 MFMA operand delivery + VALU scoring (the fp32 matrix rate no longer binds: only ~35 of the '
                      f'{n} tokens per query are scored in fp32)',
                # the roofline fractions of what the kernels actually execute come first ...
                f16_mfma_tflops=f16_flops / t / 1e12, f16_mfma_frac=f16_flops / t / 1e12 / PEAK_F16_MATRIX_TFLOPS,
                hbm_algorithmic_gbps=b_alg / t / 1e9, hbm_algorithmic_frac=b_alg / t / 1e9 / PEAK_HBM_GBPS,
                hbm_counter_traffic_over_algorithmic=_affinity_counter_ratio(b_alg),
                fp32_kernels_frac_of_fp32_matrix_roof=flops / (us[0] * 1e-6) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                # ... then the NOMINAL figure (the reference's 4*64*N*HW FLOPs, which the pre-filter no longer performs,
                # over the read time): a speed-up expressed in TFLOP/s, NOT a roofline fraction
                reference_flops_over_read_time_tflops_nominal=flops / t / 1e12,
                reference_flops_over_read_time_vs_fp32_matrix_peak_nominal=flops / t / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                peak_fp32_matrix_tflops=PEAK_FP32_MATRIX_TFLOPS,
                hbm_materialised_equiv_gbps=b_mat / t / 1e9,
                parity_gate='tests/test_gpu_g_fullsize.py::test_affinity_at_bench_shapes + '
                            'tests/test_gpu_d_affinity.py::test_fp16_prefilter_is_bit_identical_to_the_fp32_kernels')


def _affinity_counter_ratio(b_alg):
    """HBM counter traffic of one read (committed PMC pass of this round, or the last one) over the algorithmic bytes"""
    for d in ('pmc_r06', 'pmc_r05', 'pmc_r04', 'pmc_r03'):
        path = os.path.join(ROOT, 'profiles', d, 'affinity_read.json')
        if os.path.exists(path):
            try:
                with open(path) as f:
                    j = json.load(f)
                # tools/pmc_summary.py read: per-kernel per-dispatch averages + the sum over the kernels of one read
                tot = j.get('read_total', j.get('prefilter_total', {})).get('hbm_bytes')
                if tot:
                    return {'ratio': tot / b_alg, 'hbm_bytes_per_read': tot, 'source': f'profiles/{d}/affinity_read.json',
                            'counters': 'FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, summed over the kernels of one read'}
            except Exception:  # noqa: BLE001
                pass
    return None


def pointwise_rooflines(device, tag, no, h16, w16, iters=20):
    """HBM-bound kernels of one frame, each event-timed on its own with the tensor shapes of the
    frame: algorithmic bytes (every operand once) / time / 8 TB/s.  h16 x w16 = the 1/16 map, no = objects."""
    from deva.hip import ops
    g = torch.Generator(device='cpu').manual_seed(5)

    def t(*shape):
        x = ops._alloc(shape, device)
        x.copy_(torch.randn(*shape, generator=g))
        return x

    rows = []

    def bench(name, fn, nbytes):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        rows.append({'kernel': name, 'shape': tag, 'us': us, 'algorithmic_bytes': nbytes, 'gbps': nbytes / us / 1e3,
                     'frac_of_hbm_peak': nbytes / us / 1e3 / PEAK_HBM_GBPS})

    f = 4.0
    p16, d8 = t(no, 512, h16, w16), t(1, 512, 2 * h16, 2 * w16)
    bench('upsample2x_add_kernel (p16 -> 1/8, 512 ch)', lambda: ops.upsample2x_add(p16, d8),
          f * (p16.numel() + d8.numel() + no * 512 * 4 * h16 * w16))
    p8, d4 = t(no, 256, 2 * h16, 2 * w16), t(1, 256, 4 * h16, 4 * w16)
    bench('upsample2x_add_kernel (p8 -> 1/4, 256 ch)', lambda: ops.upsample2x_add(p8, d4),
          f * (p8.numel() + d4.numel() + no * 256 * 16 * h16 * w16))
    p4 = t(no, 256, 4 * h16, 4 * w16)
    bench('area_downsample_kernel (p4, factor 4)', lambda: ops.area_downsample(p4, 4), f * (p4.numel() + p4.numel() / 16))
    bench('area_downsample_kernel (p8, factor 2)', lambda: ops.area_downsample(p8, 2), f * (p8.numel() + p8.numel() / 4))
    pc = ops.pack_conv(torch.randn(1, 256, 3, 3, generator=g) * 0.02, torch.zeros(1), device=device)
    bench('conv3x3_cout1_rows_kernel (mask-logit head 256 -> 1)', lambda: ops.conv2d(pc, p4, pad=1, relu_in=True),
          f * (p4.numel() + no * 16 * h16 * w16 + 9 * 256))
    stem = t(1, 64, 8 * h16, 8 * w16)
    bench('maxpool3x3s2_kernel (key-encoder stem, 64 ch)', lambda: ops.maxpool3x3s2(stem), f * (stem.numel() * 1.25))
    logits = t(no + 1, 4 * h16, 4 * w16)
    bench('upsample4x_softmax_kernel (logits -> full resolution)', lambda: ops.upsample4x_softmax(logits, need_logits=False),
          f * (logits.numel() * 17))
    return rows


def readout_roofline(device, tag, no, hw, n, k=30, iters=20):
    """readout_sparse_kernel at one shape: SURVEY 8d's B_ro = 4*no*CV*(min(N, k*HW) + HW) + 8*k*HW"""
    from deva.hip import ops
    g = torch.Generator(device='cpu').manual_seed(6)
    cv = 512  # one launch per object (memory_manager.py:_readout_into): the row is one object's read-out
    vals = torch.randn(n, cv, generator=g).to(device)
    idx = torch.randint(0, n, (hw, k), generator=g, dtype=torch.int32).to(device)
    w = torch.rand(hw, k, generator=g).to(device)
    out = torch.empty((cv, hw), dtype=torch.float32, device=device)
    no = 1

    def fn():
        ops.readout_sparse(idx, w, None, 0, vals, out)

    try:
        for _ in range(3):
            fn()
    except Exception as exc:  # noqa: BLE001  (signature drift must not cost the line)
        return [{'kernel': 'readout_sparse_kernel', 'shape': tag, 'error': f'{type(exc).__name__}: {exc}'[:200]}]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    nbytes = 4.0 * no * cv * (min(n, k * hw) + hw) + 8.0 * k * hw
    return [{'kernel': 'readout_sparse_kernel (one object)', 'shape': tag, 'us': us, 'algorithmic_bytes': nbytes, 'gbps': nbytes / us / 1e3,
             'frac_of_hbm_peak': nbytes / us / 1e3 / PEAK_HBM_GBPS}]


def cpu_affinity_kernels(budget_s=40.0):
    """the reference's get_similarity + do_softmax(top_k=30, return_usage) on the host cores at the five
    kernel-only shapes of SURVEY 8d (oracle port, bit-identical arithmetic): one warm-up call, then the median of three
    timed calls per shape (fewer once the time budget of the bounded CPU sample is spent: `timed_calls_after_one_warm_up`)"""
    from oracle import deva_oracle as O
    from workload import synth
    rows, t_start = [], time.perf_counter()
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:  # noqa: BLE001
        avail = 16 << 30
    for n, hw in ((10000, 1620), (24580, 1620), (10000, 8160), (83440, 8160), (50000, 32400)):
        if time.perf_counter() - t_start > budget_s:
            rows.append({'n': n, 'hw': hw, 'skipped': 'time budget of the bounded CPU sample'})
            continue
        if 6 * 4 * n * hw > avail // 4:  # the materialised N x HW passes need ~6 such matrices; never risk the box
            rows.append({'n': n, 'hw': hw, 'skipped': f'needs ~{6 * 4 * n * hw / 2**30:.0f} GiB of host memory for the reference\'s '
                                                      'materialised N x HW matrices'})
            continue
        mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=0)
        sims, softs = [], []
        for rep in range(4):  # one warm-up call (first-touch of the N x HW buffers, thread pool), then up to three timed
            t0 = time.perf_counter()
            sim = O.get_similarity(mk, ms, qk, qe)
            t1 = time.perf_counter()
            O.dense_affinity(sim, 30)  # top-30 -> exp / normalise -> scatter into the dense matrix -> usage (row sums)
            t2 = time.perf_counter()
            del sim
            if rep > 0:
                sims.append((t1 - t0) * 1e3)
                softs.append((t2 - t1) * 1e3)
            if rep > 0 and time.perf_counter() - t_start > budget_s:
                break
        med = lambda v: sorted(v)[(len(v) - 1) // 2]
        rows.append({'n': n, 'hw': hw, 'get_similarity_ms': med(sims), 'do_softmax_top30_ms': med(softs),
                     'timed_calls_after_one_warm_up': len(sims), 'get_similarity_ms_range': [min(sims), max(sims)],
                     'do_softmax_top30_ms_range': [min(softs), max(softs)], 'cores': torch.get_num_threads()})
    return rows


def cpu_baseline(sd, cfg, height, width, num_objects, frames_cpu, runs=3, warm=2):
    """the CPU oracle on the first frames of the same clip: `runs` runs of the SAME frames, each from
    a fresh core: annotated frame, `warm` untimed warm-up frames, then the timed frames (memory frames every
    mem_every-th included); `value` = median FPS of the runs, their range beside it; per-stage milliseconds per
    propagated frame (median run) from timers wrapped around the oracle's own functions.  BASELINE configs[1] runs with
    the long-term memory disabled, so there is no consolidation to include."""
    from oracle import deva_oracle as O
    from workload import synth
    stages = {}

    def timed(name, fn):
        def wrapper(*a, **kw):
            t0 = time.perf_counter()
            out = fn(*a, **kw)
            stages[name] = stages.get(name, 0.0) + time.perf_counter() - t0
            return out
        return wrapper

    names = ('encode_image', 'transform_key', 'segment', 'encode_mask')
    saved = {n: getattr(O, n) for n in names}
    mem_saved = (O.OracleMemory.match, O.OracleMemory.add)
    for n in names:
        setattr(O, n, timed(n, saved[n]))
    O.OracleMemory.match = timed('match_memory', mem_saved[0])
    O.OracleMemory.add = timed('add_memory', mem_saved[1])
    n = len(frames_cpu) - 1 - warm
    assert n >= 1, 'cpu_baseline needs more frames than the warm-up'
    results = []
    try:
        mask = synth.box_mask(height, width, num_objects)
        for _ in range(runs):
            core = O.OracleCore(sd, cfg)
            core.step(frames_cpu[0], mask, list(range(1, num_objects + 1)))
            for f in frames_cpu[1:1 + warm]:
                core.step(f)
            stages.clear()
            t0 = time.perf_counter()
            for f in frames_cpu[1 + warm:]:
                core.step(f)
            results.append((n / (time.perf_counter() - t0), dict(stages)))
            del core
    finally:
        for k_, v in saved.items():
            setattr(O, k_, v)
        O.OracleMemory.match, O.OracleMemory.add = mem_saved
    results.sort(key=lambda r: r[0])
    fps, st = results[(len(results) - 1) // 2]
    # "reference" would be the reference's own modules timed here; /root/reference does not exist on the GPU box, so
    # the baseline is the oracle (bit-identical to the reference in the build container, tests/test_oracle_golden.py)
    return dict(value=fps, unit='frames/s', cores=torch.get_num_threads(), kind='port',
                runs_fps=[r[0] for r in results], range_fps=[results[0][0], results[-1][0]],
                spread_frac=(results[-1][0] - results[0][0]) / fps,
                stage_ms_per_frame={k_: 1e3 * v / n for k_, v in st.items()},
                sample=f'median of {runs} runs; each: annotated frame + {warm} warm-up frames (untimed) + {n} timed propagated '
                       f'frames (memory frames every {cfg["mem_every"]}th), same workload and weights (CPU oracle, fp32)')


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


def run_long4k(net, device, steps, warmup, seed, shard, dist, size=(2160, 3840), bank_tokens=50000):
    """BASELINE configs[4]: one 4K clip, 1 object, 50 000-token long-term bank.  shard = None (one GPU) |
    'owner' | 'queries' | 'bank' (MemoryManager modes, all ranks of the group step the same clip).
    -> (FPS, bank sizes at the end, collective bytes this rank moved per timed frame)"""
    from workload import synth
    cfg = synth.base_config(max_long_term_elements=bank_tokens)
    n_frames = 1 + warmup + steps
    frames = make_clip(size[0], size[1], n_frames, seed=seed, device=device)
    core = start_clip(net, cfg, frames, 1, device, lt_prefill=bank_tokens - cfg['num_prototypes'], shard=shard)
    for t in range(1, 1 + warmup):
        core.step(frames[t])
    comm0 = core.memory.comm_bytes

    def timed_steps():
        for t in range(1 + warmup, n_frames):
            core.step(frames[t])

    elapsed = timed_region(timed_steps, dist, device)
    mem = core.memory
    bank = {'long': {b: mem.long_mem.size(b) for b in mem.long_mem.buckets},
            'work': {b: mem.work_mem.size(b) for b in mem.work_mem.buckets}}
    return steps / elapsed, bank, (mem.comm_bytes - comm0) / steps


def run_prefetched(net, device, cfg, height, width, num_objects, steps, warmup, seed):
    """the headline loop with ONE addition: before frame t is stepped, the key encoder of frame t+1 is started on a
    side stream (`ImageFeatureStore.prefetch`, an extension of this package -- the reference's drivers do not call
    it).  Same kernels, same results; the encoder's latency-bound batch-1 launches overlap the decoder of frame t."""
    from deva.utils.tensor_utils import pad_divide_by
    n_frames = 1 + warmup + steps
    frames = make_clip(height, width, n_frames + 1, seed=seed, device=device)
    padded = [pad_divide_by(f, 16)[0].unsqueeze(0) for f in frames]
    core = start_clip(net, cfg, frames, num_objects, device)

    def step(t):
        core.image_feature_store.prefetch(core.curr_ti + 2, padded[t + 1])
        core.step(frames[t])

    for t in range(1, 1 + warmup):
        step(t)
    elapsed = timed_region(lambda: [step(t) for t in range(1 + warmup, n_frames)], None, device)
    core.image_feature_store.delete(core.curr_ti + 1)
    return steps / elapsed


def run_480p_single(net, device, steps, warmup, seed=100):
    """BASELINE configs[0] shape on the GPU: 854x480, ONE object, default flags (long-term memory on) -- the regime
    where the frame is short enough for launch gaps to show"""
    from workload import synth
    cfg = synth.base_config()
    n_frames = 1 + warmup + steps
    frames = make_clip(480, 854, n_frames, seed=seed, device=device)
    core = start_clip(net, cfg, frames, 1, device)
    for t in range(1, 1 + warmup):
        core.step(frames[t])
    elapsed = timed_region(lambda: [core.step(frames[t]) for t in range(1 + warmup, n_frames)], None, device)
    return steps / elapsed


def run_480p_headline(net, device, cfg, args, seed=100, conv_roofline=True):
    """the headline workload (BASELINE configs[1]) on another network build (--f16_split): same clip, same loop"""
    n_frames = 1 + args.warmup + args.steps
    frames = make_clip(args.height, args.width, n_frames, seed=seed, device=device)
    core = start_clip(net, cfg, frames, args.objects, device)
    for t in range(1, 1 + args.warmup):
        core.step(frames[t])
    elapsed = timed_region(lambda: [core.step(frames[t]) for t in range(1 + args.warmup, n_frames)], None, device)
    state = {'work': {b: core.memory.work_mem.size(b) for b in core.memory.work_mem.buckets}}
    if conv_roofline:
        more = make_clip(args.height, args.width, 2 + min(20, args.steps), seed=999, device=device)
        for f in more[:2]:
            core.step(f)
        torch.cuda.synchronize()
        with ConvTimer(spacer=True) as ct:
            for f in more[2:]:
                core.step(f)
        state['conv_roofline'] = conv_roofline_report(ct, len(more) - 2)
        from deva.hip import ops as _ops
        state['split_fallbacks'] = _ops.split_fallbacks(device)
    return args.steps / elapsed, state


def run_1080p(net, device, steps, warmup, detections, seed=7, conv_roofline=False):
    """The 1080p lines (1920x1080, padded 1088x1920, ONE object, long-term memory pre-filled to 10 000 tokens):
    detections=False -- pure propagation (the north-star target line: >= 30 FPS with a 10k-element bank);
    detections=True  -- BASELINE configs[2]: a precomputed detection of the object is merged every 5th frame
                        through incorporate_detection (eval_with_detections' online setting) with
                        --max_num_objects 1 --max_missed_detection_count 1000000, i.e. the tracker keeps exactly
                        that one object (with recipe weights the propagated mask does not reach IoU 0.5 with the
                        detection, and unbounded flags would add a new object per detection)."""
    from workload import synth
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    H, W, every = 1080, 1920, 5
    cfg = synth.base_config(max_missed_detection_count=10**6, max_num_objects=1)
    n_frames = 1 + warmup + steps
    frames = make_clip(H, W, n_frames, seed=seed, device=device)
    dets = {t: synth.detection_frame(H, W, t, segments=1) for t in range(0, n_frames, every)}
    dets = {t: (m.to(device), info) for t, (m, info) in dets.items()}
    core = DEVAInferenceCore(net, cfg)

    def run(t):
        if t in dets and (detections or t == 0):
            m, info = dets[t]
            core.incorporate_detection(frames[t], m, [ObjectInfo(**i) for i in info])
        else:
            core.step(frames[t])

    run(0)
    key, shr, vals = synth.prefill_bank(10000, core.object_manager.all_obj_ids, seed=1)
    core.memory.long_mem.add(key.to(device), {o: v.to(device) for o, v in vals.items()}, shr.to(device),
                             selection=None, supposed_bucket_id=0)
    for t in range(1, 1 + warmup):
        run(t)
    elapsed = timed_region(lambda: [run(t) for t in range(1 + warmup, n_frames)], None, device)
    mem = core.memory
    state = {'long': {b: mem.long_mem.size(b) for b in mem.long_mem.buckets},
             'work': {b: mem.work_mem.size(b) for b in mem.work_mem.buckets},
             'objects': core.object_manager.num_obj}
    if conv_roofline:  # event-timed continuation of the clip
        more = make_clip(H, W, 8, seed=seed + 1, device=device)
        for f in more[:2]:
            core.step(f)
        torch.cuda.synchronize()
        with ConvTimer(spacer=True) as ct:
            for f in more[2:]:
                core.step(f)
        state['conv_roofline'] = conv_roofline_report(ct, len(more) - 2)
        from deva.hip import ops as _ops
        state['split_fallbacks'] = _ops.split_fallbacks(device)
    if os.environ.get('DEVA_BENCH_LAYERS_1080') and not detections:  # per-layer table of this line (tuning aid)
        more = make_clip(H, W, 6, seed=seed + 1, device=device)
        with ConvTimer() as ct:
            for f in more:
                core.step(f)
        with open(os.environ['DEVA_BENCH_LAYERS_1080'], 'w') as f:
            json.dump(ct.per_layer(len(more)), f, indent=1)
    return steps / elapsed, state


def run_1080p_segments(net, device, steps, warmup, segments=8, seed=7, size=(1080, 1920), conv_roofline=False):
    """BASELINE configs[2] as SURVEY.md 8d defines it: 1920x1080, a precomputed detection with `segments` segments
    merged every 5th frame through incorporate_detection (online setting of evaluation/eval_with_detections.py:
    280-297, --max_missed_detection_count 1, no object cap), long-term memory pre-filled to 10 000 tokens.
    The detections are tracker-consistent (workload/detections.py): most segments re-detect tracked objects
    (IoU 0.875 -> matched and merged), two per detection are new (-> new objects in a new memory bucket) and
    objects that go unseen twice are purged with their memories.  They are a function of the tracker's own
    forward masks, so the clip is DEFINED by an untimed recording pass (hook around match_and_merge) and then
    replayed through the public interface in the timed pass; the kernels are deterministic, so the replay
    reproduces the recording pass (asserted on the final object table)."""
    from workload import detections, synth
    from deva.inference.inference_core import DEVAInferenceCore
    from deva.inference.object_info import ObjectInfo
    (H, W), every = size, 5
    cfg = synth.base_config(max_missed_detection_count=1, max_num_objects=-1)
    n_frames = 1 + warmup + steps
    frames = make_clip(H, W, n_frames, seed=seed, device=device)
    empty = torch.zeros(H, W, dtype=torch.long, device=device)

    def prefill(core):
        key, shr, vals = synth.prefill_bank(10000, core.object_manager.all_obj_ids, seed=1)
        core.memory.long_mem.add(key.to(device), {o: v.to(device) for o, v in vals.items()}, shr.to(device),
                                 selection=None, supposed_bucket_id=0)

    def table(core):
        return [(int(o.id), int(o.poke_count)) for o in core.object_manager.obj_to_tmp_id]

    # pass 1 (untimed): the clip is generated
    rec_core = DEVAInferenceCore(net, cfg)
    detector = detections.ConsistentDetector(H, W, segments=segments, new_per_frame=2)
    recorded, now = {}, [0]
    for t in range(n_frames):
        now[0] = t
        if t % every == 0:
            with detections.record_on_package(rec_core, detector, ObjectInfo, recorded, lambda: now[0]):
                rec_core.incorporate_detection(frames[t], empty, [])
        else:
            rec_core.step(frames[t])
        if t == 0:
            prefill(rec_core)
    want = table(rec_core)
    del rec_core
    dets = {t: (m.to(device), info) for t, (m, info) in recorded.items()}

    # pass 2 (timed): replay through the public interface
    core = DEVAInferenceCore(net, cfg)
    live = []

    def run(t):
        if t in dets:
            m, info = dets[t]
            core.incorporate_detection(frames[t], m, [ObjectInfo(**i) for i in info])
        else:
            core.step(frames[t])
        live.append(core.object_manager.num_obj)

    run(0)
    prefill(core)
    for t in range(1, 1 + warmup):
        run(t)
    elapsed = timed_region(lambda: [run(t) for t in range(1 + warmup, n_frames)], None, device)
    assert table(core) == want, ('the timed replay did not reproduce the recording pass', table(core), want)
    mem = core.memory
    timed_live = live[1 + warmup:]
    state = {'long': {b: mem.long_mem.size(b) for b in mem.long_mem.buckets},
             'work': {b: mem.work_mem.size(b) for b in mem.work_mem.buckets},
             'objects': core.object_manager.num_obj,
             'objects_per_timed_frame': {'min': min(timed_live), 'mean': sum(timed_live) / len(timed_live),
                                         'max': max(timed_live)},
             'detections_in_timed_region': sum(1 for t in dets if t > warmup),
             'segments_per_detection': segments,
             'matched_segments': sum(1 for t, (_, info) in dets.items() if t > warmup for i in info if i['id'] > 100000),
             'object_table_at_end(id, missed detections)': table(core)}
    if conv_roofline:
        # event-timed replay of the same clip (third pass): convolution FLOPs / time, split by the kernels that took them
        core2 = DEVAInferenceCore(net, cfg)
        live.clear()
        core, core2 = core2, core
        run(0)
        prefill(core)
        for t in range(1, 1 + warmup):
            run(t)
        torch.cuda.synchronize()
        with ConvTimer(spacer=True) as ct:
            for t in range(1 + warmup, n_frames):
                run(t)
        state['conv_roofline'] = conv_roofline_report(ct, steps)
        from deva.hip import ops as _ops
        state['split_fallbacks'] = _ops.split_fallbacks(device)
    return steps / elapsed, state


def long4k(args, net, rank, world, device, dist):
    """`--workload long4k`: strong scaling of ONE clip over the GPUs of the node"""
    mode = args.long4k_mode if dist is not None else None
    tiny = dict(size=(96, 128), bank_tokens=600) if os.environ.get('DEVA_BENCH_EMULATED') == '1' else {}  # (CPU test)
    fps, bank, comm = run_long4k(net, device, args.steps, args.warmup, seed=11, shard=mode, dist=dist, **tiny)
    what = {None: 'one GPU', 'owner': 'frame owner (rank 0 encodes / decodes) + query-sharded read',
            'queries': 'every rank steps the clip, query-sharded read',
            'bank': 'every rank steps the clip, token-sharded read (candidate keys all-gathered, exact merge)',
            'owner_bank': 'frame owner (rank 0 encodes / decodes) + token-sharded read over a value-sharded bank (candidate keys '
                          'all-gathered, partial read-outs reduced to the owner)'}[mode]
    if rank == 0:
        print(json.dumps({
            'metric': 'propagation FPS @4K (1 object, 50k-token long-term bank)',
            'value': fps, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 / fps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[4]: one synthetic 3840x2160 clip, 1 object, long-term memory '
                                   'pre-filled to 50 000 tokens',
                       'bank_tokens_at_end': bank, 'parallelism': f'{what} x{world}',
                       'collective_bytes_per_frame_rank0': comm},
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def extra_lines(net, device, cfg, args):
    """the further lines of the metric (`also`): each with its own warm-up and timed frames, each naming its parity gate"""
    def line(metric, run, steps, warmup, workload, gate, state_key=None, **more):
        """one further line of the metric; a failure is reported in place and never costs the headline line"""
        entry = {'metric': metric, 'unit': 'frames/s', 'steps': steps, 'warmup': warmup,
                 'config': {'workload': workload}, 'parity_gate': gate, **more}
        try:
            out = run()
            fps = out[0] if isinstance(out, tuple) else out
            entry.update(value=fps, ms_per_step=1e3 / fps)
            if state_key is not None:
                entry['config'][state_key] = out[1]
        except Exception as exc:  # noqa: BLE001
            entry.update(value=None, error=f'{type(exc).__name__}: {exc}'[:400])
            torch.cuda.synchronize()
        return entry

    gate1080 = 'tests/test_gpu_g_fullsize.py::test_1080p_detections_10k_bank_against_oracle'
    SPLIT_DTYPE = 'f32 via 3x f16 split, f32 acc (value encoder, mask decoder); f32 elsewhere'
    SPLIT_GATE = ('tests/test_gpu_a_conv.py::test_conv_split_matches_cpu (the fp32 cases at the fp32 bound 2e-5) + '
                  'test_conv_split_is_fp32_accurate (against fp64) + tests/test_gpu_e_network.py::test_480p_lockstep_teacher_forced / '
                  'test_1080p_lockstep_teacher_forced [f16_split] (fp32 bounds, one oracle pass with the fp32 build) + '
                  'test_f16_split_e2e_against_reference_golden[f16_split-five_obj]; free-running at bench size under the superset '
                  'mode: the gates of the key_encoder lines')
    SPLIT_ALL_GATE = ('the conv tests of --f16_split + tests/test_gpu_e_network.py::test_480p_lockstep_teacher_forced / '
                      'test_1080p_lockstep_teacher_forced [f16_split+key_encoder] + test_f16_split_e2e_against_reference_golden'
                      '[f16_split+key_encoder-*] + test_f16_split_480p_five_objects_against_oracle + tests/test_gpu_g_fullsize.py::'
                      'test_1080p_eight_segment_detections_against_oracle[f16_split+key_encoder] + test_4k_lockstep '
                      '[f16_split+key_encoder] -- all in the default -m gpu suite, fp32 bounds')
    SPLIT_ALL_DTYPE = 'f32 via 3x f16 split, f32 acc (key encoder, value encoder, mask decoder); f32 elsewhere'
    _split, _split_all = [], []

    def split_net():
        if not _split:
            _split.append(build_network(device, split=True)[0])
        return _split[0]

    def split_all_net():
        if not _split_all:
            _split_all.append(build_network(device, split=True, split_key_encoder=True)[0])
        return _split_all[0]
    return [
        line('propagation FPS @480p (5 objects, working memory only) WITH next-frame key-encoder prefetch',
             lambda: run_prefetched(net, device, cfg, args.height, args.width, args.objects, args.steps, args.warmup,
                                    seed=100),
             args.steps, args.warmup,
             'the headline clip and loop, plus ImageFeatureStore.prefetch(t+1) on a side stream before every step (an '
             'extension: the unchanged drivers do not call it; identical results)',
             'tests/test_gpu_e_network.py::test_prefetched_key_encoder_is_bit_identical'),
        line('propagation FPS @480p (1 object, default flags)',
             lambda: run_480p_single(net, device, steps=60, warmup=10), 60, 10,
             'BASELINE configs[0] shape on the GPU: synthetic 854x480 clip, one object, long-term memory on (the '
             'launch-gap-sensitive regime)',
             'tests/test_gpu_e_network.py::test_vos_example_against_reference_golden'),
        line('propagation FPS @1080p (1 object, 10k-token long-term bank)',
             lambda: run_1080p(net, device, steps=25, warmup=6, detections=False), 25, 6,
             'north-star target line: synthetic 1920x1080 clip (padded 1088x1920), one object initialised from a '
             'detection, long-term memory pre-filled to 10 000 tokens + working memory, pure propagation',
             gate1080, 'state_at_end', target_fps=30.0),
        line('propagation FPS @1080p (detections merged every 5th frame, 1 object, 10k-token long-term bank)',
             lambda: run_1080p(net, device, steps=25, warmup=6, detections=True), 25, 6,
             'the same clip with a fixed-box detection handed to incorporate_detection every 5th frame under '
             '--max_num_objects 1: the detection is DISCARDED (segment_merging.py:115-122), so this line times the '
             'forward pass + argmax + histogram of a detection frame, not a merge -- the 8-segment line below is '
             'BASELINE configs[2]',
             gate1080, 'state_at_end', target_fps=30.0),
        line('propagation FPS @1080p (8-segment detections merged every 5th frame, ~10 live objects, 10k-token '
             'long-term bank)',
             lambda: run_1080p_segments(net, device, steps=25, warmup=6, segments=8), 25, 6,
             'BASELINE configs[2] as SURVEY.md 8d defines it: tracker-consistent detections with 8 segments '
             '(re-detections that match and merge, 2 new objects per detection in a new bucket, objects unseen twice '
             'purged), online setting, no object cap',
             'tests/test_gpu_g_fullsize.py::test_1080p_eight_segment_detections_against_oracle[fp32] (8 segments, 14 live objects '
             'at 1080p, the same generator) + tests/test_gpu_e_network.py::test_consistent_detection_clip_against_reference_golden',
             'state_at_end'),
        line('propagation FPS @1080p, --amp (8-segment detections merged every 5th frame, ~10 live objects, 10k-token '
             'long-term bank)',
             lambda: run_1080p_segments(build_network(device, amp=True)[0], device, steps=25, warmup=6, segments=8,
                                        conv_roofline=True), 25, 6,
             'the 8-segment clip above with --amp: fp16 operands / fp32 accumulation (v_mfma_f32_32x32x16_f16) in the '
             'value encoder and the mask decoder, key encoder / key projection / memory read / aggregate / mask-logit head '
             'fp32 (NOT the parity target: the headline and every other line are fp32)',
             'tests/test_gpu_a_conv.py::test_conv_amp_matches_fp16_rounded_cpu (single convolutions, 2e-5) + '
             'tests/test_gpu_e_network.py::test_amp_lockstep_teacher_forced (stages, quantisation-noise bounds)',
             'state_at_end', dtype='f16-in/f32-acc (value encoder, mask decoder); f32 elsewhere', target_fps=25.0),
        line('propagation FPS @480p, --f16_split (5 objects, working memory only)',
             lambda: run_480p_headline(split_net(), device, cfg, args), args.steps, args.warmup,
             'the headline clip and loop with --f16_split: fp32-ACCURATE convolutions on the f16 matrix pipes (hi/lo fp16 '
             'split of both operands, three v_mfma_f32_32x32x16_f16 per block, fp32 accumulation) in the value encoder and '
             'the mask decoder; key encoder / key projection / memory read on the fp32 kernels',
             SPLIT_GATE, 'state_at_end', dtype=SPLIT_DTYPE),
        line('propagation FPS @1080p, --f16_split (1 object, 10k-token long-term bank)',
             lambda: run_1080p(split_net(), device, steps=25, warmup=6, detections=False, conv_roofline=True), 25, 6,
             'the north-star target line with --f16_split', SPLIT_GATE, 'state_at_end', dtype=SPLIT_DTYPE, target_fps=30.0),
        line('propagation FPS @1080p, --f16_split (8-segment detections merged every 5th frame, ~10 live objects, 10k-token '
             'long-term bank)',
             lambda: run_1080p_segments(split_net(), device, steps=25, warmup=6, segments=8, conv_roofline=True), 25, 6,
             'BASELINE configs[2] (the 8-segment clip above) with --f16_split', SPLIT_GATE, 'state_at_end',
             dtype=SPLIT_DTYPE, target_fps=30.0),
        line('propagation FPS @480p, --f16_split --f16_split_key_encoder (5 objects, working memory only)',
             lambda: run_480p_headline(split_all_net(), device, cfg, args), args.steps, args.warmup,
             'the headline clip and loop with the key encoder on the split kernels too (second level of the opt-in)',
             SPLIT_ALL_GATE, 'state_at_end', dtype=SPLIT_ALL_DTYPE),
        line('propagation FPS @1080p, --f16_split --f16_split_key_encoder (1 object, 10k-token long-term bank)',
             lambda: run_1080p(split_all_net(), device, steps=25, warmup=6, detections=False, conv_roofline=True), 25, 6,
             'the north-star target line with the key encoder on the split kernels too', SPLIT_ALL_GATE, 'state_at_end',
             dtype=SPLIT_ALL_DTYPE, target_fps=30.0),
        line('propagation FPS @1080p, --f16_split --f16_split_key_encoder (8-segment detections merged every 5th frame, ~10 live '
             'objects, 10k-token long-term bank)',
             lambda: run_1080p_segments(split_all_net(), device, steps=25, warmup=6, segments=8, conv_roofline=True), 25, 6,
             'BASELINE configs[2] (the 8-segment clip above) with the key encoder on the split kernels too (second level of the '
             'opt-in: the memory read\'s inputs then move by fp32 round-off)', SPLIT_ALL_GATE, 'state_at_end',
             dtype=SPLIT_ALL_DTYPE, target_fps=30.0),
        line('propagation FPS @4K (1 object, 50k-token long-term bank), one GPU',
             lambda: run_long4k(net, device, steps=20, warmup=5, seed=11, shard=None, dist=None)[:2], 20, 5,
             'BASELINE configs[4] on ONE GPU: synthetic 3840x2160 clip, 1 object, long-term memory pre-filled to '
             '50 000 tokens',
             'tests/test_gpu_g_fullsize.py::test_4k_free_running_50k_bank_against_oracle + test_4k_lockstep + '
             'test_affinity_at_bench_shapes (every query)', 'bank_tokens_at_end'),
        line('propagation FPS @4K, --f16_split --f16_split_key_encoder (1 object, 50k-token long-term bank), one GPU',
             lambda: run_long4k(split_all_net(), device, steps=20, warmup=5, seed=11, shard=None, dist=None)[:2], 20, 5,
             'BASELINE configs[4] on ONE GPU with the fp32-accurate split kernels in every scope',
             'tests/test_gpu_g_fullsize.py::test_4k_lockstep [f16_split+key_encoder] (teacher-forced, fp32 bounds; the free-running '
             '4K gate runs the fp32 build) + the gates of the other key_encoder lines', 'bank_tokens_at_end', dtype=SPLIT_ALL_DTYPE),
    ]


LINE_LIMIT = 4096  # bytes of the final stdout line (a 27 KB line was not read back by the driver)
CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'rccl_ranks')
ROOFLINE_KEYS = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'mfma_executed_frac', 'winograd_share_of_algorithmic_flops',
                 'traffic', 'traffic_over_algorithmic',
                 'gflop_per_frame', 'ms_in_kernel_per_frame', 'launches_per_frame', 'algorithmic_bytes_per_frame',
                 'frac_of_sustained_probe')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'range_fps', 'stage_ms_per_frame', 'sample')


def _round(x, digits=4):
    """floats to `digits` significant figures (the full-precision numbers are in bench_extra.json)"""
    if isinstance(x, float):
        return float(f'{x:.{digits}g}')
    if isinstance(x, dict):
        return {k: _round(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, digits) for v in x]
    return x


def compact_line(result):
    """the ONE line of the contract: exactly CONTRACT_KEYS, scalars only below the second level, <= LINE_LIMIT bytes.
    Everything else (`also`, `also_kernels`, `affinity`, `timed_like_eval_vos`, per-shape CPU rows, method strings,
    parity gates) is in bench_extra.json."""
    line = {k: result[k] for k in CONTRACT_KEYS if k in result}
    for k in ('value', 'ms_per_step'):
        if isinstance(line.get(k), float):
            line[k] = float(f'{line[k]:.6g}')
    cfg = dict(line.get('config', {}))
    line['config'] = {k: _round(v) for k, v in cfg.items() if not isinstance(v, (dict, list)) or k in ('frame_padded', 'bank_tokens_at_end')}
    if 'also_multi_gpu' in result:  # N > 1: the one-clip-on-N-GPUs line of BASELINE configs[4], as scalars
        for e in result['also_multi_gpu']:
            line['config']['fps_4k_one_clip_on_all_gpus'] = _round(e['value'])
            line['config']['collective_bytes_per_frame_rank0'] = _round(e['config'].get('collective_bytes_per_frame_rank0'))
    if 'roofline' in result:
        r = result['roofline']
        out = {k: r[k] for k in ROOFLINE_KEYS if k in r}
        out['kernel'] = 'conv_mfma_kernel (direct) + conv_wino_kernel (Winograd F(2x2,3x3)), fp32 MFMA; achieved = algorithmic flops / time'
        out.update({k: v for k, v in r.items() if k.startswith(('affinity_', 'f16_split_frac_of_f16_peak_', 'f16_split_traffic_'))
                    and not isinstance(v, (dict, list, str))})
        p = r.get('sustained_mfma_probe')
        if isinstance(p, dict) and 'random_operands_tflops' in p:
            out['sustained_mfma_probe_tflops'] = p['random_operands_tflops']
        line['roofline'] = _round(out)
    if 'cpu_baseline' in result:
        c = result['cpu_baseline']
        out = {k: c[k] for k in CPU_KEYS if k in c}
        if 'sample' in out:
            out['sample'] = out['sample'][:160]
        line['cpu_baseline'] = _round(out)
    line['extra'] = 'bench_extra.json'
    text = json.dumps(line, separators=(',', ':'))
    # never over the limit: drop the optional scalars, least important first
    droppable = ([('roofline', k) for k in list(line.get('roofline', {})) if k.startswith('f16_split_')] +
                 [('cpu_baseline', 'sample'), ('cpu_baseline', 'stage_ms_per_frame')] +
                 [('roofline', k) for k in list(line.get('roofline', {})) if k.startswith('affinity_')] +
                 [('config', k) for k in reversed(list(line['config'])) if k.startswith('fps_')])
    while len(text) > LINE_LIMIT and droppable:
        sect, k = droppable.pop(0)
        line[sect].pop(k, None)
        text = json.dumps(line, separators=(',', ':'))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def emit(result):
    """full record -> bench_extra.json (next to bench.py, and gpurun_out/ when it exists) + stderr; then the compact
    contract line as the LAST line of stdout"""
    full = json.dumps(result, indent=1)
    dirs = [os.environ['DEVA_BENCH_EXTRA_DIR']] if os.environ.get('DEVA_BENCH_EXTRA_DIR') else [ROOT, os.path.join(ROOT, 'gpurun_out')]
    for d in dirs:
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, 'bench_extra.json'), 'w') as f:
                    f.write(full + '\n')
            except OSError as exc:
                print(f'bench_extra.json not written in {d}: {exc}', file=sys.stderr)
    print(full, file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact_line(result))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    ap.add_argument('--objects', type=int, default=5)
    ap.add_argument('--workload', choices=['clips', 'long4k'], default='clips')
    ap.add_argument('--long4k_mode', choices=['owner', 'queries', 'bank', 'owner_bank'], default='owner')
    ap.add_argument('--cpu_frames', type=int, default=12, help='propagated frames of one CPU-baseline run (2 of them warm-up); three runs')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_extra', action='store_true')
    ap.add_argument('--no_affinity', action='store_true', help='skip the affinity / pointwise micro-benchmarks (profiling passes)')
    args = ap.parse_args()

    torch.set_grad_enabled(False)
    emulated = os.environ.get('DEVA_BENCH_EMULATED') == '1'  # tests/test_replicas_gloo.py only: launch path on CPU/gloo
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` (the driver's form without torchrun): become N ranks, one per GPU
        if not emulated and torch.cuda.device_count() < args.gpus:
            sys.exit(f'bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible')
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                  f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
                                  os.path.abspath(__file__), *sys.argv[1:]])
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, f'--gpus {args.gpus} but the launcher started {world} rank(s)'
    distributed = world > 1
    if emulated:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import emu_ops

        class _Set:
            setattr = staticmethod(setattr)
        emu_ops.install(_Set)
        device = torch.device('cpu')
        torch.cuda.synchronize = lambda *a, **k: None
    else:
        device = torch.device(f'cuda:{local_rank}')
        torch.cuda.set_device(device)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(backend='gloo' if emulated else 'nccl')  # "nccl" IS RCCL on ROCm
        assert dist.get_world_size() == args.gpus

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
        'rccl_ranks': world if distributed else 0,
    }
    if distributed and not args.no_extra:
        # BASELINE configs[4]: ONE 4K clip on all ranks, frame-owner mode; every rank takes part
        steps4k = max(2, min(20, args.steps))
        tiny = dict(size=(96, 128), bank_tokens=600) if emulated else {}  # (the CPU test of this launch path)
        fps_own, bank_own, comm_own = run_long4k(net, device, steps=steps4k, warmup=min(5, args.warmup), seed=11,
                                                 shard='owner', dist=dist, **tiny)
        result['also_multi_gpu'] = [
            {'metric': f'propagation FPS @4K (1 object, 50k-token long-term bank), ONE clip on {world} GPUs',
             'value': fps_own, 'unit': 'frames/s', 'steps': steps4k, 'ms_per_step': 1e3 / fps_own, 'scaling': 'strong',
             'config': {'workload': 'BASELINE configs[4]: frame owner (rank 0 encodes / decodes) + query-sharded read',
                        'bank_tokens_at_end': bank_own, 'collective_bytes_per_frame_rank0': comm_own}}]

    if rank == 0 and emulated:
        emit(result)
    elif rank == 0:
        # ---- roofline of the dominant kernel: event-timed replay of as many frames, continuing the same clip
        # (un-synchronised events, see ConvTimer); warm the replay with two frames first
        replay = make_clip(args.height, args.width, args.steps + 2, seed=999, device=device)
        for f in replay[:2]:
            core.step(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with ConvTimer() as ct:
            for f in replay[2:]:
                core.step(f)
        replay_ms = (time.perf_counter() - t0) * 1e3 / args.steps
        flops, ms, launches, alg_bytes = ct.summary()
        n_replay = args.steps
        if os.environ.get('DEVA_BENCH_LAYERS'):
            with open(os.environ['DEVA_BENCH_LAYERS'], 'w') as f:
                json.dump(ct.per_layer(n_replay), f, indent=1)
        ach = flops / (ms * 1e-3) / 1e12
        by_class = ct.split_by_precision()
        wf, wms, wn = by_class['wino']
        result['roofline'] = {
            'kernel': 'conv_mfma_kernel (direct implicit GEMM) + conv_wino_kernel (Winograd F(2x2, 3x3) for the big 3x3 layers) '
                      '(+ splitk_reduce_kernel / conv_cout1 kernels of the same deva_conv2d call), fp32 MFMA',
            # `achieved` counts ALGORITHMIC flops (2 K cout pixels of every convolution, the direct count); the Winograd layers
            # execute 1 / 2.25 of theirs on the matrix pipes -- the matrix pipes' own utilisation is `mfma_executed_frac`
            'winograd_share_of_algorithmic_flops': wf / max(flops, 1.0),
            'winograd_ms_per_frame': wms / args.steps,
            'winograd_algorithmic_tflops': wf / max(wms, 1e-9) / 1e9,
            'mfma_executed_tflops': (flops - wf * (1 - 1 / 2.25)) / (ms * 1e-3) / 1e12,
            'mfma_executed_frac': (flops - wf * (1 - 1 / 2.25)) / (ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
            'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP32_MATRIX_TFLOPS, 'unit': 'TFLOP/s',
            'frac': ach / PEAK_FP32_MATRIX_TFLOPS, 'traffic': None,
            'method': 'HIP events around every deva_conv2d launch on the launch stream, no synchronisation, '
                      f'{n_replay} frames replayed after the timed region',
            'launches_per_frame': launches / n_replay,
            'gflop_per_frame': flops / n_replay / 1e9,
            'ms_in_kernel_per_frame': ms / n_replay,
            'ms_per_frame_of_the_instrumented_replay': replay_ms,
            'algorithmic_bytes_per_frame': alg_bytes / n_replay,
        }
        try:
            probe = mfma_probe(device)
            result['roofline']['sustained_mfma_probe'] = probe
            result['roofline']['frac_of_sustained_probe'] = ach / probe['random_operands_tflops']
        except Exception as exc:  # noqa: BLE001
            result['roofline']['sustained_mfma_probe'] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
            torch.cuda.synchronize()
        # HBM traffic of these kernels per frame, from the committed rocprofv3 --pmc passes over this same
        # command (tools/pmc_bench.sh; FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes)
        for pmc_dir in ('pmc_r06', 'pmc_r05', 'pmc_r04', 'pmc_r03'):
            pmc = os.path.join(ROOT, 'profiles', pmc_dir, 'conv_traffic.json')
            if not os.path.exists(pmc):
                continue
            with open(pmc) as f:
                d = json.load(f)
            result['roofline']['traffic'] = d['hbm_bytes_per_frame']
            result['roofline']['traffic_source'] = f'profiles/{pmc_dir}/conv_traffic.json'
            result['roofline']['traffic_over_algorithmic'] = d['hbm_bytes_per_frame'] / (alg_bytes / n_replay)
            # the PMC passes ran `python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_extra --no_affinity` (this
            # workload, this loop); they belong to THIS build if the conv sources hash the same
            result['roofline']['traffic_measured_on_this_build'] = d.get('conv_source_sha1') == conv_source_sha1()
            break
        # ---- the same loop timed the reference's way (evaluation/eval_vos.py:150-186): an event pair + synchronize per
        # frame around step + argmax / id remap (prob_to_obj_cls); no resize at 480p (video_reader.py:139-144)
        try:
            ref_ms = []
            for f in replay[2:2 + min(20, args.steps)]:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                prob = core.step(f)
                core.object_manager.prob_to_obj_cls(prob)
                b.record()
                torch.cuda.synchronize()
                ref_ms.append(a.elapsed_time(b))
            result['timed_like_eval_vos'] = {
                'fps': 1e3 * len(ref_ms) / sum(ref_ms), 'ms_per_frame': sum(ref_ms) / len(ref_ms), 'frames': len(ref_ms),
                'method': 'per frame: start.record(); step(); argmax + id remap; end.record(); torch.cuda.synchronize() '
                          '(evaluation/eval_vos.py:150-186) -- the per-frame synchronize exposes the launch latency of the '
                          'first kernels of every frame, `value` above is the un-synchronised loop'}
        except Exception as exc:  # noqa: BLE001
            result['timed_like_eval_vos'] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
            torch.cuda.synchronize()
        if not args.no_affinity:
            try:
                result['affinity'] = affinity_microbench(device)
            except Exception as exc:  # noqa: BLE001  (never at the cost of the headline line)
                result['affinity'] = {'error': f'{type(exc).__name__}: {exc}'[:400]}
                torch.cuda.synchronize()
            try:
                result['also_kernels'] = (pointwise_rooflines(device, '480p, 5 objects', 5, 30, 54) +
                                          readout_roofline(device, '480p, 5 objects, 16 200-token bank', 5, 1620, 16200) +
                                          pointwise_rooflines(device, '1080p, 1 object', 1, 68, 120) +
                                          readout_roofline(device, '1080p, 1 object, 10 000 + 8 160-token bank', 1, 8160, 18160))
            except Exception as exc:  # noqa: BLE001
                result['also_kernels'] = {'error': f'{type(exc).__name__}: {exc}'[:400]}
                torch.cuda.synchronize()
        if not args.no_extra:
            del core
            result['also'] = extra_lines(net, device, cfg, args)
            # the lines a reader looks for first, as scalars inside `config` / `roofline` (a record that keeps only the
            # standard keys of the contract still carries them; full entries: `also`)
            short = (('fps_480p_1obj', '@480p (1 object'), ('fps_1080p_1obj_10k_bank', '@1080p (1 object, 10k'),
                     ('fps_1080p_8seg', '@1080p (8-segment'), ('fps_1080p_8seg_amp', '@1080p, --amp'),
                     ('fps_480p_5obj_f16_split', '@480p, --f16_split'), ('fps_1080p_1obj_10k_bank_f16_split', '@1080p, --f16_split (1 object'),
                     ('fps_1080p_8seg_f16_split', '@1080p, --f16_split (8-segment'),
                     ('fps_480p_5obj_f16_split_key_encoder', '@480p, --f16_split --f16_split_key_encoder'),
                     ('fps_1080p_1obj_10k_bank_f16_split_key_encoder', '@1080p, --f16_split --f16_split_key_encoder (1 object'),
                     ('fps_1080p_8seg_f16_split_key_encoder', '@1080p, --f16_split --f16_split_key_encoder (8-segment'),
                     ('fps_4k_1obj_50k_bank', '@4K (1 object'), ('fps_4k_1obj_50k_bank_f16_split_key_encoder', '@4K, --f16_split'))
            for key, frag in short:
                for e in result['also']:
                    if frag in e['metric'] and e.get('value') is not None:
                        result['config'][key] = round(e['value'], 2)
                        cr = (e.get('config', {}).get('state_at_end') or {}).get('conv_roofline', {}).get('split_kernels')
                        if cr and 'f16_split' in key:
                            result['roofline'][key.replace('fps_', 'f16_split_fp32_equiv_tflops_')] = round(cr['fp32_equivalent_tflops'], 1)
                            result['roofline'][key.replace('fps_', 'f16_split_frac_of_f16_peak_')] = round(cr['frac_of_f16_mfma_peak'], 3)
                        if key == 'fps_1080p_8seg_f16_split_key_encoder':
                            # HBM counter traffic of this clip's convolution kernels (tools/pmc_split.sh, committed summary)
                            pmc = os.path.join(ROOT, 'profiles', 'pmc_r06', 'conv_traffic_split.json')
                            if os.path.exists(pmc):
                                with open(pmc) as f:
                                    d = json.load(f)
                                e['config']['hbm_counter_traffic'] = {
                                    'hbm_bytes_per_frame': d['hbm_bytes_per_frame'], 'algorithmic_bytes_per_frame': d['algorithmic_bytes_per_frame'],
                                    'traffic_over_algorithmic': d['traffic_over_algorithmic'], 'source': 'profiles/pmc_r06/conv_traffic_split.json',
                                    'measured_on_this_build': d.get('conv_source_sha1') == conv_source_sha1()}
                                if d.get('traffic_over_algorithmic'):
                                    result['roofline']['f16_split_traffic_over_algorithmic_1080p_8seg_key_encoder'] = round(d['traffic_over_algorithmic'], 3)
                        break
        if isinstance(result.get('affinity'), dict) and 'us_read' in result['affinity']:
            a = result['affinity']
            result['roofline']['affinity_read_us_10k_x_8160'] = round(a['us_read'], 1)
            result['roofline']['affinity_read_us_10k_x_8160_bank_cached'] = round(a['us_read_bank_operands_cached'], 1)
            result['roofline']['affinity_f16_mfma_frac'] = round(a['f16_mfma_frac'], 4)
            result['roofline']['affinity_hbm_algorithmic_frac'] = round(a['hbm_algorithmic_frac'], 5)
            if a.get('hbm_counter_traffic_over_algorithmic'):
                result['roofline']['affinity_hbm_counter_traffic_over_algorithmic'] = round(a['hbm_counter_traffic_over_algorithmic']['ratio'], 2)
        if not args.no_cpu_baseline and world == 1:
            frames_cpu = [f.cpu() for f in frames[:1 + min(args.cpu_frames, len(frames) - 1)]]
            result['cpu_baseline'] = cpu_baseline(sd, cfg, args.height, args.width, args.objects, frames_cpu)
            try:
                result['cpu_baseline']['affinity_kernels'] = cpu_affinity_kernels()
            except Exception as exc:  # noqa: BLE001
                result['cpu_baseline']['affinity_kernels'] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
        emit(result)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
