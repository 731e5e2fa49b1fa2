"""Fused similarity -> top-k -> softmax -> usage kernel and the sparse readout, against the
reference's own outputs (tests/golden/memory_ops.pt) and the CPU contract at larger sizes.

Exactness: the kernel reproduces the fp32 FMA chains of the CPU GEMMs, so the selected token SETS
must be identical wherever the k-th / (k+1)-th scores are not within a few ulp of each other; the
test measures that gap and only tolerates a differing index where the scores tie to 1e-6 relative.
Weights: 1e-5; readout: 1e-4."""
import os

import pytest
import torch

import emu_ops
from deva.hip import ops
from gpu_util import dev, max_err, to_dev
from workload import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _run(mk, ms, qk, qe, k, n_long=0, splits=None, usage=True):
    """mk [64,N] channel-major like the reference -> token-major arenas split at n_long"""
    n = mk.shape[1]
    rows, shr = mk.t().contiguous(), ms.reshape(-1).contiguous()
    kl, sl = (to_dev(rows[:n_long]), to_dev(shr[:n_long])) if n_long else (None, None)
    kw, sw = to_dev(rows[n_long:].contiguous()), to_dev(shr[n_long:].contiguous())
    fix = torch.zeros(n, dtype=torch.int64, device=dev()) if usage else None
    idx, w = ops.affinity_topk(kl, sl, n_long, kw, sw, n - n_long, to_dev(qk), to_dev(qe), k, fix, splits)
    torch.cuda.synchronize()
    return idx.cpu(), w.cpu(), (fix.cpu().double() / 2**40).float() if usage else None


def _compare(name, idx, w, usage, sim, k):
    """sim: reference similarity [N,HW] (CPU).  Returns number of queries whose index set differs."""
    vals, ridx = torch.topk(sim, k=k + 1, dim=0)
    rw = vals[:k].exp()
    rw = rw / rw.sum(0, keepdim=True)
    hw = sim.shape[1]
    bad = 0
    for q in range(hw):
        a, b = set(idx[q].tolist()), set(ridx[:k, q].tolist())
        if a != b:
            gap = (vals[k - 1, q] - vals[k, q]).abs().item()
            scale = vals[k - 1, q].abs().item() + 1e-30
            assert gap <= 1e-6 * scale, f'{name}: query {q} index set differs with a non-tie gap {gap:.3e}'
            bad += 1
    # order: descending score
    got_scores = torch.gather(sim.t(), 1, idx.long())
    assert (got_scores[:, :-1] >= got_scores[:, 1:]).all(), f'{name}: not sorted by score'
    same = torch.tensor([idx[q].tolist() == ridx[:k, q].tolist() for q in range(hw)])
    # weights; queries whose scores all underflow give 0/0 = NaN in the reference (no max
    # subtraction, memory_utils.py:59-60) and must give NaN here too
    ref_w = rw.t()
    # where even the best score is below -80 exp() lands in the fp32 denormal range and CPU / GPU
    # flush differently (0/0 = NaN vs tiny finite): outside the operating range, not compared
    sane = (vals[0] > -80.0)
    nan_ref, nan_got = torch.isnan(ref_w), torch.isnan(w)
    assert torch.equal(nan_ref[sane], nan_got[sane]), f'{name}: NaN pattern differs from the reference'
    fin = (same & sane)[:, None] & ~nan_ref
    werr = (w[fin] - ref_w[fin]).abs().max().item() if fin.any() else 0.0
    assert werr <= 1e-5, f'{name}: weight error {werr:.3e}'
    if usage is not None:
        dense = torch.zeros_like(sim).scatter_(0, ridx[:k], torch.nan_to_num(rw))
        uerr = (usage - dense.sum(1)).abs().max().item()
        if bad == 0 and bool(sane.all()):
            assert uerr <= 1e-4, f'{name}: usage error {uerr:.3e}'
    print(f'{name}: N={sim.shape[0]} HW={hw} tie-swapped queries={bad} identical order={int(same.sum())}/{hw} '
          f'weight err={werr:.2e}')
    return bad


def test_against_reference_golden(golden_dir):
    cases = torch.load(os.path.join(golden_dir, 'memory_ops.pt'))
    for name, c in cases.items():
        mk, ms, qk, qe = synth.affinity_inputs(c['n'], c['hw'], seed=c['seed'], key_scale=c['scale'])
        for n_long, splits in [(0, None), (c['n'] // 3, 1), (c['n'] // 2 + 1, 3)]:
            idx, w, usage = _run(mk, ms, qk, qe, 30, n_long, splits)
            _compare(f'{name}/long{n_long}/s{splits}', idx, w, usage, c['sim'], 30)
        # readout through the real sparse kernel vs the reference's dense matmul
        idx, w, usage = _run(mk, ms, qk, qe, 30)
        v = synth.value_inputs(2, 512, c['n'], seed=c['seed'])
        for o in range(2):
            out = torch.empty(512, c['hw'], device=dev())
            ops.readout_sparse(to_dev(idx), to_dev(w), None, 0, to_dev(v[o].t().contiguous()), out)
            err = max_err(out, c['readout'][o])
            print(f'{name}: readout obj{o} max abs err {err:.3e}')
            assert err <= 1e-4 * max(1.0, c['readout'].abs().max().item())
        assert max_err(usage, c['usage']) <= 1e-4


@pytest.mark.parametrize('n,hw,scale,k', [(5000, 1620, 4.0, 30), (2000, 257, 1.0, 30), (64, 40, 1.0, 30),
                                           (33, 1, 1.0, 30), (999, 129, 0.2, 7), (4096, 128, 2.0, 32),
                                           (10000, 300, 3.0, 1)])
def test_larger_shapes_against_cpu(n, hw, scale, k):
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=n + hw, key_scale=scale)
    from oracle import deva_oracle as O
    sim = O.get_similarity(mk, ms, qk, qe)
    base = None
    for n_long, splits in [(0, None), (n // 4, 1), (n // 2, 5), (n - 1, 2)]:
        if n_long >= n or (splits and splits > max(1, n // 32)):
            continue
        idx, w, usage = _run(mk, ms, qk, qe, k, n_long, splits)
        _compare(f'n{n}hw{hw}/long{n_long}/s{splits}', idx, w, usage, sim, k)
        if base is None:
            base = (idx, w)
        else:  # split -> merge must be bit-identical to the unsplit result
            assert torch.equal(idx, base[0]) and torch.equal(torch.nan_to_num(w, nan=-1.0), torch.nan_to_num(base[1], nan=-1.0)), \
                'result depends on the split count'


@pytest.mark.parametrize('n,hw,scale,k', [(5000, 1620, 4.0, 48), (2000, 257, 1.0, 64), (80, 40, 1.0, 64), (999, 129, 0.2, 33)])
def test_top_k_above_32_runs_on_the_dense_kernel(n, hw, scale, k):
    """eval_args.py:40 leaves --top_k free; 33..64 are served by deva_affinity_dense behind the same entry point"""
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=n + hw + k, key_scale=scale)
    from oracle import deva_oracle as O
    sim = O.get_similarity(mk, ms, qk, qe)
    base = None
    for n_long in (0, n // 3):
        idx, w, usage = _run(mk, ms, qk, qe, k, n_long)
        assert tuple(idx.shape) == (hw, k)
        _compare(f'dense n{n}hw{hw}k{k}/long{n_long}', idx, w, usage, sim, k)
        if base is None:
            base = (idx, w)
        else:
            assert torch.equal(idx, base[0]) and torch.equal(torch.nan_to_num(w, nan=-1.0), torch.nan_to_num(base[1], nan=-1.0)), \
                'result depends on where the bank is split into long-term and working segment'
    with pytest.raises(Exception, match='unsupported'):
        _run(mk, ms, qk, qe, 65)


def _run_dense(mk, ms, qk, qe, k, n_long=0):
    n = mk.shape[1]
    rows, shr = mk.t().contiguous(), ms.reshape(-1).contiguous()
    kl, sl = (to_dev(rows[:n_long]), to_dev(shr[:n_long])) if n_long else (None, None)
    kw, sw = to_dev(rows[n_long:].contiguous()), to_dev(shr[n_long:].contiguous())
    fix = torch.zeros(n, dtype=torch.int64, device=dev())
    idx, w = ops.affinity_dense(kl, sl, n_long, kw, sw, n - n_long, to_dev(qk), to_dev(qe), k, fix)
    torch.cuda.synchronize()
    return idx.cpu(), w.cpu(), (fix.cpu().double() / 2**40).float()


@pytest.mark.parametrize('n,hw,k,n_long', [(4096, 2000, 30, 0), (5000, 333, 32, 1200), (70, 65, 30, 7), (700, 64, 1, 0),
                                            (9000, 1000, 30, 4000)])
def test_dense_kernel_is_bit_identical_to_the_list_kernels(n, hw, k, n_long):
    """same fp32 FMA chains, same (score, index) order, same exp / sequential sum: where both kernels apply the dense one
    must reproduce the list kernels' indices, weights and usage counters bit for bit (the last shape takes the fp16
    pre-filter + exact re-scoring on the other side)"""
    if os.environ.get('DEVA_TEST_DRYRUN') == '1':
        pytest.skip('kernel-path property: nothing to compare on the emulated ops')
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=3 * n + hw, key_scale=1.5)
    _assert_identical(f'dense vs lists n{n}hw{hw}k{k}', _run(mk, ms, qk, qe, k, n_long), _run_dense(mk, ms, qk, qe, k, n_long))


def _both_paths(mk, ms, qk, qe, k, n_long=0):
    """the same read through the fp32 kernels (pre-filter off) and through the automatic choice"""
    from deva.hip import lib
    if os.environ.get('DEVA_TEST_DRYRUN') == '1':
        pytest.skip('kernel-path property: nothing to compare on the emulated ops')
    out = []
    try:
        for mode in (0, 1):
            lib().deva_affinity_force_prefilter(mode)
            out.append(_run(mk, ms, qk, qe, k, n_long))
        flag = ops.affinity_last_read_flag(dev())
    finally:
        lib().deva_affinity_force_prefilter(1)
    return out[0], out[1], flag


def _assert_identical(tag, a, b):
    (i0, w0, u0), (i1, w1, u1) = a, b
    assert torch.equal(i0, i1), f'{tag}: {int((i0 != i1).any(1).sum())} queries select different tokens'
    assert torch.equal(w0.view(torch.int32), w1.view(torch.int32)), f'{tag}: weights differ'
    assert torch.equal(u0.view(torch.int32), u1.view(torch.int32)), f'{tag}: usage counters differ'


@pytest.mark.parametrize('n,hw,scale,k,n_long', [(4096, 2000, 1.0, 30, 0), (5000, 1620, 2.0, 30, 1200), (10001, 801, 1.0, 30, 7),
                                                 (40000, 257, 0.3, 30, 39999), (5000, 1700, 30.0, 30, 0),
                                                 (8200, 4100, 1.0, 5, 100), (4096, 2048, 0.01, 32, 0),
                                                 (6000, 4099, 1.0, 30, 0)])
def test_fp16_prefilter_is_bit_identical_to_the_fp32_kernels(n, hw, scale, k, n_long):
    """deva_affinity_read: fp16 MFMA bounds -> group-maxima threshold -> candidates -> exact fp32 re-scoring must
    reproduce the fp32 kernels bit for bit (indices, order, weights, usage) WITHOUT falling back, on ragged sizes,
    a long + working bank, large / tiny key magnitudes (the operand scales are data-dependent powers of two)"""
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=n + hw, key_scale=scale)
    from deva.hip import lib
    assert lib().deva_affinity_prefilter_enabled(n, hw, k) == 1, 'shape below the library\'s pre-filter policy'
    fp32, pre, flag = _both_paths(mk, ms, qk, qe, k, n_long)
    _assert_identical(f'n{n}hw{hw}', fp32, pre)
    assert flag == 0, f'the pre-filter fell back (flag {flag})'


def test_fp16_prefilter_falls_back_where_its_bound_does_not_hold():
    """inputs outside the bound's premises must raise the device flag and still give the fp32 kernels' result:
    a negative selection value (the Cauchy-Schwarz step needs qe >= 0), a non-finite key, and a flat bank (every
    token identical: every score is a candidate, the sub-lists overflow)"""
    n, hw, k = 8192, 1024, 30  # (large enough for the library to choose the pre-filter)
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=5)
    qe_neg = qe.clone()
    qe_neg[3, 17] = -0.25
    fp32, pre, flag = _both_paths(mk, ms, qk, qe_neg, k)
    _assert_identical('negative selection', fp32, pre)
    assert flag & 2, flag
    mk_inf = mk.clone()
    mk_inf[5, 1000] = float('inf')
    fp32, pre, flag = _both_paths(mk_inf, ms, qk, qe, k)
    assert flag & 1, flag
    assert torch.equal(fp32[0], pre[0])
    flat = mk[:, :1].repeat(1, n).contiguous()
    fp32, pre, flag = _both_paths(flat, torch.ones_like(ms), qk, qe, k)
    _assert_identical('flat bank', fp32, pre)
    assert flag & 12, flag


def test_fp16_prefilter_fallback_counts_usage_once():
    """A few LATE queries with 2 049 .. 4 096 candidates spread thinly over the sub-lists (no sub-list overflows: flag 8
    alone, not 4): the fall-back decision must be taken before the re-score kernel has added any query's weights to
    the usage counters, otherwise the fp32 fall-back counts the earlier queries twice (affinity.hip)."""
    n, hw, k = 9000, 2048, 30
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=11)
    mk, ms, qk, qe = mk.clone(), ms.clone(), qk.clone(), qe.clone()
    dup = torch.arange(0, n, 3)  # 3 000 identical tokens, spread over every token range
    mk[:, dup] = mk[:, :1].clone()
    ms[:, dup] = ms[:, :1].clone()
    for q in (hw - 1, hw - 7, hw - 300):  # the duplicated key is THE best match of these queries: 3 000-way tie
        qk[:, q] = mk[:, 0]
        qe[:, q] = 1.0
    fp32, pre, flag = _both_paths(mk, ms, qk, qe, k)
    assert flag & 8 and not (flag & 4), f'expected the re-score capacity flag alone, got {flag}'
    _assert_identical('thinly spread candidates', fp32, pre)


def test_fp16_prefilter_shard_keys_equal_the_fp32_select():
    """the hand-over format of a bank shard (affinity_candidates) through both paths"""
    if os.environ.get('DEVA_TEST_DRYRUN') == '1':
        pytest.skip('kernel-path property: nothing to compare on the emulated ops')
    n, hw, k = 6000, 1400, 30
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=9)
    rows, shr = to_dev(mk.t().contiguous()), to_dev(ms.reshape(-1).contiguous())
    new = ops.affinity_candidates(None, None, 0, rows, shr, n, to_dev(qk), to_dev(qe), k, token_offset=12345)
    assert ops.affinity_last_read_flag(dev()) == 0  # (before the fp32 call below reuses the scratch buffer)
    old = ops.affinity_candidates(None, None, 0, rows, shr, n, to_dev(qk), to_dev(qe), k, token_offset=12345, splits=4)
    torch.cuda.synchronize()
    assert torch.equal(new[1], old[1]) and torch.equal(new[0][:, :k], old[0][:, :k])


@pytest.mark.parametrize('shape', [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_kernel_shape_gives_the_same_result(shape):
    """the kernel shapes of deva_affinity_topk (per-wave / workgroup-shared lists, one / two workgroups per CU,
    shared key tiles, early / late prefetch) forced in turn: identical indices, weights and usage counters as the
    automatic choice, on a bank with a long-term part, ragged sizes and enough tokens for several prune rounds"""
    from deva.hip import check, lib
    cases = [(5000, 1620, 2.0, 30, 1200), (999, 129, 0.2, 7, 0), (20000, 257, 1.0, 30, 333), (33, 1, 1.0, 30, 0)]
    if lib().deva_affinity_force_shape(shape) != 0:
        lib().deva_affinity_force_shape(0)
        pytest.skip('A/B variant of probe builds (make PROBES=1); the product library carries shapes 2, 4 and 8')
    try:
        lib().deva_affinity_force_prefilter(0)  # the shapes belong to the fp32 kernels
        for n, hw, scale, k, n_long in cases:
            mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=n + hw + 1, key_scale=scale)
            check(lib().deva_affinity_force_shape(0), 'force_shape')
            want = _run(mk, ms, qk, qe, k, n_long)
            check(lib().deva_affinity_force_shape(shape), 'force_shape')
            got = _run(mk, ms, qk, qe, k, n_long)
            assert torch.equal(got[0], want[0]) and torch.equal(torch.nan_to_num(got[1], nan=-1.0), torch.nan_to_num(want[1], nan=-1.0))
            assert torch.equal(got[2], want[2]), (shape, n, hw)
    finally:
        lib().deva_affinity_force_shape(0)
        lib().deva_affinity_force_prefilter(1)


def test_determinism_and_usage_clear():
    mk, ms, qk, qe = synth.affinity_inputs(3000, 500, seed=9, key_scale=2.0)
    a = _run(mk, ms, qk, qe, 30)
    b = _run(mk, ms, qk, qe, 30)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    fix = torch.zeros(3000, dtype=torch.int64, device=dev())
    fix[100:200] = 5 << 40
    use, life = torch.zeros(100, device=dev()), torch.ones(100, device=dev())
    ops.usage_update(fix, 100, use, life, 100)
    assert (fix == 0).all() and (use == 5).all() and (life == 2).all()


def test_k_larger_than_bank_raises():
    mk, ms, qk, qe = synth.affinity_inputs(20, 10, seed=1)
    with pytest.raises(Exception):
        _run(mk, ms, qk, qe, 30)


def test_readout_two_segments_ragged():
    g = torch.Generator().manual_seed(3)
    n_long, n_work, hw, k, cv = 70, 130, 45, 30, 512
    idx = torch.stack([torch.randperm(n_long + n_work, generator=g)[:k] for _ in range(hw)]).int()
    w = torch.rand(hw, k, generator=g)
    vl, vw = torch.randn(n_long, cv, generator=g), torch.randn(n_work, cv, generator=g)
    want = torch.empty(cv, hw)
    emu_ops.readout_sparse(idx, w, vl, n_long, vw, want)
    out = torch.empty(cv, hw, device=dev())
    ops.readout_sparse(to_dev(idx), to_dev(w), to_dev(vl), n_long, to_dev(vw), out)
    assert max_err(out, want) <= 1e-4


# bit-identity across differently shaped launches is a property of the kernels (fixed accumulation order per score),
# not of the PyTorch emulation the dry run substitutes (a float32 BLAS product depends on its blocking)
kernel_property = pytest.mark.skipif(os.environ.get('DEVA_TEST_DRYRUN') == '1',
                                     reason='checks a property of the HIP kernels; the dry run emulates them')


@kernel_property
def test_query_column_slices_are_bit_identical():
    """the multi-GPU read shards the queries by column (MemoryManager.shard_queries): every column's
    result must not depend on which other columns share the launch, and the fixed-point usage
    counters of the shards must add up to the unsharded counters exactly"""
    n, hw, k = 4000, 1000, 30
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=77, key_scale=2.0)
    idx, w, _ = _run(mk, ms, qk, qe, k, n_long=1500)
    rows, shr = mk.t().contiguous(), ms.reshape(-1).contiguous()
    kl, sl, kw, sw = to_dev(rows[:1500]), to_dev(shr[:1500]), to_dev(rows[1500:].contiguous()), to_dev(shr[1500:].contiguous())
    full_fix = torch.zeros(n, dtype=torch.int64, device=dev())
    ops.affinity_topk(kl, sl, 1500, kw, sw, n - 1500, to_dev(qk), to_dev(qe), k, full_fix)
    for world in (2, 3, 8):
        per = -(-hw // world)
        fix = torch.zeros(n, dtype=torch.int64, device=dev())
        for r in range(world):
            lo, hi = r * per, min(hw, (r + 1) * per)
            i_s, w_s = ops.affinity_topk(kl, sl, 1500, kw, sw, n - 1500, to_dev(qk[:, lo:hi].contiguous()),
                                         to_dev(qe[:, lo:hi].contiguous()), k, fix)
            assert torch.equal(i_s.cpu(), idx[lo:hi]) and torch.equal(w_s.cpu(), w[lo:hi]), (world, r)
        assert torch.equal(fix, full_fix), world


@kernel_property
@pytest.mark.parametrize('world', [2, 3, 8])
def test_token_sharded_read_merges_to_the_unsharded_result(world):
    """bank sharded by token range (MemoryManager.shard_bank): every shard's own top-k, in the hand-over
    format and with global token ids, merged by deva_affinity_merge must be BIT-identical to the unsharded
    read (indices, weights, usage counters); the shards' partial read-outs must add up to the full one"""
    n, hw, k, n_long = 5000, 700, 30, 1800
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=world, key_scale=2.0)
    rows, shr = mk.t().contiguous(), ms.reshape(-1).contiguous()
    kl, sl = to_dev(rows[:n_long].contiguous()), to_dev(shr[:n_long].contiguous())
    kw, sw = to_dev(rows[n_long:].contiguous()), to_dev(shr[n_long:].contiguous())
    qk_d, qe_d = to_dev(qk), to_dev(qe)
    full_fix = torch.zeros(n, dtype=torch.int64, device=dev())
    idx, w = ops.affinity_topk(kl, sl, n_long, kw, sw, n - n_long, qk_d, qe_d, k, full_fix)
    per = -(-n // world)
    keys, counts = [], []
    for r in range(world):
        lo, hi = r * per, min(n, (r + 1) * per)
        l0, l1 = min(lo, n_long), min(hi, n_long)
        w0, w1 = max(lo, n_long) - n_long, max(hi, n_long) - n_long
        kk, cc = ops.affinity_candidates(kl[l0:l1] if l1 > l0 else None, sl[l0:l1] if l1 > l0 else None, l1 - l0,
                                         kw[w0:w1] if w1 > w0 else None, sw[w0:w1] if w1 > w0 else None, w1 - w0,
                                         qk_d, qe_d, k, token_offset=lo)
        assert int(cc.min()) == k and int(cc.max()) == k
        keys.append(kk)
        counts.append(cc)
    fix = torch.zeros(n, dtype=torch.int64, device=dev())
    idx_m, w_m = ops.affinity_merge(torch.stack(keys), torch.stack(counts), k, fix)
    torch.cuda.synchronize()
    assert torch.equal(idx_m, idx) and torch.equal(w_m, w) and torch.equal(fix, full_fix)
    cv = 512
    v = synth.value_inputs(1, cv, n, seed=3)[0].t().contiguous()
    vl, vw = to_dev(v[:n_long].contiguous()), to_dev(v[n_long:].contiguous())
    full = torch.empty(cv, hw, device=dev())
    ops.readout_sparse(idx, w, vl, n_long, vw, full)
    total = torch.zeros_like(full)
    for r in range(world):
        part = torch.empty_like(full)
        ops.readout_sparse(idx, w, vl, n_long, vw, part, tok_range=(r * per, min(n, (r + 1) * per)))
        total += part
    err = max_err(total, full.cpu())
    print(f'world={world}: merged selection bit-identical; partial read-outs sum to the full one within {err:.2e}')
    assert err <= 1e-5 * max(1.0, full.abs().max().item())


def _read(rows, shr, n_long, qk, qe, k, prep=None, key=None):
    """one read through ops.affinity_topk on device-resident banks; -> (idx, weight, usage) on the host"""
    n = rows.shape[0]
    fix = torch.zeros(n, dtype=torch.int64, device=dev())
    kl, sl = (rows[:n_long], shr[:n_long]) if n_long else (None, None)
    idx, w = ops.affinity_topk(kl, sl, n_long, rows[n_long:], shr[n_long:], n - n_long, qk, qe, k, fix, prep=prep, prep_key=key)
    torch.cuda.synchronize()
    return idx.cpu(), w.cpu(), (fix.cpu().double() / 2**40).float()


@pytest.mark.parametrize('n,hw,n_long', [(10000, 2040, 0), (9000, 1620, 4000)])
def test_prepared_bank_read_is_bit_identical_and_follows_the_bank(n, hw, n_long):
    """deva_affinity_read_prepared: the bank side of the pre-filter (mean key, scales, fp16 fragments) kept between reads.
    (1) filling the buffer and (2) re-using it for other queries give the plain read's result bit for bit; (3) after the
    bank has changed -- same sizes, other rows -- a read under a NEW key gives the fresh result; (4) a buffer that is
    too small for a grown bank is replaced; no read falls back."""
    if os.environ.get('DEVA_TEST_DRYRUN') == '1':
        pytest.skip('kernel-path property: nothing to compare on the emulated ops')
    k = 30
    mk, ms, qk, qe = synth.affinity_inputs(n, hw, seed=n + 7 * hw, key_scale=1.3)
    mk2, ms2, qk2, qe2 = synth.affinity_inputs(n + 3000, hw, seed=11, key_scale=0.7)
    rows, shr = to_dev(mk.t().contiguous()), to_dev(ms.reshape(-1).contiguous())
    rows2, shr2 = to_dev(mk2.t().contiguous()), to_dev(ms2.reshape(-1).contiguous())
    q = [(to_dev(a), to_dev(b)) for a, b in ((qk, qe), (qk2, qe2))]
    prep = ops.BankPrep()
    first = _read(rows, shr, n_long, *q[0], k, prep, 'v1')
    _assert_identical('fill', _read(rows, shr, n_long, *q[0], k), first)
    assert ops.affinity_last_read_flag(dev()) == 0
    again = _read(rows, shr, n_long, *q[1], k, prep, 'v1')                       # cached operands, other queries
    _assert_identical('re-use', _read(rows, shr, n_long, *q[1], k), again)
    assert ops.affinity_last_read_flag(dev()) == 0
    rows[n_long + 5:n_long + 4000].copy_(rows2[:3995])                           # the bank changes in place ...
    shr[100:2000].mul_(1.7)
    fresh = _read(rows, shr, n_long, *q[1], k, prep, 'v2')                       # ... and the owner says so
    _assert_identical('after a change', _read(rows, shr, n_long, *q[1], k), fresh)
    assert not torch.equal(fresh[0], again[0]), 'the changed bank must change the read'
    grown = _read(rows2, shr2, n_long, *q[0], k, prep, 'v3')                     # more tokens than the buffer holds
    _assert_identical('grown bank', _read(rows2, shr2, n_long, *q[0], k), grown)
    _assert_identical('grown bank, cached', _read(rows2, shr2, n_long, *q[1], k), _read(rows2, shr2, n_long, *q[1], k, prep, 'v3'))
