"""The error bound of the fp16 pre-filter (csrc/affinity.hip, `affinity_pf_*`), executable on the CPU.

The kernels' exactness rests on an inequality, not on a tolerance: for every (token, query) pair
    Q~ - (1 + d2) P~ - (ABS + E_q)  <=  sim_fp32 * S_q  <=  Q~ - (1 - d2) P~ + (ABS + E_q)
where P~ / Q~ are the two f16-MFMA chains on operands centred on the bank's mean key, scaled by data-dependent powers of
two and rounded to fp16, and sim_fp32 is the reference's fp32 similarity (memory_utils.py:6-45).  This file restates the
operand preparation of `affinity_pf_prep_kernel` / `pf_query_operand` and the threshold of pass A / `pf_tau` with
torch.half roundings and checks, on banks chosen to stress each term (random keys, keys with a large common component,
queries next to memory tokens so that A, B, bsq cancel 100-fold, tiny and huge magnitudes, a ragged bank):
  * the inequality holds for every pair (and how much of the slack is used);
  * every token of the reference's top-k is among the candidates `hi >= k-th largest group maximum of lo - 2 (ABS + E_q)`;
  * the candidate count stays near k on uncorrelated banks.
The GPU tests assert the consequence (bit-identical indices / weights / usage against the fp32 kernels)."""
import math

import pytest
import torch

from oracle import deva_oracle as O

D2, ABS, EQ = 2.63e-3, 600.0, 4e-5   # PF_D2, PF_ABS, PF_EQ of csrc/affinity.hip
GROUPS, SPLITS = 32, 16


def pf_scale(x: float) -> float:
    """power of two P with x * P in [2^14, 2^15) (affinity.hip:pf_scale)"""
    x = float(x)
    if not (x > 0) or math.isinf(x):
        return 1.0
    _, e = math.frexp(x)
    return math.ldexp(1.0, max(-100, min(100, 15 - e)))


def prefilter(mk, ms, qk, qe, k):
    """mk [N,64] token-major, ms [N], qk/qe [64,HW] -> (lo, hi, slack) in the queries' scaled units, S [HW], candidates"""
    n, hw = mk.shape[0], qk.shape[1]
    mu = mk.mean(0)
    m = ms * 0.125
    mkc, qkc = mk - mu, qk - mu[:, None]
    p_el, q_el = (mkc * mkc) * m[:, None], 2 * mkc * m[:, None]
    sp, sq, sm = pf_scale(p_el.max()), pf_scale(q_el.abs().max()), pf_scale(m.max())
    a_p, a_q, a_m = (p_el * sp).half().float(), (q_el * sq).half().float(), (m * sm).half().float()
    bsq, p = (qe * qkc * qkc).sum(0), qkc * qe
    S = torch.tensor([min(sp * pf_scale(qe[:, j].max()), sq * pf_scale(p[:, j].abs().max()), sm * pf_scale(bsq[j]))
                      for j in range(hw)])
    e_q = EQ * float(m.max()) * (qe * (mu * mu)[:, None]).sum(0) * S
    b_e, b_p, b_b = (qe * (S / sp)).half().float(), (p * (S / sq)).half().float(), (bsq * (S / sm)).half().float()
    acc_p = a_p @ b_e + a_m[:, None] * b_b[None, :]
    acc_q = a_q @ b_p
    lo, hi = acc_q - (1 + D2) * acc_p, acc_q - (1 - D2) * acc_p
    slack = ABS + e_q
    # pass A / pf_tau: tile-cyclic ranges, one group per (range, token slot)
    tiles = (n + 31) // 32
    lo_t = torch.cat([lo, torch.full((tiles * 32 - n, hw), -float('inf'))]).view(tiles, 32, hw)
    splits = max(1, min(SPLITS, tiles // 2))
    gmax = torch.cat([lo_t[s::splits].max(0)[0] for s in range(splits)])
    thr = torch.topk(gmax, k, dim=0)[0][k - 1] - 2 * slack
    return lo, hi, slack, S, hi >= thr[None, :], acc_p


def bank(name, n, hw, g):
    base = torch.randn(1, 64, generator=g)
    if name == 'random':
        mk, shift, scale = torch.randn(n, 64, generator=g), 0.0, 1.0
    elif name == 'common component':      # keys = large shared vector + small spread (what a real clip looks like)
        mk, shift, scale = torch.randn(n, 64, generator=g), 20.0, 1.0
    elif name == 'tiny':
        mk, shift, scale = torch.randn(n, 64, generator=g), 0.0, 1e-3
    elif name == 'huge':
        mk, shift, scale = torch.randn(n, 64, generator=g), 3.0, 300.0
    mk = mk * scale + base * shift * scale
    ms = torch.rand(n, generator=g) + 1
    pick = torch.randint(0, n, (hw,), generator=g)   # every query sits next to a memory token: A, B, bsq cancel
    qk = (mk[pick] + 0.3 * scale * torch.randn(hw, 64, generator=g)).t().contiguous()
    qe = torch.rand(64, hw, generator=g)
    return mk, ms, qk, qe


@pytest.mark.parametrize('name,n', [('random', 6000), ('common component', 6000), ('tiny', 4100), ('huge', 4099)])
def test_bound_holds_and_covers_the_reference_top_k(name, n):
    g = torch.Generator().manual_seed(len(name) + n)
    hw, k = 192, 30
    mk, ms, qk, qe = bank(name, n, hw, g)
    sim = O.get_similarity(mk.t().contiguous(), ms.view(1, -1), qk, qe)   # the reference's fp32 similarity [N, HW]
    lo, hi, slack, S, cand, acc_p = prefilter(mk, ms, qk, qe, k)
    sim_s = sim * S[None, :]
    assert int(((lo - slack[None, :] > sim_s) | (hi + slack[None, :] < sim_s)).sum()) == 0, 'the bound is violated'
    used = ((0.5 * (lo + hi) - sim_s).abs() / (D2 * acc_p + slack[None, :])).max().item()
    top = torch.topk(sim, k, dim=0)[1]
    assert bool(cand.gather(0, top).all()), 'a token of the reference top-k is not a candidate'
    per_query = cand.sum(0).float()
    print(f'{name}: slack used {used:.2f}, candidates per query mean {per_query.mean():.1f} max {int(per_query.max())}')
    assert used < 1.0
    if name in ('random', 'tiny'):
        assert per_query.mean() < 2 * k


def test_uncentred_operands_would_not_filter_a_real_looking_bank():
    """why the operands are centred: with the common component left in, P is ~100x the score it cancels to and the
    same relative bound admits a large part of the bank"""
    g = torch.Generator().manual_seed(3)
    mk, ms, qk, qe = bank('common component', 4096, 96, g)
    k = 30
    sim = O.get_similarity(mk.t().contiguous(), ms.view(1, -1), qk, qe)
    m = ms * 0.125
    p_unc = m[:, None] * ((mk * mk) @ qe + (qe * qk * qk).sum(0)[None, :])
    lo, hi = sim - D2 * p_unc, sim + D2 * p_unc
    thr = torch.topk(lo, k, dim=0)[0][k - 1]
    uncentred = (hi >= thr[None, :]).sum(0).float().mean().item()
    centred = prefilter(mk, ms, qk, qe, k)[4].sum(0).float().mean().item()
    print(f'candidates per query: uncentred bound {uncentred:.0f}, centred {centred:.0f}')
    assert uncentred > 10 * centred
