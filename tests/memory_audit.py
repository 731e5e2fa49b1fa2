"""Audit helpers for the memory path (shared by the CPU host-logic tests, which run them on the
emulated ops, and by the -m gpu tests, which run them on the HIP kernels).

1. `ReadTap` / `OracleTap`: record every top-k memory read of a HIP `MemoryManager` and of the CPU
   oracle (bank, queries, selected token lists).
2. `TieFollowing` / `FollowMerge` / `Drift`: free-running clips -- the oracle adopts the HIP run's decisions at
   measured near-ties and the outputs are then compared under the north-star bound as written.
   `explain_flips`: compare the two runs' selections query by query.  A differing selection is
   *explained* only if the reference's own score gap between the tokens that were swapped is within
   the measured score noise between the two runs (their banks/queries differ in the last bits) --
   i.e. a near-tie that any last-bit change of the keys flips.  Everything else is a kernel bug.
3. `teacher_forced_memory`: drives a scenario with the CPU oracle and mirrors every memory call
   (match / add with its consolidation and eviction) into a HIP `MemoryManager` fed with the
   ORACLE'S inputs, asserting after every call: identical banks (keys bit-exact, i.e. identical
   prototype index lists and identical surviving token sets), prototype values / shrinkage and usage
   counters within 1e-5, identical top-k sets, read-outs within 1e-5 relative.
   Reference: memory_manager.py:91-276, kv_memory_store.py:35-185, memory_utils.py:48-76.
"""
import os
from typing import Dict, List

import torch

from deva.hip import ops
from oracle import deva_oracle as O
from workload import synth


def _sim(mk_rows: torch.Tensor, ms: torch.Tensor, qk: torch.Tensor, qe: torch.Tensor) -> torch.Tensor:
    """reference similarity (memory_utils.py:6-45) of a token-major bank [N,64] on the CPU"""
    return O.get_similarity(mk_rows.t().contiguous(), ms.reshape(1, -1), qk, qe)


def roundoff_floor(mk_rows: torch.Tensor, ms: torch.Tensor, qk_col: torch.Tensor, qe_col: torch.Tensor, tokens) -> float:
    """fp32 round-off scale of the reference's OWN score formula for these tokens and this query: the score is the small
    difference of three 64-term sums (memory_utils.py:29-43), so two correct fp32 evaluations of it (ATen's blocked
    sgemm, the kernels' natural-order FMA chain) differ by ulps of the TERMS, not of the result:
    2^-22 * ms/8 * (sum mk^2 qe + 2 |sum mk qk qe| + sum qe qk^2) -- two ulps of the largest intermediate.  On the
    1080p clip the terms are ~200x the score: ~1e-5, the size of the measured bank noise; a selection that differs by
    less than a few of these is a tie of the arithmetic, whatever the inputs' noise happened to be for that pair."""
    rows = mk_rows[list(tokens)].double()                      # [t, 64]
    qk, qe = qk_col.double().reshape(-1), qe_col.double().reshape(-1)
    a_sq = (rows * rows) @ qe
    two_ab = 2 * (rows @ (qk * qe))
    b_sq = (qe * qk * qk).sum()
    scale = ms.reshape(-1)[list(tokens)].double().abs() / 8.0
    return float(((a_sq + two_ab.abs() + b_sq) * scale).max()) * 2.0 ** -22


class ReadTap:
    """records the inputs and the selection of every `ops.affinity_topk` call"""

    def __init__(self):
        self.reads: List[Dict] = []
        self._orig = None

    def __enter__(self):
        self._orig = ops.affinity_topk
        tap = self

        def wrapped(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix=None, splits=None, **prep):
            idx, w = tap._orig(key_long, shr_long, n_long, key_work, shr_work, n_work, qk, qe, k, usage_fix, splits, **prep)
            rows = [key_long[:n_long]] if n_long else []
            shr = [shr_long[:n_long]] if n_long else []
            rows.append(key_work[:n_work])
            shr.append(shr_work[:n_work])
            tap.reads.append(dict(mk=torch.cat(rows, 0).cpu(), ms=torch.cat(shr, 0).cpu(), qk=qk.cpu(), qe=qe.cpu(),
                                  idx=idx.cpu().long(), w=w.cpu(), n_long=n_long, k=k))
            return idx, w

        ops.affinity_topk = wrapped
        return self

    def __exit__(self, *exc):
        ops.affinity_topk = self._orig


class OracleTap:
    """records the similarity matrix and inputs of every top-k read of the oracle"""

    def __init__(self):
        self.reads: List[Dict] = []
        self._orig = None
        self._last_inputs = None

    def __enter__(self):
        self._orig = (O.get_similarity, O.dense_affinity)
        tap = self

        def get_similarity(mk, ms, qk, qe):
            tap._last_inputs = (mk, ms, qk, qe)
            return tap._orig[0](mk, ms, qk, qe)

        def dense_affinity(sim, k):
            if k is not None:
                mk, ms, qk, qe = tap._last_inputs
                tap.reads.append(dict(mk=mk.t().contiguous(), ms=ms.reshape(-1).clone(), qk=qk.clone(), qe=qe.clone(),
                                      sim=sim.clone(), k=k))
            return tap._orig[1](sim, k)

        O.get_similarity, O.dense_affinity = get_similarity, dense_affinity
        return self

    def __exit__(self, *exc):
        O.get_similarity, O.dense_affinity = self._orig


def explain_flips(tag: str, hip_read: Dict, ref_read: Dict, slack: float = 4.0):
    """-> (number of queries whose selected token set differs, the largest unexplained excess).
    A flip is explained when, for every token selected by one side only, the reference's distance of
    that token's score to its k-th/(k+1)-th boundary is <= slack x the measured score noise of the
    query (max |score from the HIP run's bank/queries - score from the reference's| over the tokens
    involved, both evaluated with the reference formula on the CPU)."""
    sim_ref = ref_read['sim']
    k = ref_read['k']
    n, hw = sim_ref.shape
    assert hip_read['mk'].shape[0] == n, f'{tag}: bank sizes differ ({hip_read["mk"].shape[0]} vs {n})'
    vals, ridx = torch.topk(sim_ref, k=min(k + 1, n), dim=0)
    ref_sets = ridx[:k].t()
    same = (torch.sort(hip_read['idx'], 1)[0] == torch.sort(ref_sets, 1)[0]).all(1)
    flipped = torch.nonzero(~same).flatten().tolist()
    worst_excess, lines = 0.0, []
    if flipped:
        sim_hip = _sim(hip_read['mk'], hip_read['ms'], hip_read['qk'], hip_read['qe'])
    for q in flipped:
        a, b = set(hip_read['idx'][q].tolist()), set(ref_sets[q].tolist())
        swapped = sorted(a ^ b)
        boundary = 0.5 * (vals[k - 1, q] + vals[k, q]).item() if n > k else vals[k - 1, q].item()
        gap = max(abs(sim_ref[t, q].item() - boundary) for t in swapped)
        noise = max(abs(sim_hip[t, q].item() - sim_ref[t, q].item()) for t in swapped)
        noise = max(noise, 1e-6 * abs(boundary),  # fp32 evaluation noise of the score itself ...
                    roundoff_floor(hip_read['mk'], hip_read['ms'], hip_read['qk'][:, q], hip_read['qe'][:, q], swapped))
        excess = gap / noise
        worst_excess = max(worst_excess, excess)
        lines.append(f'{tag}: query {q}: {len(swapped) // 2} token(s) swapped, boundary score {boundary:.6g}, '
                     f'reference gap {gap:.3e}, measured score noise {noise:.3e} (ratio {excess:.2f})')
    for line in lines[:8]:
        print(line)
    return len(flipped), (worst_excess if flipped else 0.0), slack


class TieFollowing:
    """While active, every top-k read of the CPU oracle is compared with the HIP run's read of the same bucket
    (`hip_reads`, in order, from a `ReadTap`) and ADOPTS the HIP selection for a query wherever the two sets differ
    at a measured near-tie -- for every token selected by one side only, the reference's distance of that token's
    score to its k-th/(k+1)-th boundary is within `slack` x the measured score noise between the two runs (both
    evaluated with the reference formula on the CPU, see explain_flips).  The adopted weights are the reference's
    own softmax over the adopted tokens' reference scores.  A differing selection that is NOT a near-tie is
    recorded in `unexplained` (the tests fail on it).

    The oracle run that results is "the reference, given the same decisions at fp32 near-ties": the HIP outputs are
    then held to the north-star bound against it WITHOUT any allowance (1e-3 max-abs, argmax-identical above a
    2e-3 margin) -- a kernel error that is not a last-bit tie-break has nowhere to hide, and a tie-break cannot
    start an unbounded "explained" exceedance."""

    def __init__(self, tag: str, hip_reads: List[Dict], slack: float = 4.0):
        self.tag, self.queue, self.slack = tag, list(hip_reads), slack
        self.adopted, self.reads, self.unexplained, self.lines = 0, 0, [], []
        self._orig = None

    def __enter__(self):
        self._orig = O.topk_softmax
        self._orig_sim = O.get_similarity
        O.topk_softmax = self._topk_softmax

        def get_similarity(mk, ms, qk, qe):
            self._last_inputs = (mk, ms, qk, qe)
            return self._orig_sim(mk, ms, qk, qe)

        O.get_similarity = get_similarity
        return self

    def __exit__(self, *exc):
        O.topk_softmax = self._orig
        O.get_similarity = self._orig_sim

    def _topk_softmax(self, sim, k):
        idx, w = self._orig(sim, k)
        assert self.queue, f'{self.tag}: the oracle reads memory more often than the HIP run did'
        hr = self.queue.pop(0)
        self.reads += 1
        n, hw = sim.shape
        assert hr['mk'].shape[0] == n and hr['idx'].shape == (hw, k), \
            f'{self.tag}: bank / query sizes differ ({hr["mk"].shape[0]} x {tuple(hr["idx"].shape)} vs {n} x {hw})'
        same = (torch.sort(hr['idx'], 1)[0] == torch.sort(idx.t(), 1)[0]).all(1)
        if os.environ.get('DEVA_AUDIT_VERBOSE'):
            mk_o, ms_o, qk_o, qe_o = self._last_inputs
            print(f'{self.tag} read {self.reads}: N={n} bank key diff {(hr["mk"] - mk_o.t()).abs().max().item():.2e} '
                  f'shrinkage diff {(hr["ms"] - ms_o.reshape(-1)).abs().max().item():.2e} query key diff '
                  f'{(hr["qk"] - qk_o).abs().max().item():.2e} differing sets {int((~same).sum())}')
        for q in torch.nonzero(~same).flatten().tolist():
            col = sim[:, q]
            vals = torch.topk(col, k=min(k + 1, n))[0]
            boundary = 0.5 * (vals[k - 1] + vals[k]).item() if n > k else vals[k - 1].item()
            hip_set, ref_set = hr['idx'][q], idx[:, q]
            swapped = sorted(set(hip_set.tolist()) ^ set(ref_set.tolist()))
            col_hip = _sim(hr['mk'], hr['ms'], hr['qk'][:, q:q + 1], hr['qe'][:, q:q + 1])[:, 0]
            gap = max(abs(col[t].item() - boundary) for t in swapped)
            noise = max(max(abs(col_hip[t].item() - col[t].item()) for t in swapped), 1e-6 * abs(boundary),
                        roundoff_floor(hr['mk'], hr['ms'], hr['qk'][:, q], hr['qe'][:, q], swapped))
            excess = gap / noise
            line = (f'{self.tag} read {self.reads} query {q}: {len(swapped) // 2} token(s) swapped, boundary score '
                    f'{boundary:.6g}, reference gap {gap:.3e}, measured score noise {noise:.3e} (ratio {excess:.2f})')
            if excess > self.slack:
                self.unexplained.append(line)
                continue
            if len(self.lines) < 6:
                self.lines.append(line)
            # adopt: the HIP tokens in the reference's order (score desc, index asc), weighted by the reference's scores
            s = col[hip_set]
            order = sorted(range(k), key=lambda j: (-s[j].item(), int(hip_set[j])))
            new = hip_set[order]
            e = col[new].exp()
            idx[:, q] = new
            w[:, q] = e / e.sum()
            self.adopted += 1
        return idx, w

    def check(self):
        for line in self.lines:
            print(line)
        self.lines = []
        assert not self.unexplained, (f'{self.tag}: top-k selections differ from the reference beyond a near-tie:\n'
                                      + '\n'.join(self.unexplained[:8]))


NORTH_STAR = 1e-3  # BASELINE.json north_star: max-abs bound on the soft outputs


def margin_aware_mismatch(got: torch.Tensor, ref: torch.Tensor, floor: float = None) -> int:
    """argmax mismatches at pixels whose REFERENCE top-1/top-2 margin exceeds 2 x max(floor, 1e-3), where
    `floor` is the reference's own noise on this frame (reference vs the reference under a 1e-6 relative
    input perturbation) -- NOT the error under test (with the observed error as the margin
    the check cannot fail).  Without a floor run the margin is the fixed 2 x 1e-3 of the north-star bound."""
    return argmax_flips(got, ref, floor)[1]


def argmax_flips(got: torch.Tensor, ref: torch.Tensor, floor: float = None, margin: float = None):
    """-> (raw argmax flips, flips at pixels whose reference margin exceeds `margin`)
    margin defaults to 2 x max(floor, 1e-3)"""
    top2 = ref.topk(2, dim=0)[0]
    if margin is None:
        margin = 2 * max(floor or 0.0, NORTH_STAR)
    flipped = got.argmax(0) != ref.argmax(0)
    return int(flipped.sum()), int((flipped & ((top2[0] - top2[1]) > margin)).sum())


class FollowMerge:
    """The other discrete decision on the path: `incorporate_detection` merges the ARGMAX of the forward pass
    (inference_core.py:166).  While active, the oracle's merge adopts the HIP run's forward mask wherever the two
    differ at pixels whose top-1/top-2 probability margin in the oracle's own forward pass is <= 2 x 1e-3 (a
    near-tie under the north-star bound); a differing pixel with a larger margin is recorded in `unexplained`."""

    def __init__(self, tag: str, orc_forward_prob, hip_forward: torch.Tensor):
        self.tag, self.prob, self.hip_forward = tag, orc_forward_prob, hip_forward
        self.adopted, self.unexplained = 0, []
        self._orig = None

    def __enter__(self):
        self._orig = O.merge_detection
        follow = self

        def merge_detection(forward, detected, table, segments, history, **kw):
            if follow.hip_forward is not None:
                hip_fwd = follow.hip_forward.to(forward.dtype)
                assert hip_fwd.shape == forward.shape, (hip_fwd.shape, forward.shape)
                differ = forward != hip_fwd
                if bool(differ.any()):
                    top2 = follow.prob().topk(2, dim=0)[0]
                    margin = (top2[0] - top2[1])[differ]
                    if float(margin.max()) > 2 * NORTH_STAR:
                        follow.unexplained.append(f'{follow.tag}: forward argmax differs at a pixel with reference margin '
                                                  f'{float(margin.max()):.2e}')
                    else:
                        follow.adopted += int(differ.sum())
                        forward = hip_fwd
            return follow._orig(forward, detected, table, segments, history, **kw)

        O.merge_detection = merge_detection
        return self

    def __exit__(self, *exc):
        O.merge_detection = self._orig

    def check(self):
        assert not self.unexplained, '\n'.join(self.unexplained[:4])


class Drift:
    """Free-running clips.  `ref` is the TIE-FOLLOWING reference (the CPU oracle run under `TieFollowing` /
    `FollowMerge`: identical arithmetic, the HIP run's decisions adopted at measured fp32 near-ties only), and the
    HIP outputs are held to the north-star numbers against it as written, on every frame:
        max-abs <= 1e-3 on the soft outputs, argmax-identical at every pixel whose reference margin exceeds 2e-3.
    Reported beside it (not asserted, except where nothing was adopted yet): the error against the CLEAN reference
    (the reference's golden outputs / the plain oracle) and the reference's own drift under a 1e-6 relative input
    perturbation -- the noise floor a tie-break costs the reference itself."""

    def __init__(self, tag, stride=1, floor_bound=False):
        """floor_bound=True (the peaky recipe only, whose keys are 15x larger: the score noise between two fp32
        implementations grows with the gain squared and reaches the soft outputs through the softmax weights even
        without a single differing decision): the bound is max(1e-3, 10 x the reference's own drift under a 1e-6
        input perturbation), measured in the same test (`ref_perturbed` against the tie-following run)"""
        self.tag, self.stride, self.floor_bound = tag, stride, floor_bound
        self.rows = []

    @staticmethod
    def _stats(a, b):
        d = (a - b).abs()
        return d.max().item(), (d > 1e-3).float().mean().item()

    def add(self, got, ref, clean=None, ref_perturbed=None, frame=None, adopted_so_far=0):
        frame = len(self.rows) if frame is None else frame
        err, frac = self._stats(got, ref)
        raw, decisive = argmax_flips(got, ref, margin=2 * NORTH_STAR)
        row = dict(frame=frame, err=err, raw=raw, decisive=decisive, adopted=adopted_so_far)
        msg = (f'{self.tag} frame {frame}: HIP vs tie-following ref max-abs {err:.2e} argmax flips raw {raw}, at margin > '
               f'2e-3: {decisive}; decisions adopted so far {adopted_so_far}')
        if clean is not None:
            row['clean'], _ = self._stats(got, clean)
            row['clean_raw'] = argmax_flips(got, clean)[0]
            msg += f' | vs clean ref max-abs {row["clean"]:.2e} flips raw {row["clean_raw"]}'
        if ref_perturbed is not None:
            base = clean if clean is not None else ref  # (the tie-following run is the clean one up to the adopted ties)
            row['floor'], _ = self._stats(ref_perturbed, base)
            row['floor_raw'] = argmax_flips(ref_perturbed, base)[0]
            msg += f' | ref vs ref(1e-6 input noise) max-abs {row["floor"]:.2e} flips raw {row["floor_raw"]}'
        print(msg)
        self.rows.append(row)

    def finish(self):
        worst = max(r['err'] for r in self.rows)
        report = dict(max_abs_vs_tie_following=worst, decisive_flips=sum(r['decisive'] for r in self.rows),
                      raw_flips=sum(r['raw'] for r in self.rows), adopted=max(r['adopted'] for r in self.rows))
        if any('clean' in r for r in self.rows):
            report['max_abs_vs_clean'] = max(r.get('clean', 0.0) for r in self.rows)
            report['raw_flips_vs_clean'] = sum(r.get('clean_raw', 0) for r in self.rows)
        if any('floor' in r for r in self.rows):
            report['reference_self_drift'] = max(r.get('floor', 0.0) for r in self.rows)
            report['reference_self_flips'] = sum(r.get('floor_raw', 0) for r in self.rows)
        bound = NORTH_STAR
        if self.floor_bound:
            assert 'reference_self_drift' in report, 'floor_bound needs the perturbed-oracle run'
            bound = max(NORTH_STAR, 10 * report['reference_self_drift'])
        report['bound'] = bound
        print(f'{self.tag}: ' + ', '.join(f'{k} {v:.3g}' for k, v in report.items()))
        for r in self.rows:
            assert r['err'] <= bound, (f'{self.tag} frame {r["frame"]}: max-abs {r["err"]:.2e} vs the tie-following '
                                       f'reference exceeds {bound:.1e}')
            assert r['decisive'] == 0, (f'{self.tag} frame {r["frame"]}: {r["decisive"]} argmax flips at pixels whose '
                                        'reference margin exceeds 2e-3')
            if 'clean' in r and r['adopted'] == 0:
                assert r['clean'] <= NORTH_STAR, (f'{self.tag} frame {r["frame"]}: {r["clean"]:.2e} vs the clean reference '
                                                  'before any decision was adopted')
        return report


def paired_steps(tag, n_frames, hip_call, following_call, clean_call=None, noisy_call=None, floor_bound=False):
    """Frame-by-frame driver of a free-running comparison: hip_call(t) runs the HIP core (its memory reads are
    tapped), following_call(t) the CPU oracle under `TieFollowing` of exactly those reads; clean_call / noisy_call
    (optional) the plain oracle and the oracle on 1e-6-perturbed inputs (or stored golden outputs).  All return the
    frame's probabilities on the CPU.  -> Drift report (asserted)."""
    drift, adopted = Drift(tag, floor_bound=floor_bound), 0
    for t in range(n_frames):
        with ReadTap() as tap:
            a = hip_call(t)
        with TieFollowing(f'{tag} frame {t}', tap.reads) as tf:
            b = following_call(t)
        tf.check()
        assert not tf.queue, f'{tag} frame {t}: the HIP run read memory {len(tf.queue)} more time(s) than the oracle'
        adopted += tf.adopted
        drift.add(a, b, None if clean_call is None else clean_call(t), None if noisy_call is None else noisy_call(t),
                  frame=t, adopted_so_far=adopted)
    return drift.finish()


def _cmp(name, got, ref, tol, worst):
    got = got.detach().float().cpu()
    e = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item()) if ref.numel() else 0.0
    worst[name] = max(worst.get(name, 0.0), e)
    assert e <= tol, (name, e)


def _compare_banks(tag, hmem, omem, worst):
    """keys / selections bit-exact (they are copies: equality <=> same tokens in the same order),
    computed quantities (prototype values & shrinkage, usage counters) to 1e-5"""
    stores = [('work', hmem.work_mem, omem.work)]
    if omem.long_term and omem.long.buckets:
        stores.append(('long', hmem.long_mem, omem.long))
    for sname, hs, os_ in stores:
        assert sorted(hs.buckets) == sorted(os_.buckets), (tag, sname, hs.buckets, os_.buckets)
        hk, hshr = hs.key, hs.shrinkage
        for b in os_.buckets:
            assert hs.size(b) == os_.size(b), (tag, sname, b, hs.size(b), os_.size(b))
            assert torch.equal(hk[b].cpu(), os_.k[b]), f'{tag}: {sname} bucket {b}: key bank differs (different tokens kept / chosen)'
            _cmp(f'{sname}.shrinkage', hshr[b], os_.s[b], 1e-5, worst)
            if os_.keep_selection:
                assert torch.equal(hs.selection[b].cpu(), os_.e[b]), (tag, sname, b, 'selection')
            if os_.keep_usage:
                use, life = hs.usage_arenas(b)
                _cmp(f'{sname}.use_cnt', use[:hs.size(b)], os_.use[b], 1e-5, worst)
                _cmp(f'{sname}.life_cnt', life[:hs.size(b)], os_.life[b], 1e-6, worst)
        hv = hs.value
        for o, v in os_.v.items():
            _cmp(f'{sname}.value', hv[o], v, 1e-5, worst)


def teacher_forced_memory(P, sc: Dict, device, frames: int = None) -> Dict[str, float]:
    """see the module docstring (3).  Returns the worst relative error per compared quantity plus
    event counters."""
    from deva.inference.memory_manager import MemoryManager
    cfg = synth.base_config(**sc['cfg'])
    sc = dict(sc, frames=frames or sc['frames'])
    hmem = MemoryManager(cfg)
    worst: Dict[str, float] = {}
    events = dict(reads=0, adds=0, consolidations=0, evictions=0, tie_swapped_queries=0)
    state = dict(core=None, frame=-1)

    def install(core):
        omem = core.memory
        match0, add0 = omem.match, omem.add

        def match(key, selection):
            tag = f'frame {core.curr_ti}'
            with OracleTap() as otap:
                ro_o = match0(key, selection)
            with ReadTap() as htap:
                ro_h = hmem.match_memory(key.to(device), selection.to(device))
            assert len(otap.reads) == len(htap.reads)
            for bi, (hr, orr) in enumerate(zip(htap.reads, otap.reads)):
                assert torch.equal(hr['mk'], orr['mk']) and torch.equal(hr['qk'], orr['qk'])
                n_flip, excess, slack = explain_flips(f'{tag} bucket#{bi}', hr, orr)
                # identical inputs: a differing set is only acceptable at an fp32 tie
                assert excess <= slack, f'{tag}: top-k selection differs from the reference beyond a tie'
                events['tie_swapped_queries'] += n_flip
            for o, r in ro_o.items():
                _cmp('readout', ro_h[o], r, 1e-5 if events['tie_swapped_queries'] == 0 else 1e-2, worst)
            events['reads'] += 1
            _compare_banks(tag + ' after read', hmem, omem, worst)
            return ro_o

        def add(key, shrinkage, value, objects, selection):
            tag = f'frame {core.curr_ti} add'
            long_before = {b: omem.long.size(b) for b in omem.long.buckets} if omem.long_term else {}
            add0(key, shrinkage, value, objects, selection)
            hmem.add_memory(key.to(device), shrinkage.to(device), value.to(device), list(objects),
                            selection=selection.to(device))
            events['adds'] += 1
            if omem.long_term:
                for b in omem.long.buckets:
                    grown = omem.long.size(b) - long_before.get(b, 0)
                    if b not in long_before or grown != 0:
                        events['consolidations'] += 1
                        if grown < cfg['num_prototypes']:
                            events['evictions'] += 1
            _compare_banks(tag, hmem, omem, worst)

        omem.match, omem.add = match, add
        return core

    import scenarios
    scenarios.run_scenario(lambda c: install(O.OracleCore(P, c)), sc)
    worst.update({k: float(v) for k, v in events.items()})
    return worst
