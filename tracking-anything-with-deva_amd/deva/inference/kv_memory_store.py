"""`KeyValueMemoryStore` with the reference's interface (deva/inference/kv_memory_store.py:5-276)
on pre-allocated TOKEN-MAJOR device arenas.

The reference stores channel-major [C, N] tensors and re-allocates every one of them with
`torch.cat` on each append.  Here each bucket owns growable arenas whose row n is memory token n
(key [cap,CK], shrinkage [cap], selection [cap,CK], usage counters [cap]; per object value
[cap,CV]); appending a frame is one transpose-copy kernel per array, and the memory-read kernels
consume the arenas in place (a row is one 256-B key / 2-KiB value: the unit the affinity and
readout kernels gather).  The reference's channel-major tensors remain available through the
`key` / `value` / `shrinkage` / `selection` properties (exported on demand, not used on the hot
path).
"""
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from deva.hip import ops

_MIN_ROWS = 2048


class _Arena:
    """growable [cap, channels] fp32 buffer; only rows [0, owner.size) are meaningful"""

    def __init__(self, channels: int, device, fill: float = 0.0):
        self.channels, self.device, self.fill = channels, device, fill
        self.buf: Optional[torch.Tensor] = None

    def ensure(self, rows: int, live_rows: int) -> torch.Tensor:
        if self.buf is None or self.buf.shape[0] < rows:
            cap = max(_MIN_ROWS, rows, 0 if self.buf is None else 2 * self.buf.shape[0])
            shape = (cap, self.channels) if self.channels > 1 else (cap,)
            new = torch.empty(shape, dtype=torch.float32, device=self.device)
            if self.buf is not None and live_rows > 0:
                new[:live_rows].copy_(self.buf[:live_rows])  # device memcpy, amortised O(1)
            self.buf = new
        return self.buf


class _Bucket:
    def __init__(self, ck: int, device, with_selection: bool, with_usage: bool):
        self.n = 0
        # value-sharded storage (shard_values): local row of every token in this rank's value arenas (-1 = the
        # row lives on another rank) and the number of rows this rank holds
        self.lrow: Optional[torch.Tensor] = None
        self.n_local = 0
        self.k = _Arena(ck, device)
        self.s = _Arena(1, device)
        self.e = _Arena(ck, device) if with_selection else None
        self.use = _Arena(1, device) if with_usage else None
        self.life = _Arena(1, device) if with_usage else None


class KeyValueMemoryStore:
    """
    Key/value storage for the working and the long-term memory.
    Objects that first appear in the same frame share a bucket: one key bank per bucket, one
    value bank per object (kv_memory_store.py:10-16).
    """

    def __init__(self, save_selection: bool = False, save_usage: bool = False):
        self.save_selection = save_selection
        self.save_usage = save_usage
        self.global_bucket_id = 0  # never decreases
        self.buckets: Dict[int, List[int]] = {}
        self._b: Dict[int, _Bucket] = {}
        self._v: Dict[int, _Arena] = {}
        self._obj_bucket: Dict[int, int] = {}
        self._vshard: Optional[Tuple[int, int]] = None  # (rank, world) once shard_values() was called
        # version of every bucket's key / shrinkage rows: bumped by every call that adds, drops or moves rows (add, the
        # sieve / eviction / range removals through _rebuild); what the cached pre-filter operands of the memory read
        # are keyed on (memory_manager.py: _bank_prep) -- the usage counters change on every read and are not part of it
        self._ver: Dict[int, int] = {}
        self._tick = 0

    # ------------------------------------------------------------------ value-sharded storage (several GPUs, one clip)
    def shard_values(self, rank: int, world: int) -> None:
        """From now on this store keeps only ITS share of the VALUE rows (the bulk of a memory token: CV = 512 floats
        per object against 64 + 64 + 3 for key / selection / shrinkage / usage, which stay replicated because every
        rank scores and ranks them).  Of every batch of tokens appended together (a memory frame, a set of
        prototypes) rank r owns the contiguous block [n*r/world, n*(r+1)/world); ownership travels with the token
        through sieves and evictions (`row_map`).  Must be called on an empty store."""
        assert not self._b and not self._v, 'shard_values: the store already holds memory'
        assert 0 <= rank < world
        self._vshard = (rank, world)

    def row_map(self, bucket_id: int) -> Optional[torch.Tensor]:
        """int32 [>= size]: local value row of every token of the bucket, -1 = another rank's; None if not sharded"""
        return self._b[bucket_id].lrow if self._vshard is not None else None

    def local_size(self, bucket_id: int) -> int:
        bk = self._b[bucket_id]
        return bk.n_local if self._vshard is not None else bk.n

    def local_rows(self, bucket_id: int, lo: int, hi: int) -> Tuple[int, int, Optional[torch.Tensor]]:
        """tokens [lo, hi) of a bucket -> (first local row, one past the last local row, int64 offsets within
        [lo, hi) of the tokens this rank owns); local rows keep the global order, so the range is contiguous.
        Unsharded: (lo, hi, None).  One host synchronisation (memory events only)."""
        if self._vshard is None:
            return lo, hi, None
        lrow = self._b[bucket_id].lrow
        own = torch.nonzero(lrow[lo:hi] >= 0).flatten()
        first = int((lrow[:lo] >= 0).sum().item())
        return first, first + own.numel(), own

    # ------------------------------------------------------------------ internal accessors
    def version(self, bucket_id: int) -> int:
        """changes whenever the rows of the bucket's key / shrinkage arenas change (0: no such bucket)"""
        return self._ver.get(bucket_id, 0)

    def _bump(self, bucket_id: int) -> None:
        self._tick += 1
        self._ver[bucket_id] = self._tick

    def bucket_of(self, obj: int) -> int:
        return self._obj_bucket[obj]

    def key_arena(self, bucket_id: int) -> torch.Tensor:
        return self._b[bucket_id].k.buf

    def shrinkage_arena(self, bucket_id: int) -> torch.Tensor:
        return self._b[bucket_id].s.buf

    def selection_arena(self, bucket_id: int) -> torch.Tensor:
        return self._b[bucket_id].e.buf

    def usage_arenas(self, bucket_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        b = self._b[bucket_id]
        return b.use.buf, b.life.buf

    def value_arena(self, obj: int) -> torch.Tensor:
        return self._v[obj].buf

    # ------------------------------------------------------------------ add
    def add(self, key: torch.Tensor, values: Dict[int, torch.Tensor], shrinkage: torch.Tensor,
            selection: Optional[torch.Tensor], supposed_bucket_id: int = -1, *,
            token_major: bool = False) -> None:
        """
        key: C*N   values: {obj: C*N}   shrinkage: 1*N   selection: C*N   (kv_memory_store.py:35-66)
        supposed_bucket_id: put everything into this bucket (keeps working/long-term ids in sync).
        token_major=True (internal): the tensors are already rows, i.e. N*C / N.
        """
        assert key.dim() == 2 and shrinkage.dim() in (1, 2)
        assert not self.save_selection or (selection is not None and selection.dim() == 2)
        n_new = key.shape[0] if token_major else key.shape[1]
        device = key.device

        if supposed_bucket_id >= 0:
            touched = [supposed_bucket_id]
            exists = supposed_bucket_id in self.buckets
            for obj in values:
                assert (obj in self._v) == exists
                assert not exists or obj in self.buckets[supposed_bucket_id]
            self.buckets[supposed_bucket_id] = list(values.keys())
        else:
            touched, fresh = [], None
            for obj in values:
                if obj in self._v:
                    b = self._obj_bucket[obj]
                else:
                    if fresh is None:
                        fresh = self.global_bucket_id
                        self.global_bucket_id += 1
                        self.buckets[fresh] = []
                    self.buckets[fresh].append(obj)
                    b = fresh
                if b not in touched:
                    touched.append(b)

        for b in touched:
            self._bump(b)

        def put(arena: _Arena, src: torch.Tensor, n_old: int):
            dst = arena.ensure(n_old + n_new, n_old)
            if token_major:
                ops.bank_gather_rows(src.contiguous(), None, dst[n_old:n_old + n_new], n_new)
            elif arena.channels == 1:
                ops.bank_gather_rows(src.reshape(-1).contiguous(), None, dst[n_old:n_old + n_new], n_new)
            else:
                ops.bank_append(src.contiguous(), dst, n_old)

        if self._vshard is not None:
            rank, world = self._vshard
            i0, i1 = n_new * rank // world, n_new * (rank + 1) // world  # this rank's block of the batch
        for obj, val in values.items():
            if supposed_bucket_id >= 0:
                b = supposed_bucket_id
            elif obj in self._obj_bucket:
                b = self._obj_bucket[obj]
            else:
                b = fresh
            if obj not in self._v:
                self._v[obj] = _Arena(val.shape[1] if token_major else val.shape[0], device)
                self._obj_bucket[obj] = b
            if self._vshard is None:
                put(self._v[obj], val, self._b[b].n if b in self._b else 0)
            elif i1 > i0:
                n_loc = self._b[b].n_local if b in self._b else 0
                part = val[i0:i1] if token_major else val[:, i0:i1]
                dst = self._v[obj].ensure(n_loc + (i1 - i0), n_loc)
                if token_major:
                    ops.bank_gather_rows(part.contiguous(), None, dst[n_loc:n_loc + (i1 - i0)], i1 - i0)
                else:
                    ops.bank_append(part.contiguous(), dst, n_loc)
            else:
                self._v[obj].ensure(1, 0)

        ck = key.shape[1] if token_major else key.shape[0]
        for b in touched:
            if b not in self._b:
                self._b[b] = _Bucket(ck, device, self.save_selection, self.save_usage)
            bk = self._b[b]
            put(bk.k, key, bk.n)
            put(bk.s, shrinkage, bk.n)
            if self.save_selection:
                put(bk.e, selection, bk.n)
            if self.save_usage:
                use = bk.use.ensure(bk.n + n_new, bk.n)
                life = bk.life.ensure(bk.n + n_new, bk.n)
                ops.usage_init(use[bk.n:bk.n + n_new], life[bk.n:bk.n + n_new])  # 0 and 1e-7 (kv_memory_store.py:93-95)
            if self._vshard is not None:
                if bk.lrow is None or bk.lrow.numel() < bk.n + n_new:
                    grown = torch.full((max(_MIN_ROWS, 2 * (bk.n + n_new)),), -1, dtype=torch.int32, device=device)
                    if bk.lrow is not None:
                        grown[:bk.n] = bk.lrow[:bk.n]
                    bk.lrow = grown
                bk.lrow[bk.n:bk.n + n_new] = -1
                if i1 > i0:
                    bk.lrow[bk.n + i0:bk.n + i1] = torch.arange(bk.n_local, bk.n_local + (i1 - i0), dtype=torch.int32,
                                                                device=device)
                bk.n_local += i1 - i0
            bk.n += n_new

    # ------------------------------------------------------------------ usage
    def update_bucket_usage(self, bucket_id: int, usage: torch.Tensor) -> None:
        """kv_memory_store.py:118-125 (dense fp32 usage; API compatibility -- the frame path feeds
        the fixed-point counters of the affinity kernel through `apply_usage_fix`)."""
        if not self.save_usage:
            return
        bk = self._b[bucket_id]
        bk.use.buf[:bk.n] += usage.reshape(-1)
        bk.life.buf[:bk.n] += 1

    def apply_usage_fix(self, bucket_id: int, usage_fix: torch.Tensor, offset: int) -> None:
        """use += usage (2^-40 fixed point, rows offset..offset+size of usage_fix), life += 1, and
        clear the consumed counters.  Without usage counting only the clearing happens."""
        bk = self._b[bucket_id]
        if self.save_usage:
            ops.usage_update(usage_fix, offset, bk.use.buf, bk.life.buf, bk.n)
        else:
            ops.usage_update(usage_fix, offset, None, None, bk.n)

    def get_usage(self, bucket_id: int) -> torch.Tensor:
        if not self.save_usage:
            raise RuntimeError('I did not count usage!')
        bk = self._b[bucket_id]
        return bk.use.buf[:bk.n] / bk.life.buf[:bk.n]

    # ------------------------------------------------------------------ compaction
    def _rebuild(self, bucket_id: int, parts: Sequence[Union[Tuple[int, int], Tuple[torch.Tensor, int]]]):
        """keep `parts` (each a (lo, hi) row range or an (int32 row-index tensor, count)) in order"""
        self._bump(bucket_id)
        bk = self._b[bucket_id]
        total = sum((p[1] - p[0]) if isinstance(p[0], int) else p[1] for p in parts)
        arenas = [bk.k, bk.s] + ([bk.e] if bk.e else []) + ([bk.use, bk.life] if bk.use else [])
        if self._vshard is None:
            arenas += [self._v[o] for o in self.buckets[bucket_id]]
        else:
            # the kept tokens' local rows, in order; the survivors are renumbered 0 .. n_local-1
            dev = bk.lrow.device
            keep = torch.cat([torch.arange(p[0], p[1], device=dev) if isinstance(p[0], int) else p[0][:p[1]].long()
                              for p in parts]) if parts else torch.zeros(0, dtype=torch.long, device=dev)
            old_l = bk.lrow[keep]
            mine = old_l >= 0
            local_idx = old_l[mine].contiguous()
            n_local = int(local_idx.numel())
            new_lrow = torch.full_like(bk.lrow, -1)
            new_lrow[:total] = torch.where(mine, (torch.cumsum(mine, 0) - 1).int(), torch.full_like(old_l, -1))
            for o in self.buckets[bucket_id]:
                a = self._v[o]
                old = a.buf
                a.buf = None
                new = a.ensure(max(n_local, old.shape[0]), 0)
                if n_local:
                    ops.bank_gather_rows(old, local_idx, new[:n_local], n_local)
            bk.lrow, bk.n_local = new_lrow, n_local
        for a in arenas:
            old = a.buf
            a.buf = None
            new = a.ensure(max(total, old.shape[0]), 0)
            at = 0
            for p in parts:
                if isinstance(p[0], int):
                    cnt = p[1] - p[0]
                    if cnt > 0:
                        ops.bank_gather_rows(old[p[0]:p[1]], None, new[at:at + cnt], cnt)
                else:
                    cnt = p[1]
                    if cnt > 0:
                        ops.bank_gather_rows(old, p[0], new[at:at + cnt], cnt)
                at += cnt
        bk.n = total

    def sieve_by_range(self, bucket_id: int, start: int, end: int, min_size: int) -> None:
        """keep only the tokens outside [start, end) (end <= 0 counts from the back; 0 = to the end);
        buckets with <= min_size tokens are left alone (kv_memory_store.py:127-159)"""
        n = self.size(bucket_id)
        if n <= min_size:
            return
        stop = n if end == 0 else n + end
        assert end <= 0 and 0 <= start <= stop <= n
        self._rebuild(bucket_id, [(0, start), (stop, n)])

    def remove_old_memory(self, bucket_id: int, start_idx: int, max_len: int) -> None:
        self.sieve_by_range(bucket_id, start_idx, -max_len + start_idx, max_len)

    def remove_obsolete_features(self, bucket_id: int, max_size: int) -> None:
        """drop the (size - max_size) least-used tokens; every token whose normalised usage is <=
        the threshold goes, ties included (kv_memory_store.py:164-185)"""
        if not self.save_usage:
            raise RuntimeError('I did not count usage!')
        bk = self._b[bucket_id]
        n_remove = bk.n - max_size
        if n_remove <= 0:
            # torch.topk(k=0)[0][-1] in the reference (kv_memory_store.py:170-174)
            raise IndexError('index -1 is out of bounds for dimension 0 with size 0')
        rank_asc, usage = ops.rank(bk.use.buf, bk.n, False, life=bk.life.buf)
        idx, count = ops.evict_select(usage, rank_asc, n_remove)
        kept = int(count.item())  # host sync: the new size is host-side state (once per consolidation)
        self._rebuild(bucket_id, [(idx, kept)])

    # ------------------------------------------------------------------ reference-layout views
    def get_all_sliced(self, bucket_id: int, start: int, end: int):
        """k, sk, ek, values, normalised usage sliced [start:end) in the reference's channel-major
        layout (kv_memory_store.py:195-214; end == 0 means "to the end")"""
        n = self.size(bucket_id)
        stop = n if end == 0 else (n + end if end < 0 else end)
        cnt = stop - start
        bk = self._b[bucket_id]
        if self._vshard is not None:
            raise NotImplementedError('get_all_sliced: the value rows of a sharded store live on several ranks')
        k = ops.bank_export(bk.k.buf[start:stop], cnt)
        sk = bk.s.buf[start:stop].clone().unsqueeze(0)
        ek = ops.bank_export(bk.e.buf[start:stop], cnt) if self.save_selection else None
        value = {o: ops.bank_export(self._v[o].buf[start:stop], cnt) for o in self.buckets[bucket_id]}
        usage = self.get_usage(bucket_id)[start:stop] if self.save_usage else None
        return k, sk, ek, value, usage

    def purge_except(self, obj_keep_idx: List[int]) -> None:
        # kv_memory_store.py:216-239
        keep = set(obj_keep_idx)
        for b in list(self.buckets):
            self.buckets[b] = [o for o in self.buckets[b] if o in keep]
            if not self.buckets[b]:
                del self.buckets[b]
                self._b.pop(b, None)
                self._ver.pop(b, None)
        self._v = {o: a for o, a in self._v.items() if o in keep}
        self._obj_bucket = {o: b for o, b in self._obj_bucket.items() if o in keep}

    def get_v_size(self, obj_id: int) -> int:
        return self._b[self._obj_bucket[obj_id]].n

    def size(self, bucket_id: int) -> int:
        return self._b[bucket_id].n if bucket_id in self._b else 0

    def engaged(self, bucket_id: Optional[int] = None) -> bool:
        return len(self.buckets) > 0 if bucket_id is None else bucket_id in self.buckets

    @property
    def num_objects(self) -> int:
        return len(self._v)

    @property
    def key(self) -> Dict[int, torch.Tensor]:
        return {b: ops.bank_export(bk.k.buf, bk.n) for b, bk in self._b.items()}

    @property
    def value(self) -> Dict[int, torch.Tensor]:
        """reference layout {obj: CV x N}; a value-sharded store returns THIS RANK'S rows (CV x local_size)"""
        return {o: ops.bank_export(a.buf, self.local_size(self._obj_bucket[o])) for o, a in self._v.items()}

    @property
    def shrinkage(self) -> Dict[int, torch.Tensor]:
        return {b: bk.s.buf[:bk.n].clone().unsqueeze(0) for b, bk in self._b.items()}

    @property
    def selection(self) -> Dict[int, torch.Tensor]:
        return {b: ops.bank_export(bk.e.buf, bk.n) for b, bk in self._b.items()}

    def __contains__(self, key):
        return key in self._v
