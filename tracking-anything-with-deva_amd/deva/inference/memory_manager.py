"""`MemoryManager` with the reference's interface (deva/inference/memory_manager.py:14-292): the
sensory / working / long-term memories and the transitions between them, driven by the fused
gfx950 kernels of libdeva_hip:

* `match_memory`: per bucket ONE fused similarity -> top-k -> softmax (+usage) pass over the
  virtual concatenation [long-term rows | working rows] of the token-major banks, then one sparse
  readout per object.  Nothing of size N x HW is ever materialised (the reference builds and
  re-reads that matrix about a dozen times, memory_utils.py:29-74, and `torch.cat`s the whole bank
  first, memory_manager.py:110-113).
* `add_memory` / `compress_features` / `consolidation`: appends are transpose-copies into the
  arenas; consolidation runs the dense potentiation step on device and performs the prototype
  value/shrinkage readouts as fp32-MFMA GEMMs.
* `shard_queries(group)`: one clip on several GPUs (SURVEY.md §8e "replicate bank, shard queries").
  Every rank steps the same clip and therefore holds an identical bank (all kernels are
  deterministic, so the replicas never diverge); the memory read -- the only part of a frame that
  splits -- is partitioned by query column: rank r matches and reads out columns [r*per, (r+1)*per),
  the read-out columns are all-gathered and the fixed-point usage counters all-reduced (integer
  sums: exact in any order).  Results are bit-identical to the unsharded read.
"""
from typing import Dict, List, Optional, Tuple

import torch

from deva.hip import ops
from deva.inference.kv_memory_store import KeyValueMemoryStore


class StackedReadout(dict):
    """{object id: CV*H*W read-out} whose values are consecutive rows of ONE [objects, CV, H, W] tensor
    (`stack`, rows in `order`), so that `ObjectManager.realize_dict` can hand that tensor to the decoder
    without re-stacking it"""

    def __init__(self, stack: Optional[torch.Tensor] = None, order=()):
        super().__init__({obj: stack[i] for i, obj in enumerate(order)} if stack is not None else {})
        self.stack, self.order = stack, list(order)


class MemoryManager:
    """
    Manages all three memory stores and the transition between working/long-term memory
    """

    # 1..32: the list kernels (fp16 pre-filter / fused fp32 kernels); 33..64: the dense kernel behind deva_affinity_read
    # (correct, ~4 ms per read at 1080p / 10k tokens); the read-out gathers one term per lane: 64 at most
    MAX_TOP_K = 64
    MAX_TOP_K_SHARDED_BANK = 32  # the hand-over format of a token-sharded bank holds <= 32 entries per range

    @classmethod
    def _checked_top_k(cls, top_k) -> int:
        if top_k is None or not 1 <= int(top_k) <= cls.MAX_TOP_K:
            raise ValueError(f'top_k={top_k} is not supported by the HIP memory read: the limit is {cls.MAX_TOP_K} (1..32 on '
                             f'the list kernels, 33..{cls.MAX_TOP_K} on the dense kernel; the reference default is 30)')
        return int(top_k)

    def __init__(self, config: Dict):
        self.sensory_dim = config['value_dim']
        self.key_dim = config.get('key_dim', 64)
        self.top_k = self._checked_top_k(config['top_k'])
        self.use_long_term = config['enable_long_term']
        self.count_long_term_usage = config['enable_long_term_count_usage']
        self.chunk_size = config['chunk_size']
        if self.use_long_term:
            self.max_mem_frames = config['max_mid_term_frames']
            self.min_mem_frames = config['min_mid_term_frames']
            self.num_prototypes = config['num_prototypes']
            self.max_long_tokens = config['max_long_term_elements']

        # inferred from the first memory frame
        self.CK = self.CV = None
        self.H = self.W = None

        # sensory memory, indexed by object id, each C^h x H x W
        self.sensory: Dict[int, torch.Tensor] = {}
        self._sensory_stack: Optional[torch.Tensor] = None  # [1,no,C,h,w] backing the dict entries
        self._sensory_ids: List[int] = []

        self.work_mem = KeyValueMemoryStore(save_selection=self.use_long_term,
                                            save_usage=self.use_long_term)
        if self.use_long_term:
            self.long_mem = KeyValueMemoryStore(save_usage=self.count_long_term_usage)

        self._usage_fix: Optional[torch.Tensor] = None  # int64 fixed-point usage scratch, kept zeroed
        self._shard_group = None  # torch.distributed group the memory read is sharded over
        self._values_sharded = False  # shard_bank(shard_values=True): each rank stores 1/world of the value rows
        self._shard_mode: Optional[str] = None  # 'queries' | 'bank'
        self._shard_owner: Optional[int] = None  # group rank that owns the encoder / decoder (frame-owner mode)
        self.comm_bytes = 0  # bytes this rank sent + received in collectives since sharding was enabled

        self.config_stale = True
        self.engaged = False

    def update_config(self, config: Dict) -> None:
        # memory_manager.py:47-62
        self.config_stale = True
        self.sensory_dim = config['value_dim']
        top_k = self._checked_top_k(config['top_k'])
        if getattr(self, '_shard_mode', None) == 'bank' and self._shard_group is not None and top_k > self.MAX_TOP_K_SHARDED_BANK:
            raise ValueError(f'top_k={top_k} > {self.MAX_TOP_K_SHARDED_BANK} (the largest top_k of a token-sharded bank: its '
                             f'hand-over format holds {self.MAX_TOP_K_SHARDED_BANK} entries per range; --top_k up to '
                             f'{self.MAX_TOP_K} needs an unsharded or query-sharded read)')
        self.top_k = top_k
        assert self.use_long_term == config['enable_long_term'], 'cannot update this'
        assert self.count_long_term_usage == config['enable_long_term_count_usage'], 'cannot update this'
        if self.use_long_term:
            self.max_mem_frames = config['max_mid_term_frames']
            self.min_mem_frames = config['min_mid_term_frames']
            self.num_prototypes = config['num_prototypes']
            self.max_long_tokens = config['max_long_term_elements']

    # ------------------------------------------------------------------ read
    def _long_term_mem_available(self) -> bool:
        return self.use_long_term and self.long_mem.engaged()

    def _usage_scratch(self, n: int, device) -> torch.Tensor:
        if self._usage_fix is None or self._usage_fix.numel() < n:
            self._usage_fix = torch.zeros(max(2 * n, 1 << 16), dtype=torch.int64, device=device)
        return self._usage_fix

    # ------------------------------------------------------------------ one clip on several GPUs
    def shard_queries(self, group=None, owner: Optional[int] = None) -> None:
        """Partition every following `match_memory` by QUERY COLUMN over `group` (default: the world
        group); every rank keeps a full replica of the bank (SURVEY.md 8e "replicate bank, shard queries").
        owner=None: every rank steps the same clip and receives every read-out (all-gather).
        owner=r   : frame-owner mode -- rank r alone runs the encoder / decoder; `DEVAInferenceCore.step`
                    broadcasts the query key / selection from it, the read-out columns are gathered to it
                    only, and on memory frames it broadcasts the new key / shrinkage / selection / value
                    rows to the other ranks' banks.  Only `step` is routed in this mode
                    (`incorporate_detection` raises NotImplementedError)."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('shard_queries: torch.distributed is not initialised')
        if getattr(self, '_values_sharded', False):
            # the value rows of a value-sharded manager live on different ranks: a query-sharded read would return
            # un-summed partial read-outs
            raise RuntimeError('shard_queries: this manager stores value-sharded banks (shard_bank(shard_values=True)); '
                               'build a new MemoryManager to change the mode')
        self._shard_group = group if group is not None else dist.group.WORLD
        self._shard_mode = 'queries'
        self._shard_owner = owner
        self.comm_bytes = 0

    def shard_bank(self, group=None, shard_values: bool = True, owner: Optional[int] = None) -> None:
        """Partition every following `match_memory` by MEMORY TOKEN RANGE over `group` (SURVEY.md 8e
        "shard the bank"): rank r matches the queries against rows [r*per, (r+1)*per) of the virtual
        long-then-work bank only, the per-shard top-k candidates (64-bit score|token keys, hw*k*8 B per
        rank) are all-gathered and merged to the exact global top-k on every rank, each rank reads out
        the value rows it holds and the partial read-outs are summed (all-reduce).  The merged selection,
        weights and usage counters are bit-identical to the unsharded read.

        shard_values=True (default; call before the first memory frame): the STORAGE is partitioned too --
        each rank keeps 1/world of the value rows of both stores (`KeyValueMemoryStore.shard_values`: of every
        memory frame / prototype batch rank r appends only its block; sieves and evictions drop only local rows;
        a consolidation forms the prototype values from the local candidate rows and all-reduces the P x CV
        partial sums).  Keys, shrinkage, selection and usage counters (131 of the 131 + 512 x objects floats of a
        token) stay replicated: every rank scores and ranks them, which is what keeps the decisions identical.

        owner=r: additionally frame-owner mode (see `shard_queries`): rank r alone runs the encoder / decoder and
        broadcasts the query and the new memory rows; the partial read-outs are then REDUCED to it (`dist.reduce`,
        half the bytes of the all-reduce every-rank-decodes needs) and nobody else receives them.  The configuration
        for a bank that outgrows one GPU: every rank stores and scores 1/world of it, one rank decodes."""
        import torch.distributed as dist
        if self.top_k > self.MAX_TOP_K_SHARDED_BANK:
            raise ValueError(f'shard_bank: top_k={self.top_k} > {self.MAX_TOP_K_SHARDED_BANK} is served by the dense read kernel, '
                             'which has no per-shard hand-over; use shard_queries or a smaller top_k')
        if not dist.is_initialized():
            raise RuntimeError('shard_bank: torch.distributed is not initialised')
        self._shard_group = group if group is not None else dist.group.WORLD
        if dist.get_world_size(self._shard_group) > 32:
            # deva_affinity_merge takes at most 32 candidate lists (MAX_SPLITS, include/deva_hip.h)
            raise ValueError('shard_bank: at most 32 ranks per group (one candidate list per rank is merged)')
        if shard_values and (self.work_mem.buckets or (self.use_long_term and self.long_mem.buckets)):
            raise RuntimeError('shard_bank(shard_values=True) must be called before the first memory frame: the stores '
                               'already hold replicated value rows (pass shard_values=False to shard the read only)')
        self._shard_mode = 'bank'
        self._shard_owner = owner
        self.comm_bytes = 0
        self._values_sharded = bool(shard_values)
        if shard_values:
            rank, world = dist.get_rank(self._shard_group), dist.get_world_size(self._shard_group)
            self.work_mem.shard_values(rank, world)
            if self.use_long_term:
                self.long_mem.shard_values(rank, world)

    @property
    def is_frame_owner(self) -> bool:
        """True unless another rank owns the encoder / decoder of this clip"""
        if self._shard_group is None or self._shard_owner is None:
            return True
        import torch.distributed as dist
        return dist.get_rank(self._shard_group) == self._shard_owner

    def _owner_global_rank(self) -> int:
        import torch.distributed as dist
        return dist.get_global_rank(self._shard_group, self._shard_owner)

    def _shard_range(self, n: int) -> Tuple[int, int, int, int]:
        """-> (rank, world, items per rank, first item of this rank) for n columns / tokens"""
        import torch.distributed as dist
        world = dist.get_world_size(self._shard_group)
        rank = dist.get_rank(self._shard_group)
        per = -(-n // world)
        return rank, world, per, rank * per

    def _readout_into(self, out: torch.Tensor, idx, weight, bucket_id: int, obj: int, with_long: bool, n_long: int,
                      tok_range=None) -> None:
        obj_long = with_long and obj in self.long_mem
        if getattr(self, '_values_sharded', False):
            # value-sharded storage: this rank adds the terms of the rows it holds (tok_range does not apply)
            ops.readout_sparse(idx, weight, self.long_mem.value_arena(obj) if obj_long else None,
                               n_long if obj_long else 0, self.work_mem.value_arena(obj), out,
                               row_map_long=self.long_mem.row_map(bucket_id) if obj_long else None,
                               row_map_work=self.work_mem.row_map(bucket_id))
            return
        ops.readout_sparse(idx, weight, self.long_mem.value_arena(obj) if obj_long else None,
                           n_long if obj_long else 0, self.work_mem.value_arena(obj), out, tok_range=tok_range)

    def match_memory(self, query_key: torch.Tensor, selection: torch.Tensor) -> Dict[int, torch.Tensor]:
        """query_key, selection: 1 x C^k x H x W  ->  {object id: C^v x H x W readout}
        (memory_manager.py:91-169).  The read-outs of all objects are views of ONE [objects, C^v, H, W]
        tensor in bucket order (`ObjectManager.realize_dict` returns it without a copy when the order is
        the tmp-id order).  In frame-owner mode the other ranks return an empty dict."""
        assert query_key.shape[0] == 1
        h, w = query_key.shape[-2:]
        hw = h * w
        qk = query_key[0].reshape(query_key.shape[1], hw)
        qe = selection[0].reshape(selection.shape[1], hw)
        mode = self._shard_mode if self._shard_group is not None else None
        order = [obj for bucket in self.work_mem.buckets.values() for obj in bucket]
        receives = mode != 'queries' or self.is_frame_owner
        stack = (torch.empty((len(order), self.CV, h, w), dtype=torch.float32, device=qk.device)
                 if receives else None)
        at = 0
        for bucket_id, bucket in self.work_mem.buckets.items():
            rows = None if stack is None else stack[at:at + len(bucket)]
            if mode == 'queries':
                self._read_bucket_query_sharded(bucket_id, bucket, qk, qe, rows)
            elif mode == 'bank':
                self._read_bucket_bank_sharded(bucket_id, bucket, qk, qe, rows)
            else:
                self._read_bucket(bucket_id, bucket, qk, qe, rows)
            at += len(bucket)
        if stack is None:
            return StackedReadout()
        return StackedReadout(stack, order)

    def _bucket_extent(self, bucket_id: int):
        with_long = self.use_long_term and self.long_mem.engaged(bucket_id)
        n_long = self.long_mem.size(bucket_id) if with_long else 0
        return with_long, n_long, self.work_mem.size(bucket_id)

    def _apply_usage(self, bucket_id: int, usage_fix, with_long: bool, n_long: int) -> None:
        # usage bookkeeping (memory_manager.py:128-152)
        self.work_mem.apply_usage_fix(bucket_id, usage_fix, n_long)
        if with_long:
            self.long_mem.apply_usage_fix(bucket_id, usage_fix, 0)

    def _read_bucket(self, bucket_id: int, bucket: List[int], qk, qe, rows: torch.Tensor) -> None:
        with_long, n_long, n_work = self._bucket_extent(bucket_id)
        usage_fix = self._usage_scratch(n_long + n_work, qk.device) if self.use_long_term else None
        idx, weight = ops.affinity_topk(
            self.long_mem.key_arena(bucket_id) if with_long else None,
            self.long_mem.shrinkage_arena(bucket_id) if with_long else None, n_long,
            self.work_mem.key_arena(bucket_id), self.work_mem.shrinkage_arena(bucket_id), n_work,
            qk, qe, self.top_k, usage_fix, **self._prep_of(bucket_id, with_long))
        if self.use_long_term:
            self._apply_usage(bucket_id, usage_fix, with_long, n_long)
        for i, obj in enumerate(bucket):
            self._readout_into(rows[i], idx, weight, bucket_id, obj, with_long, n_long)

    def _prep_of(self, bucket_id: int, with_long: bool) -> dict:
        """the bucket's prepared pre-filter operands and the key of the bank's present state: the bank changes on memory
        frames only (add / consolidation / eviction / purge bump the stores' bucket versions), the frames between read it
        unchanged and skip the bank kernels of the read (memory_manager.py:91-169 reads, :171-218 writes)"""
        if not getattr(self, 'bank_prep_enabled', True):  # (tests: the same clip with and without the cache)
            return {}
        preps = self.__dict__.setdefault('_bank_prep', {})
        for b in [b for b in preps if b not in self.work_mem.buckets]:  # purged buckets
            del preps[b]
        prep = preps.get(bucket_id)
        if prep is None:
            prep = preps[bucket_id] = ops.BankPrep()
        key = (self.work_mem.version(bucket_id), self.long_mem.version(bucket_id) if with_long else -1,
               self.work_mem.key_arena(bucket_id).data_ptr(), self.long_mem.key_arena(bucket_id).data_ptr() if with_long else 0)
        return dict(prep=prep, prep_key=key)

    def _read_bucket_query_sharded(self, bucket_id: int, bucket: List[int], qk, qe, rows) -> None:
        """rank r matches and reads out query columns [r*per, (r+1)*per); the column slabs are gathered
        (to every rank, or to the frame owner only) and the fixed-point usage counters all-reduced
        (integer sums: exact in any order)"""
        import torch.distributed as dist
        hw = qk.shape[1]
        rank, world, per, lo = self._shard_range(hw)
        n_mine = max(0, min(hw, lo + per) - lo)
        # an empty tail rank still matches one column (the kernels need hw >= 1); its result is
        # discarded and its usage contribution suppressed
        cols = slice(lo, lo + n_mine) if n_mine else slice(0, 1)
        qk_r, qe_r = qk[:, cols].contiguous(), qe[:, cols].contiguous()
        with_long, n_long, n_work = self._bucket_extent(bucket_id)
        usage_fix = self._usage_scratch(n_long + n_work, qk.device) if self.use_long_term else None
        idx, weight = ops.affinity_topk(
            self.long_mem.key_arena(bucket_id) if with_long else None,
            self.long_mem.shrinkage_arena(bucket_id) if with_long else None, n_long,
            self.work_mem.key_arena(bucket_id), self.work_mem.shrinkage_arena(bucket_id), n_work,
            qk_r, qe_r, self.top_k, usage_fix if (self.use_long_term and n_mine) else None,
            **self._prep_of(bucket_id, with_long))
        if self.use_long_term:
            dist.all_reduce(usage_fix[:n_long + n_work], op=dist.ReduceOp.SUM, group=self._shard_group)
            self.comm_bytes += 8 * (n_long + n_work)
            self._apply_usage(bucket_id, usage_fix, with_long, n_long)
        # this rank's [objects, CV, per] slab; the read-out kernel writes straight into it when the rank
        # has a full set of columns (every rank but possibly the last)
        nobj = len(bucket)
        mine = torch.empty((nobj, self.CV, per), dtype=torch.float32, device=qk.device)
        for i, obj in enumerate(bucket):
            if n_mine == per:
                self._readout_into(mine[i], idx, weight, bucket_id, obj, with_long, n_long)
            else:
                part = torch.empty((self.CV, qk_r.shape[1]), dtype=torch.float32, device=qk.device)
                self._readout_into(part, idx, weight, bucket_id, obj, with_long, n_long)
                mine[i].zero_()
                if n_mine:
                    mine[i, :, :n_mine] = part
        slab_bytes = mine.numel() * 4
        if self._shard_owner is None:
            gathered = torch.empty((world, nobj, self.CV, per), dtype=torch.float32, device=qk.device)
            dist.all_gather_into_tensor(gathered.view(world * nobj, self.CV, per), mine, group=self._shard_group)
            self.comm_bytes += slab_bytes * (world - 1)
            slabs = list(gathered.unbind(0))
        elif self.is_frame_owner:
            slabs = [torch.empty_like(mine) for _ in range(world)]
            dist.gather(mine, slabs, dst=self._owner_global_rank(), group=self._shard_group)
            self.comm_bytes += slab_bytes * (world - 1)
        else:
            dist.gather(mine, None, dst=self._owner_global_rank(), group=self._shard_group)
            self.comm_bytes += slab_bytes
            return
        # [rank][obj, CV, per] column slabs -> [obj, CV, hw] (one strided copy per slab, ragged tail cut)
        flat = rows.view(nobj, self.CV, hw)
        for r, slab in enumerate(slabs):
            c0 = r * per
            c1 = min(hw, c0 + per)
            if c1 > c0:
                flat[:, :, c0:c1] = slab[:, :, :c1 - c0]

    def _sum_partial_readouts(self, rows: torch.Tensor) -> None:
        """partial read-outs of the ranks' value rows -> their sum: on every rank, or on the frame owner only"""
        import torch.distributed as dist
        world = dist.get_world_size(self._shard_group)
        if self._shard_owner is None:
            dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self._shard_group)
            self.comm_bytes += 2 * rows.numel() * 4 * (world - 1) // world
        else:
            dist.reduce(rows, dst=self._owner_global_rank(), op=dist.ReduceOp.SUM, group=self._shard_group)
            self.comm_bytes += rows.numel() * 4 * (world - 1) // world

    def _read_bucket_bank_sharded(self, bucket_id: int, bucket: List[int], qk, qe, rows) -> None:
        import torch.distributed as dist
        with_long, n_long, n_work = self._bucket_extent(bucket_id)
        n = n_long + n_work
        rank, world, per, lo = self._shard_range(n)
        if n < max(128, self.top_k + world) * world:  # first frames of a clip: every shard must hold >= top_k tokens
            self._read_bucket(bucket_id, bucket, qk, qe, rows)  # (every rank scores the whole replicated key bank)
            if getattr(self, '_values_sharded', False):  # ... but holds only its share of the value rows
                self._sum_partial_readouts(rows)
            return
        hi = min(n, lo + per)
        hw = qk.shape[1]
        # this rank's rows of the virtual bank [long | work]
        l0, l1 = min(lo, n_long), min(hi, n_long)
        w0, w1 = max(lo, n_long) - n_long, max(hi, n_long) - n_long
        keys, counts = ops.affinity_candidates(
            self.long_mem.key_arena(bucket_id)[l0:l1] if l1 > l0 else None,
            self.long_mem.shrinkage_arena(bucket_id)[l0:l1] if l1 > l0 else None, l1 - l0,
            self.work_mem.key_arena(bucket_id)[w0:w1] if w1 > w0 else None,
            self.work_mem.shrinkage_arena(bucket_id)[w0:w1] if w1 > w0 else None, w1 - w0,
            qk, qe, self.top_k, token_offset=lo)
        all_keys = torch.empty((world, *keys.shape), dtype=keys.dtype, device=keys.device)
        all_counts = torch.empty((world, *counts.shape), dtype=counts.dtype, device=counts.device)
        dist.all_gather_into_tensor(all_keys.view(world * hw, -1), keys, group=self._shard_group)
        dist.all_gather_into_tensor(all_counts.view(-1), counts, group=self._shard_group)
        self.comm_bytes += (keys.numel() * 8 + counts.numel() * 4) * (world - 1)
        usage_fix = self._usage_scratch(n, qk.device) if self.use_long_term else None
        idx, weight = ops.affinity_merge(all_keys, all_counts, self.top_k, usage_fix)
        if self.use_long_term:  # every rank merged the complete selection: the counters are already global
            self._apply_usage(bucket_id, usage_fix, with_long, n_long)
        for i, obj in enumerate(bucket):
            self._readout_into(rows[i], idx, weight, bucket_id, obj, with_long, n_long, tok_range=(lo, hi))
        self._sum_partial_readouts(rows)

    def broadcast_memory_frame(self, key, shrinkage, value, selection, objects: List[int], h: int, w: int, device):
        """frame-owner mode, memory frame: the owner's new key (1*CK*h*w), shrinkage (1*1*h*w), selection
        (1*CK*h*w) and value (1*objects*CV*h*w) rows are broadcast as one packed buffer so that every rank
        appends the same tokens to its replica -- the 'RCCL all-gather of memory keys' of BASELINE.json.
        The other ranks pass None for the tensors and receive them."""
        import torch.distributed as dist
        ck, cv = self.key_dim, self.sensory_dim
        nobj = len(objects)
        if self.is_frame_owner:
            packed = torch.cat([key[0].reshape(ck, h * w), shrinkage[0].reshape(1, h * w),
                                selection[0].reshape(ck, h * w), value[0].reshape(nobj * cv, h * w)], 0)
        else:
            packed = torch.empty((2 * ck + 1 + nobj * cv, h * w), dtype=torch.float32, device=device)
        dist.broadcast(packed, src=self._owner_global_rank(), group=self._shard_group)
        self.comm_bytes += packed.numel() * 4
        return (packed[:ck].view(1, ck, h, w), packed[ck:ck + 1].view(1, 1, h, w),
                packed[2 * ck + 1:].view(1, nobj, cv, h, w), packed[ck + 1:2 * ck + 1].view(1, ck, h, w))

    def broadcast_query(self, key, selection, h: int, w: int, device):
        """frame-owner mode, every frame: the owner's query key / selection (1*CK*h*w each) -> all ranks"""
        import torch.distributed as dist
        ck = self.key_dim
        packed = (torch.cat([key[0].reshape(ck, h * w), selection[0].reshape(ck, h * w)], 0) if self.is_frame_owner
                  else torch.empty((2 * ck, h * w), dtype=torch.float32, device=device))
        dist.broadcast(packed, src=self._owner_global_rank(), group=self._shard_group)
        self.comm_bytes += packed.numel() * 4
        return packed[:ck].view(1, ck, h, w), packed[ck:].view(1, ck, h, w)

    # ------------------------------------------------------------------ write
    def add_memory(self, key: torch.Tensor, shrinkage: torch.Tensor, value: torch.Tensor,
                   objects: List[int], selection: torch.Tensor = None) -> None:
        """key 1*C*H*W, shrinkage 1*1*H*W, value 1*num_objects*C*H*W (memory_manager.py:171-218)"""
        self.engaged = True
        if self.H is None or self.config_stale:
            self.config_stale = False
            self.H, self.W = value.shape[-2:]
            self.HW = self.H * self.W
            if self.use_long_term:
                self.max_work_tokens = self.max_mem_frames * self.HW
                self.min_work_tokens = self.min_mem_frames * self.HW

        key = key[0].flatten(start_dim=1)
        shrinkage = shrinkage[0].flatten(start_dim=1)
        self.CK = key.shape[0]
        value = value[0].flatten(start_dim=2)
        self.CV = value.shape[1]
        if selection is not None:
            selection = selection[0].flatten(start_dim=1)

        self.work_mem.add(key, {obj: value[i] for i, obj in enumerate(objects)}, shrinkage, selection)

        if self.use_long_term:
            for bucket_id in list(self.work_mem.buckets.keys()):
                if self.work_mem.size(bucket_id) >= self.max_work_tokens:
                    room = self.max_long_tokens - self.num_prototypes
                    if self.long_mem.size(bucket_id) >= room:
                        self.long_mem.remove_obsolete_features(bucket_id, room)
                    self.compress_features(bucket_id)

    def purge_except(self, obj_keep_idx: List[int]) -> None:
        # memory_manager.py:220-229
        self.work_mem.purge_except(obj_keep_idx)
        if self._long_term_mem_available():
            self.long_mem.purge_except(obj_keep_idx)
        self.sensory = {k: v for k, v in self.sensory.items() if k in obj_keep_idx}
        self._sensory_stack, self._sensory_ids = None, []
        if not self.work_mem.engaged():
            self.engaged = False

    def compress_features(self, bucket_id: int) -> None:
        """move the middle frames of the working memory into `num_prototypes` long-term tokens
        (memory_manager.py:231-249)"""
        HW = self.HW
        n = self.work_mem.size(bucket_id)
        lo, hi = HW, n - (self.min_work_tokens - HW)
        objs = self.work_mem.buckets[bucket_id]
        use, life = self.work_mem.usage_arenas(bucket_id)
        # value-sharded storage: this rank holds rows [v_lo, v_hi) of the candidates, `owned` = their offsets in [lo, hi)
        v_lo, v_hi, owned = self.work_mem.local_rows(bucket_id, lo, hi)
        proto_key, proto_val, proto_shr = self.consolidation(
            self.work_mem.key_arena(bucket_id)[lo:hi], self.work_mem.shrinkage_arena(bucket_id)[lo:hi],
            self.work_mem.selection_arena(bucket_id)[lo:hi],
            {o: self.work_mem.value_arena(o)[v_lo:v_hi] for o in objs}, (use[lo:hi], life[lo:hi]), owned_rows=owned)
        self.work_mem.sieve_by_range(bucket_id, HW, -self.min_work_tokens + HW,
                                     min_size=self.min_work_tokens + HW)
        self.long_mem.add(proto_key, proto_val, proto_shr, selection=None, supposed_bucket_id=bucket_id,
                          token_major=True)

    def consolidation(self, candidate_key: torch.Tensor, candidate_shrinkage: torch.Tensor,
                      candidate_selection: torch.Tensor, candidate_value: Dict[int, torch.Tensor],
                      usage: Tuple[torch.Tensor, torch.Tensor], *, owned_rows: Optional[torch.Tensor] = None):
        """memory_manager.py:251-276 on TOKEN-MAJOR candidates: key/selection [Nc,CK], shrinkage
        [Nc], values {obj: [Nc,CV]}, usage = (use_cnt, life_cnt) rows.  Returns token-major
        prototype key [P,CK], values {obj: [P,CV]}, shrinkage [P].
        owned_rows (value-sharded storage): `candidate_value` holds only the candidates this rank owns (their
        offsets among the Nc candidates); the P x CV partial sums are all-reduced over the shard group."""
        n_cand = candidate_key.shape[0]
        P = self.num_prototypes
        # prototypes = the P candidates with the highest normalised usage (torch.topk, sorted)
        rank_desc, _ = ops.rank(usage[0], n_cand, True, life=usage[1])
        proto_idx = ops.rank_select(rank_desc, P)
        # potentiation: softmax over ALL candidates for every prototype query
        aff = ops.similarity_dense(candidate_key, candidate_shrinkage, candidate_selection, proto_idx, n_cand)
        ops.softmax_columns(aff, P)
        gemm = ops.PackedConv(aff, None, n_cand, P, aff.shape[1], 1, 1)  # weight[k=n][m=p]
        proto_key = torch.empty((P, candidate_key.shape[1]), dtype=torch.float32, device=aff.device)
        ops.bank_gather_rows(candidate_key, proto_idx, proto_key, P)
        proto_val = {}
        if owned_rows is not None:
            import torch.distributed as dist
            n_own = int(owned_rows.numel())
            gemm_own = ops.PackedConv(aff.index_select(0, owned_rows).contiguous(), None, n_own, P, aff.shape[1], 1, 1) if n_own else None
            # the partial sums of ALL objects travel in one collective (one all_reduce per object is latency-bound with
            # many objects; the sum per element is the same)
            objs = list(candidate_value)
            world = dist.get_world_size(self._shard_group)
            parts = []
            for obj in objs:
                v = candidate_value[obj]
                cv = v.shape[1]
                parts.append(ops.conv2d(gemm_own, v.contiguous().reshape(1, n_own, 1, cv)).view(P, cv) if n_own
                             else torch.zeros((P, cv), dtype=torch.float32, device=aff.device))
            if parts:
                stacked = torch.stack(parts, 0)
                dist.all_reduce(stacked, op=dist.ReduceOp.SUM, group=self._shard_group)
                self.comm_bytes += 2 * stacked.numel() * 4 * (world - 1) // world  # like _sum_partial_readouts
                for i, obj in enumerate(objs):
                    proto_val[obj] = stacked[i]
        for obj, v in ([] if owned_rows is not None else candidate_value.items()):
            cv = v.shape[1]
            proto_val[obj] = ops.conv2d(gemm, v.reshape(1, n_cand, 1, cv)).view(P, cv)
        proto_shr = ops.conv2d(gemm, candidate_shrinkage.reshape(1, n_cand, 1, 1)).view(P)
        return proto_key, proto_val, proto_shr

    # ------------------------------------------------------------------ sensory memory
    def initialize_sensory_if_needed(self, sample_key: torch.Tensor, ids: List[int]):
        # memory_manager.py:278-283
        for obj in ids:
            if obj not in self.sensory:
                h, w = sample_key.shape[-2:]
                self.sensory[obj] = torch.zeros((self.sensory_dim, h, w), device=sample_key.device)

    def update_sensory(self, sensory: torch.Tensor, ids: List[int]):
        # sensory: 1*num_objects*C*H*W  (memory_manager.py:285-288)
        self._sensory_stack, self._sensory_ids = sensory, list(ids)
        for obj_id, obj in enumerate(ids):
            self.sensory[obj] = sensory[0, obj_id]

    def get_sensory(self, ids: List[int]) -> torch.Tensor:
        # returns 1*num_objects*C*H*W  (memory_manager.py:290-292)
        if self._sensory_stack is not None and list(ids) == self._sensory_ids and all(
                self.sensory[o].data_ptr() == self._sensory_stack[0, i].data_ptr() for i, o in enumerate(ids)):
            return self._sensory_stack
        # a changed object set (detections added / purged objects): re-stack into a GUARD-BANDED tensor -- the stack feeds
        # the GRU convolutions, whose vector gathers (and with them the f16 kernels) need readable slack around their
        # inputs; a plain torch.stack result sent those launches down the scalar-gather fp32 kernels
        rows = [self.sensory[obj] for obj in ids]
        out = ops._alloc((1, len(rows), *rows[0].shape), rows[0].device)
        torch.stack(rows, dim=0, out=out[0])
        return out
