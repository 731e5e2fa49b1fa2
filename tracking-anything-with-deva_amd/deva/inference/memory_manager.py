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


class MemoryManager:
    """
    Manages all three memory stores and the transition between working/long-term memory
    """

    MAX_TOP_K = 32  # the fused affinity kernel keeps one candidate per lane pair: 1 <= top_k <= 32

    @classmethod
    def _checked_top_k(cls, top_k) -> int:
        if top_k is None or not 1 <= int(top_k) <= cls.MAX_TOP_K:
            raise ValueError(f'top_k={top_k} is not supported by the HIP memory read (1..{cls.MAX_TOP_K}; the '
                             'reference default is 30)')
        return int(top_k)

    def __init__(self, config: Dict):
        self.sensory_dim = config['value_dim']
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

        self.config_stale = True
        self.engaged = False

    def update_config(self, config: Dict) -> None:
        # memory_manager.py:47-62
        self.config_stale = True
        self.sensory_dim = config['value_dim']
        self.top_k = self._checked_top_k(config['top_k'])
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

    def shard_queries(self, group=None) -> None:
        """Partition every following `match_memory` by query column over `group` (default: the world
        group).  All ranks of the group must step the same clip."""
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('shard_queries: torch.distributed is not initialised')
        self._shard_group = group if group is not None else dist.group.WORLD

    def _shard_range(self, hw: int) -> Tuple[int, int, int, int]:
        """-> (rank, world, columns per rank, first column of this rank)"""
        import torch.distributed as dist
        world = dist.get_world_size(self._shard_group)
        rank = dist.get_rank(self._shard_group)
        per = -(-hw // world)
        return rank, world, per, rank * per

    def match_memory(self, query_key: torch.Tensor, selection: torch.Tensor) -> Dict[int, torch.Tensor]:
        """query_key, selection: 1 x C^k x H x W  ->  {object id: C^v x H x W readout}
        (memory_manager.py:91-169)"""
        assert query_key.shape[0] == 1
        h, w = query_key.shape[-2:]
        hw = h * w
        qk = query_key[0].reshape(query_key.shape[1], hw)
        qe = selection[0].reshape(selection.shape[1], hw)
        sharded = self._shard_group is not None
        if sharded:
            import torch.distributed as dist
            rank, world, per, lo = self._shard_range(hw)
            n_mine = max(0, min(hw, lo + per) - lo)
            # this rank's query columns (an empty tail rank still matches one column: the kernels
            # need hw >= 1; its result is discarded and its usage contribution masked below)
            cols = slice(lo, lo + n_mine) if n_mine else slice(0, 1)
            qk, qe = qk[:, cols].contiguous(), qe[:, cols].contiguous()
        readouts: Dict[int, torch.Tensor] = {}
        for bucket_id, bucket in self.work_mem.buckets.items():
            with_long = self.use_long_term and self.long_mem.engaged(bucket_id)
            n_long = self.long_mem.size(bucket_id) if with_long else 0
            n_work = self.work_mem.size(bucket_id)
            count_usage = self.use_long_term and not (sharded and n_mine == 0)
            usage_fix = self._usage_scratch(n_long + n_work, qk.device) if self.use_long_term else None
            idx, weight = ops.affinity_topk(
                self.long_mem.key_arena(bucket_id) if with_long else None,
                self.long_mem.shrinkage_arena(bucket_id) if with_long else None, n_long,
                self.work_mem.key_arena(bucket_id), self.work_mem.shrinkage_arena(bucket_id), n_work,
                qk, qe, self.top_k, usage_fix if count_usage else None)
            if self.use_long_term:
                if sharded:
                    dist.all_reduce(usage_fix[:n_long + n_work], op=dist.ReduceOp.SUM, group=self._shard_group)
                # usage bookkeeping (memory_manager.py:128-152)
                self.work_mem.apply_usage_fix(bucket_id, usage_fix, n_long)
                if with_long:
                    self.long_mem.apply_usage_fix(bucket_id, usage_fix, 0)
            if not sharded:
                for obj in bucket:
                    obj_long = with_long and obj in self.long_mem
                    out = torch.empty((self.CV, h, w), dtype=torch.float32, device=qk.device)
                    ops.readout_sparse(idx, weight, self.long_mem.value_arena(obj) if obj_long else None,
                                       n_long if obj_long else 0, self.work_mem.value_arena(obj), out)
                    readouts[obj] = out
                continue
            # sharded: [objects, CV, per] column slabs of this rank -> all-gather -> [objects, CV, hw]
            nq = qk.shape[1]
            mine = torch.zeros((len(bucket), self.CV, per), dtype=torch.float32, device=qk.device)
            for i, obj in enumerate(bucket):
                obj_long = with_long and obj in self.long_mem
                out = torch.empty((self.CV, nq), dtype=torch.float32, device=qk.device)
                ops.readout_sparse(idx, weight, self.long_mem.value_arena(obj) if obj_long else None,
                                   n_long if obj_long else 0, self.work_mem.value_arena(obj), out)
                if n_mine:
                    mine[i, :, :n_mine] = out
            gathered = torch.empty((world * len(bucket), self.CV, per), dtype=torch.float32, device=qk.device)
            dist.all_gather_into_tensor(gathered, mine, group=self._shard_group)  # rank-major concatenation
            full = gathered.view(world, len(bucket), self.CV, per).permute(1, 2, 0, 3).reshape(len(bucket), self.CV, world * per)[:, :, :hw]
            for i, obj in enumerate(bucket):
                readouts[obj] = full[i].reshape(self.CV, h, w).contiguous()
        return readouts

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
        proto_key, proto_val, proto_shr = self.consolidation(
            self.work_mem.key_arena(bucket_id)[lo:hi], self.work_mem.shrinkage_arena(bucket_id)[lo:hi],
            self.work_mem.selection_arena(bucket_id)[lo:hi],
            {o: self.work_mem.value_arena(o)[lo:hi] for o in objs}, (use[lo:hi], life[lo:hi]))
        self.work_mem.sieve_by_range(bucket_id, HW, -self.min_work_tokens + HW,
                                     min_size=self.min_work_tokens + HW)
        self.long_mem.add(proto_key, proto_val, proto_shr, selection=None, supposed_bucket_id=bucket_id,
                          token_major=True)

    def consolidation(self, candidate_key: torch.Tensor, candidate_shrinkage: torch.Tensor,
                      candidate_selection: torch.Tensor, candidate_value: Dict[int, torch.Tensor],
                      usage: Tuple[torch.Tensor, torch.Tensor]):
        """memory_manager.py:251-276 on TOKEN-MAJOR candidates: key/selection [Nc,CK], shrinkage
        [Nc], values {obj: [Nc,CV]}, usage = (use_cnt, life_cnt) rows.  Returns token-major
        prototype key [P,CK], values {obj: [P,CV]}, shrinkage [P]."""
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
        for obj, v in candidate_value.items():
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
        return torch.stack([self.sensory[obj] for obj in ids], dim=0).unsqueeze(0)
