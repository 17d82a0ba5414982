"""Row-sharded search over the GPUs of one node: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI), SURVEY.md section 8e.

Rank r holds rows ``[r N / G, (r+1) N / G)`` of the corpus in its own :class:`TheoremIndex` whose
``row_offset`` makes the returned ids global.  A search is: every rank runs the single-GPU kernels
on its shard for the (replicated) query batch, one all-gather exchanges the per-shard top-k
(``nq * k * 12`` bytes per rank), and every rank merges the ``G * k`` candidates per query with the
device merge kernel.  The exchange is latency-bound (20 KB per rank at nq = 256, k = 10); there is no
other collective on the path.

``local_search`` / ``merge`` can be injected so that the partition + exchange logic is testable with
the gloo backend on CPU ranks (tests/test_distributed_cpu.py); the defaults are the HIP kernels.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank`` (balanced to within one row)."""
    return n * rank // world, n * (rank + 1) // world


class ShardedSearcher:
    """Search over a row-sharded corpus; works with host arrays (numpy in, numpy out)."""

    def __init__(self, local_search: Callable[[np.ndarray, int], Tuple[np.ndarray, np.ndarray]],
                 merge: Optional[Callable[[np.ndarray, np.ndarray, int], Tuple[np.ndarray, np.ndarray]]] = None,
                 group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_search = local_search
        if merge is None:
            from .index import merge_topk
            merge = merge_topk
        self.merge = merge

    @classmethod
    def from_local_rows(cls, local_rows, total_rows: int, dtype: str = "bf16", metric: str = "cos", device: int = 0,
                        group=None) -> "ShardedSearcher":
        """Build this rank's shard index from its slice of the corpus (rows ``shard_bounds(...)``)."""
        import torch.distributed as dist
        from .index import TheoremIndex
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(total_rows, world, rank)
        ix = TheoremIndex.from_embeddings(local_rows, dtype=dtype, metric=metric, device=device, row_offset=lo)
        assert ix.n == hi - lo, f"rank {rank} expected {hi - lo} rows, got {ix.n}"
        obj = cls(lambda q, k: ix.search(q, k), group=group)
        obj.index = ix
        return obj

    def search(self, queries: np.ndarray, k: int, mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """``mask``: bool per row of the WHOLE corpus (a metadata filter); every rank applies its slice of it."""
        import torch
        if mask is not None:
            ix = self.index
            lo = ix.row_offset
            scores, idx = ix.search(queries, k, mask=np.asarray(mask, dtype=bool)[lo:lo + ix.n])
        else:
            scores, idx = self.local_search(queries, k)
        if self.world == 1:
            return scores, idx
        nq = scores.shape[0]
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        # one exchange: scores and ids packed into a single int64 payload (score bits in the low word)
        payload = np.empty((nq, k, 2), dtype=np.int64)
        payload[..., 0] = scores.astype(np.float32).view(np.int32).astype(np.int64)
        payload[..., 1] = idx
        mine = torch.from_numpy(payload).to(dev)
        gathered = torch.empty((self.world * nq, k, 2), dtype=torch.int64, device=dev)  # rank-major concatenation
        self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
        g = gathered.cpu().numpy().reshape(self.world, nq, k, 2)
        all_scores = g[..., 0].astype(np.int32).view(np.float32)
        all_idx = np.ascontiguousarray(g[..., 1])
        return self.merge(np.ascontiguousarray(all_scores), all_idx, k)

    def rank_of(self, queries: np.ndarray, rows, local_rank_of=None, local_count_above=None) -> np.ndarray:
        """Rank of global row ``rows[i]`` for query ``i`` over the WHOLE sharded corpus (0 = best; -1 = no such row):
        the owning shard scores the row (`TheoremIndex.rank_of`), one all-reduce (max) makes the scores known to every
        rank, every other shard counts its rows that rank before ``(score, row)`` (`count_above`), one all-reduce
        (sum) adds the counts.  The two callables default to this rank's index; they can be injected for CPU tests."""
        import torch
        ix = getattr(self, "index", None)
        local_rank_of = local_rank_of or (lambda q, r: ix.rank_of(q, r))
        local_count_above = local_count_above or (lambda q, s, r: ix.count_above(q, s, r))
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        ranks, scores = local_rank_of(queries, rows)
        if self.world == 1:
            return ranks
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        mine = ranks >= 0
        sc = torch.from_numpy(np.where(mine, scores, -np.inf).astype(np.float32)).to(dev)
        self.dist.all_reduce(sc, op=self.dist.ReduceOp.MAX, group=self.group)  # exactly one shard owns each row
        sc_host = sc.cpu().numpy()
        known = np.isfinite(sc_host)
        counts = np.where(mine, ranks, 0).astype(np.int64)
        others = known & ~mine
        if others.any():
            c = local_count_above(queries[others], sc_host[others], rows[others])
            counts[others] = c
        total = torch.from_numpy(counts).to(dev)
        self.dist.all_reduce(total, op=self.dist.ReduceOp.SUM, group=self.group)
        out = total.cpu().numpy()
        out[~known] = -1
        return out
