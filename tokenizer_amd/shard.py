"""Document sharding for multi-GPU scoring (one process per GPU).

Documents are independent (SolutionFile::Tokenize has no cross-document state,
/root/reference/src/files/SolutionFile.cpp:83-100; the reference parallelises over articles,
/root/reference/src/main.cpp:38-43), so a corpus is split into contiguous document ranges
balanced by BYTES (article lengths are heavy-tailed) and the only exchange is one SUM
all-reduce of counts | total_tokens | total_bytes (u64, wrap-around addition)."""
from __future__ import annotations

import numpy as np


def shard_range(doc_off: np.ndarray, rank: int, world: int):
    """[d0, d1) of the documents rank `rank` scores; ranges tile [0, D) in rank order."""
    off = np.asarray(doc_off, dtype=np.uint64)
    D = len(off) - 1
    if world <= 1:
        return 0, D
    total = int(off[-1] - off[0])
    cuts = [0]
    for r in range(1, world):
        target = int(off[0]) + total * r // world
        d = int(np.searchsorted(off, target, side="left"))
        cuts.append(min(max(d, cuts[-1]), D))
    cuts.append(D)
    return cuts[rank], cuts[rank + 1]


def shard_corpus(corpus: np.ndarray, doc_off: np.ndarray, rank: int, world: int):
    """(bytes, offsets) of rank's shard, offsets rebased to 0."""
    d0, d1 = shard_range(doc_off, rank, world)
    off = np.asarray(doc_off, dtype=np.uint64)
    lo, hi = int(off[d0]), int(off[d1])
    return corpus[lo:hi], (off[d0:d1 + 1] - off[d0]).astype(np.uint64)


def all_reduce_counts(vec, dist=None):
    """In-place SUM all-reduce of the V+5 result vector (an int64 torch tensor viewing the u64
    words: two's-complement addition is the u64 wrap-around sum).  `dist` = torch.distributed."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec)
    return vec
