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


def merge_candidates_across_ranks(tokens, dist, device="cpu"):
    """Candidate extraction shards by documents like the scoring (GetTokens feeds file after file
    into one trie, /root/reference/src/tokenizer/GetTokens.cpp:31-90): every rank extracts the
    candidates of its own documents, and the sets are merged pairwise up a binary tree of ranks
    (log2(world) rounds of point-to-point transfers -- on xGMI each round uses disjoint links).
    Returns the merged `annealing.Tokens` on rank 0, None on the others.  `tokens` is consumed.
    `device`: where the transfer buffers live ("cuda" for the RCCL backend, "cpu" for gloo)."""
    import torch

    from . import annealing as A

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return tokens
    rank, world = dist.get_rank(), dist.get_world_size()
    views = (np.int32, np.int64, np.int32, np.int32)  # chr, uses, lpar, rpar as torch-friendly signed views
    step = 1
    while step < world:
        if rank % (2 * step) == step:
            arrays = tokens.arrays()
            tokens.close()
            dist.send(torch.tensor([len(arrays[0])], dtype=torch.int64, device=device), dst=rank - step)
            for a, v in zip(arrays, views):
                if len(a):
                    dist.send(torch.from_numpy(a.view(v)).to(device), dst=rank - step)
            return None
        if rank % (2 * step) == 0 and rank + step < world:
            n = torch.zeros(1, dtype=torch.int64, device=device)
            dist.recv(n, src=rank + step)
            got = []
            for v, u in zip(views, (np.uint32, np.uint64, np.uint32, np.uint32)):
                t = torch.zeros(int(n.item()), dtype=torch.int32 if v is np.int32 else torch.int64, device=device)
                if int(n.item()):
                    dist.recv(t, src=rank + step)
                got.append(t.cpu().numpy().view(u))
            other = A.Tokens.from_arrays(*got)
            merged = tokens.merge(other)
            tokens.close()
            other.close()
            tokens = merged
        step *= 2
    return tokens
