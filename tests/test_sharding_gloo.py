"""N > 1 path on CPU: world_size-2 gloo.  Each rank takes its byte-balanced document shard,
produces the V+5 result vector and all-reduces it exactly as bench.py does over RCCL.  The
per-rank scorer here is the CPU oracle (test stand-in: the GPU scorer needs a GPU); what is
under test is the partition and the reduction, against the oracle on the whole corpus."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import subprocess
import sys

import oracle_ffi as O
from tokenizer_amd import annealing as A
from tokenizer_amd import shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tokens, corpus, off, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench  # the step, the reduction, the timing loop and --split's partition are bench.py's own code, not a copy of it
    c, o = bench.split_shard(corpus, off, rank, world)
    c2, o2 = shard.shard_corpus(corpus, off, rank, world)
    assert np.array_equal(c, c2) and np.array_equal(o, o2)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    t = torch.zeros(len(tokens) + 5, dtype=torch.int64)
    calls = []

    def step():  # bench.py's step with the CPU oracle standing in for score_async (no GPU here)
        counts, tt, tb = voc.score(c, o)
        vec = np.concatenate([counts, np.array([tt, tb], dtype=np.uint64)])
        t.copy_(torch.from_numpy(vec.view(np.int64).copy()))
        bench.reduce_result(t, dist, world, host_reduce=True)
        calls.append(1)

    elapsed = bench.timed_steps(step, dist.barrier, steps=2, warmup=1)
    assert len(calls) == 3 and elapsed > 0
    slowest = bench.max_over_ranks(elapsed, dist, world, "cpu")
    assert slowest >= elapsed
    if rank == 0:
        q.put(t.numpy().view(np.uint64).copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_tile_the_corpus_and_balance_bytes():
    corpus, off = synth.wiki_like(3_000_000, seed=4)
    for world in (1, 2, 3, 8):
        ranges = [shard.shard_range(off, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == len(off) - 1
        for a, b in zip(ranges, ranges[1:]):
            assert a[1] == b[0]
        sizes = [int(off[d1] - off[d0]) for d0, d1 in ranges]
        assert sum(sizes) == len(corpus)
        assert max(sizes) - min(sizes) < 300_000  # within one (long) article of each other
    # degenerate: more ranks than documents
    tiny = np.array([0, 5, 9], dtype=np.uint64)
    rs = [shard.shard_range(tiny, r, 4) for r in range(4)]
    assert rs[0][0] == 0 and rs[-1][1] == 2 and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))


@pytest.mark.timeout(300)
def test_two_rank_gloo_allreduce_equals_single_evaluation():
    corpus, off = synth.wiki_like(1_500_000, seed=12)
    tokens = synth.build_vocab(corpus[:400_000], 4000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    counts, tt, tb = voc.score(corpus, off)
    # START/END are per document, so the reduced counts[0:2] must equal the document count
    expect = np.concatenate([counts, np.array([tt, tb], dtype=np.uint64)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tokens, corpus, off, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, expect)


def _extract_worker(rank, world, port, corpus, off, max_cp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c, o = shard.shard_corpus(corpus, off, rank, world)
    # (the host builder stands in for the GPU extraction, which gives the same arrays: test_gpu_anneal.py)
    merged = shard.merge_candidates_across_ranks(A.GetTokens(c, o, max_cp), dist)
    assert (merged is not None) == (rank == 0)
    if rank == 0:
        q.put(merged.arrays())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_candidates_merged_up_the_rank_tree_equal_one_extraction(world):
    corpus, off = synth.wiki_like(400_000, seed=21, n_words=3000, median_doc=600)
    expect = A.GetTokens(corpus, off, 6).arrays()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_extract_worker, args=(r, world, port, corpus, off, 6, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(got, expect):
        assert np.array_equal(a, b)


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` starts N ranks by itself; with fewer than N GPUs visible it must fail
    loudly, never run fewer ranks and print a line that says n_gpus = 1."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert "refusing" in r.stderr and "n_gpus" not in r.stdout
