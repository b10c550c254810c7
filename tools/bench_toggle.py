#!/usr/bin/env python3
"""Annealing-style moves on the GPU: toggle one token, re-score (moves/sec (ii)).
Bench shard (100 MB wiki-like, V = 32,000) or a CJK-heavy one; tokens are drawn uniformly from the
multi-character part of the vocabulary, as the annealer's proposals are.  Host wall time per
move, incremental path against TOKSCORE_NOINCR=1 (every re-score a full evaluation).
Usage: tools/bench_toggle.py [wiki|cjk|spec] [mb] [moves]
`spec` is SURVEY section 8d's config 5 as written: CJK-heavy 1 GB, V = 32,000, 1,000 single-token
toggles (500 disable / 500 enable, seed 5), each followed by an incremental re-score, the first 100
checked against a full re-score (outside the timed calls)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def run(tokens, corpus, off, moves, seed, paired=False, verify=0):
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    s.score()
    rng = np.random.default_rng(seed)
    multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8", "ignore")) > 1]
    enabled = np.ones(len(tokens), dtype=bool)
    if paired:  # every picked token is disabled once and enabled again later
        toks = rng.choice(multi, size=moves // 2, replace=False)
        picks = rng.permutation(np.concatenate([toks, toks]))
    else:
        picks = rng.choice(multi, size=moves)
    verified = 0
    # the first incremental call builds the occurrence index: timed on its own, then undone
    t0 = time.perf_counter()
    s.toggle(int(picks[0]), False)
    s.score(incremental=True)
    first = time.perf_counter() - t0
    s.toggle(int(picks[0]), True)
    s.score(incremental=True)
    s.drain_timings()
    chunks, wall = [], []
    buf = np.zeros(len(tokens) + 3, dtype=np.uint64)
    t0 = time.perf_counter()
    for i in picks:
        t1 = time.perf_counter()
        enabled[i] = not enabled[i]
        s.toggle(int(i), bool(enabled[i]))
        out = s.score(incremental=True, out=buf)
        wall.append(time.perf_counter() - t1)
        chunks.append(s.info().rescored_chunks)
        if len(wall) <= verify:
            full = s.score()
            assert np.array_equal(full[0], out[0]) and full[1:] == out[1:], "move %d differs from a full re-score" % len(wall)
            verified += 1
    dt = float(np.sum(wall))
    tm = np.array(s.drain_timings())
    if len(tm):
        # an incremental evaluation records: whole | subtract pass | slot patch | add pass (the three
        # passes only with TOKSCORE_TIMING=2; by default just start and end, two events fewer per pass)
        # (a fallback to a full evaluation leaves other entries behind: keep the short ones)
        short = tm[tm[:, 0] < 0.35]
        if len(short):
            print("GPU ms over %d incremental evaluations: whole %.3f  subtract %.3f  patch %.3f  add %.3f"
                  % ((len(short),) + tuple(short.mean(axis=0))), file=sys.stderr)
    full = s.score()
    ok = bool(np.array_equal(full[0], out[0]) and full[1:] == out[1:])
    nchunks = s.info().chunks
    s.close()
    if verify:
        print("verified %d moves against a full re-score" % verified, file=sys.stderr)
    return dt / moves, chunks, nchunks, ok, first, np.array(wall)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "wiki"
    mb = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    moves = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    paired, verify = False, 0
    if kind == "spec":
        kind, paired, verify = "cjk", True, 100
        mb = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
        moves = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    if kind == "wiki":
        src, _ = synth.wiki_like(2_000_000, seed=20201120)
        tokens = synth.build_vocab(src, 32000)
        corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
    else:
        corpus, off = synth.cjk_heavy(int(mb * 1e6), seed=5)
        tokens = synth.build_vocab(corpus[:2_000_000], 32000)
    os.environ.pop("TOKSCORE_NOINCR", None)
    inc, chunks, nchunks, ok, first, wall = run(tokens, corpus, off, moves, 5, paired, verify)
    os.environ["TOKSCORE_NOINCR"] = "1"
    full, _, _, ok2, _, _ = run(tokens, corpus, off, max(20, moves // 5), 5)
    ch = np.array(chunks, dtype=np.int64)
    fast = ch[ch != 0xFFFFFFFF]
    on_fast = wall[ch != 0xFFFFFFFF]
    print(json.dumps({"metric": "GPU toggle + re-score moves/s", "corpus": kind, "bytes": int(len(corpus)), "vocab": len(tokens),
                      "moves": moves, "incremental_ms_per_move": inc * 1e3, "incremental_moves_per_s": 1 / inc,
                      "full_ms_per_move": full * 1e3, "full_moves_per_s": 1 / full, "speedup": full / inc,
                      "paired_toggles": paired, "verified_against_full": verify,
                      "first_incremental_call_ms": first * 1e3,
                      "fast_path_ms_per_move": float(on_fast.mean() * 1e3) if on_fast.size else None,
                      "chunks": int(nchunks), "moves_on_fast_path": int(fast.size),
                      "median_chunks_rescanned": float(np.median(fast)) if fast.size else None,
                      "final_state_equals_full_evaluation": ok and ok2}))


if __name__ == "__main__":
    main()
