#!/usr/bin/env python3
"""tokscore_score_toggles on the bench shard: K independent moves per call, tokens drawn uniformly
from the multi-character part of the vocabulary (as the annealer's proposals are).
Usage: tools/bench_score_toggles.py [mb] [K] [calls]     -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
    tokens = synth.build_vocab(vocab_src, 32000)
    corpus, off = synth.wiki_like(mb * 1_000_000, seed=20201120)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    base = s.score()
    rng = np.random.default_rng(5)
    multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8", "ignore")) > 1]
    t0 = time.perf_counter()
    s.score_toggles(rng.choice(multi, size=K, replace=False), 0)
    first = time.perf_counter() - t0
    streams0 = s.info().toggle_streams
    batches = [rng.choice(multi, size=K, replace=False) for _ in range(calls)]
    t0 = time.perf_counter()
    deltas = [s.score_toggles(b, 0) for b in batches]
    dt = time.perf_counter() - t0
    streams = s.info().toggle_streams - streams0
    exact = True
    for k in range(0, K, max(1, K // 4)):
        s.toggle(int(batches[0][k]), False)
        tt = s.score()[1]
        s.toggle(int(batches[0][k]), True)
        exact = exact and int(deltas[0][k]) == tt - base[1]
    print(json.dumps({"mb": mb, "K": K, "calls": calls, "moves_per_s": K * calls / dt, "ms_per_call": dt * 1e3 / calls,
                      "first_call_ms": first * 1e3, "streams_walked_per_move": streams / (K * calls),
                      "streams_in_corpus": int(len(corpus) // 256), "exact": exact}))
    s.close()


if __name__ == "__main__":
    main()
