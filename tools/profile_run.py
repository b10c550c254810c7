#!/usr/bin/env python3
"""Workload for rocprofv3: the bench.py shard (wiki-like, V=32k) scored a few times.
The generated corpus is cached under /tmp so that several profiler passes in one
gpurun call do not regenerate it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    evals = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cache = "/tmp/tokscore_profile_%d.npz" % int(mb)
    if os.path.exists(cache):
        z = np.load(cache)
        corpus, off, tb, to = z["corpus"], z["off"], z["tb"], z["to"]
    else:
        vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
        tokens = synth.build_vocab(vocab_src, 32000)
        corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
        tb, to = synth.pack_vocab(tokens)
        np.savez(cache, corpus=corpus, off=off, tb=tb, to=to)
    s = ta.Scorer(0)
    s.set_vocab(tb, to)
    s.set_corpus(corpus, off)
    for _ in range(evals):
        out = s.score()
    print("bytes", out[2], "tokens", out[1], "timing", s.last_timing())
    s.close()


if __name__ == "__main__":
    main()
