#!/usr/bin/env python3
"""An annealer-like loop on the bench shard: K moves judged (tokscore_score_toggles), one of them
applied (tokscore_toggle), the exact score taken (tokscore_score), again.  ms per batch by part.
Usage: tools/dbg/anneal_like.py [K] [batches]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta
from tokenizer_amd import synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 30
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(100_000_000, seed=20201120)
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens))
s.set_corpus(corpus, off)
s.score()
rng = np.random.default_rng(7)
multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8", "ignore")) > 1]
state = {}
s.score_toggles(rng.choice(multi, size=K, replace=False), 0)
tj = ta_ = ts = 0.0
for b in range(B):
    idx = rng.choice(multi, size=K, replace=False)
    en = [0 if state.get(int(i), True) else 1 for i in idx]
    t0 = time.perf_counter()
    d = s.score_toggles(idx, en)
    t1 = time.perf_counter()
    k = int(np.argmin(d))
    s.toggle(int(idx[k]), bool(en[k]))
    state[int(idx[k])] = bool(en[k])
    t2 = time.perf_counter()
    r = s.score()
    t3 = time.perf_counter()
    tj += t1 - t0; ta_ += t2 - t1; ts += t3 - t2
print("K=%d: judge %.3f ms, apply %.3f ms, exact score %.3f ms per batch -> %.0f moves/s judged" % (K, tj / B * 1e3, ta_ / B * 1e3, ts / B * 1e3, K * B / (tj + ta_ + ts)))
