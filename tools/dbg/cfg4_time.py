#!/usr/bin/env python3
"""Config 4 (random UTF-8, V = 128k) kernel times only.  Usage: tools/dbg/cfg4_time.py [mb]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
corpus, off = synth.random_utf8(int(mb * 1e6), seed=355)
tokens = synth.sample_vocab(corpus[:50_000_000], 128_000)
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens))
s.set_corpus(corpus, off)
out = s.score()
ts = []
for _ in range(8):
    s.score()
    ts.append(s.last_timing())
ms = np.median(np.array(ts), axis=0)
c = out[0][2:]
print("lib", ta.LIB_PATH, "eval_ms %.3f scan_ms %.3f hist_reduce_ms %.3f" % (ms[0], ms[1], ms[3]),
      "tokens", out[1], "ids used", int((c > 0).sum()), "top16 share %.3f" % (np.sort(c)[::-1][:16].sum() / c.sum()),
      "top4096 share %.3f" % (np.sort(c)[::-1][:4096].sum() / c.sum()))
