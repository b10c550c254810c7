#!/usr/bin/env python3
"""Config 5 corpus (CJK-heavy, V = 32k) kernel times under the environment's plan knobs.  Usage: tools/dbg/cfg5_time.py [mb]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
corpus, off = synth.cjk_heavy(int(mb * 1e6), seed=5)
tokens = synth.build_vocab(corpus[:2_000_000], 32000)
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens))
s.set_corpus(corpus, off)
out = s.score()
ts = []
for _ in range(8):
    s.score()
    ts.append(s.last_timing())
ms = np.median(np.array(ts), axis=0)
info = s.info()
print({k: os.environ[k] for k in os.environ if k.startswith("TOKSCORE_")}, "path", info.scan_path, "lds_slots", info.trie_lds_slots, "of", info.trie_slots,
      "eval_ms %.3f scan_ms %.3f hist_reduce_ms %.3f" % (ms[0], ms[1], ms[3]), "GB/s %.0f" % (len(corpus) / ms[0] / 1e6), "tokens", out[1])
