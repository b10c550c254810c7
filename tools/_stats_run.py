#!/usr/bin/env python3
"""Debug only: runs the instrumented library variant (tools/_libtokscore_stats.so) and prints
wave-step counters per phase."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta  # noqa: E402
ta.LIB_PATH = os.path.join(ROOT, "tools", "_libtokscore_stats.so")
from tokenizer_amd import synth  # noqa: E402

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
tb, to = synth.pack_vocab(tokens)
s = ta.Scorer(0)
s.set_vocab(tb, to)
s.set_corpus(corpus, off)
lib = ta.load_library()
out = (C.c_ulonglong * 8)()
s.score()
lib.tokscore_debug_stats(out)
s.score()
lib.tokscore_debug_stats(out)
p1, fast, dual, neg, tiles, rounds = [out[i] for i in range(6)]
print("tiles %d  per tile: phase1 %.1f  fast %.1f  neg %.1f  dual %.1f  rounds %.2f  total steps %.1f" %
      (tiles, p1 / tiles, fast / tiles, neg / tiles, dual / tiles, rounds / tiles, (p1 + fast + neg + dual) / tiles))
print("timing", s.last_timing())
