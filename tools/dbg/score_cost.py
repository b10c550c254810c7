import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import tokenizer_amd as ta
from tokenizer_amd import synth
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(100_000_000, seed=20201120)
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens)); s.set_corpus(corpus, off)
buf = np.zeros(len(tokens) + 3, dtype=np.uint64)
s.score(out=buf); s.drain_timings()
t0 = time.perf_counter()
for _ in range(30): s.score(out=buf)
print("score() unchanged vocabulary: %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3), [round(x, 3) for x in np.mean(s.drain_timings(), axis=0)])
rng = np.random.default_rng(3)
multi = [i for i, t in enumerate(tokens) if len(t) > 1]
picks = rng.choice(multi, size=30, replace=False)
t0 = time.perf_counter()
for i in picks:
    s.toggle(int(i), False); s.score(out=buf)
print("toggle + score(): %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3), [round(x, 3) for x in np.mean(s.drain_timings(), axis=0)])

t0 = time.perf_counter()
for _ in range(30): s.score(out=buf)
print("score() unchanged vocabulary, again: %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3), [round(x, 3) for x in np.mean(s.drain_timings(), axis=0)])
import ctypes
t0 = time.perf_counter()
for _ in range(30): s.score(out=buf); time.sleep(0)
print("... with a yield between calls: %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
