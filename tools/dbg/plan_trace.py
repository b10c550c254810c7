"""Host phases of the other scan paths' plan (TOKSCORE_TRACE_HOST=1), asked for by the first toggle after evaluations on
the automaton.  usage: plan_trace.py MB"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(src, 32000)
corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens))
s.set_corpus(corpus, off)
a = s.score(); s.score(); s.score()
multi = [i for i, t in enumerate(tokens) if len(t) > 3]
print("---- toggle", file=sys.stderr, flush=True)
t0 = time.perf_counter(); s.toggle(int(multi[len(multi) // 2]), False); t1 = time.perf_counter()
print("---- evaluation after the toggle", file=sys.stderr, flush=True)
b = s.score(); t2 = time.perf_counter()
print("toggle %.1f ms, evaluation after it %.1f ms (total tokens %d -> %d)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, a[1], b[1]), flush=True)
s.close()
