"""Where a fresh context's first evaluation goes (host clock around the C-ABI calls; TOKSCORE_TRACE_HOST=1 adds the
library's own phases on stderr).  Two contexts one after the other: the second finds the code objects loaded."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(src, 32000)
corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
tb, to = synth.pack_vocab(tokens)
for rnd in range(2):
    t = [time.perf_counter()]
    s = ta.Scorer(0); t.append(time.perf_counter())
    s.set_vocab(tb, to); t.append(time.perf_counter())
    s.set_corpus(corpus, off); t.append(time.perf_counter())
    print("---- first evaluation, context %d" % rnd, file=sys.stderr, flush=True)
    s.score(); t.append(time.perf_counter())
    print("---- second evaluation", file=sys.stderr, flush=True)
    s.score(); t.append(time.perf_counter())
    s.score(); t.append(time.perf_counter())
    d = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print("context %d: create %.1f set_vocab %.1f set_corpus %.1f first %.1f second %.1f third %.2f ms" % (rnd, *d), flush=True)
    s.close()
