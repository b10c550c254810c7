"""Where the time before the first result goes (host clock): set_corpus, the first evaluation with and without the
automaton scan, the automaton's own plan.  Usage: tools/dbg/plan_cost.py [mb]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(src, 32000)
corpus, off = synth.wiki_like(mb * 1_000_000, seed=20201120)
for dfa in ("0", "1"):
    os.environ["TOKSCORE_DFA"] = dfa
    s = ta.Scorer(0)
    t0 = time.perf_counter(); s.set_vocab(*synth.pack_vocab(tokens)); t1 = time.perf_counter()
    s.set_corpus(corpus, off); t2 = time.perf_counter()
    ev = []
    for _ in range(5):
        a = time.perf_counter(); s.score(); ev.append((time.perf_counter() - a) * 1e3)
    i = s.info()
    print("dfa", dfa, "mb", mb, "set_vocab %.1f ms set_corpus %.1f ms evaluations %s ms; automaton plan %.1f ms (image %.1f ms)"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ["%.1f" % x for x in ev], i.dfa_plan_ms, i.dfa_build_ms), flush=True)
    s.close()
