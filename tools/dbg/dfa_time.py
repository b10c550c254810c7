"""Kernel times of a full evaluation of the bench shard: automaton scan (chains 1/2/4) vs the ranges scan."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tokenizer_amd as ta
from tokenizer_amd import synth

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
ref = None
for env in [{"TOKSCORE_DFA": "0"}] + [{"TOKSCORE_DFA": "1", "TOKSCORE_DFA_CHAINS": c} for c in sys.argv[2:] or ["1", "2", "4"]]:
    os.environ.update(env)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    t0 = time.time()
    out = s.score()
    first = time.time() - t0
    if ref is None:
        ref = out
    ok = np.array_equal(out[0], ref[0]) and out[1:] == ref[1:]
    ts = []
    for _ in range(8):
        s.score()
        ts.append(s.last_timing())
    ts = np.array(ts[2:])
    i = s.info()
    print(env, "exact" if ok else "MISMATCH", "first %.1f ms" % (first * 1e3), "eval %.4f scan %.4f fix %.4f hist %.4f ms" % tuple(ts.mean(0)),
          "dfa_active", i.dfa_active, "slots", i.dfa_slots, "lds", i.dfa_lds_slots, "hot %.4f" % i.dfa_hot_fraction, "range", i.dfa_range_bytes,
          "build %.1f plan %.1f ms" % (i.dfa_build_ms, i.dfa_plan_ms), flush=True)
    s.close()
