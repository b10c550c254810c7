"""Scan time of the automaton scan against the size of the host sample that decides what LDS holds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tokenizer_amd as ta
from tokenizer_amd import synth
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(100_000_000, seed=20201120)
for env in [{"TOKSCORE_DFA_CHAINS": c} for c in sys.argv[1:]]:
    os.environ.update(env)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    for _ in range(3):  # (the first evaluations also choose again what LDS holds)
        s.score()
    ts = []
    for _ in range(8):
        s.score()
        ts.append(s.last_timing())
    ts = np.array(ts[2:])
    i = s.info()
    print(env, os.environ.get("TOKSCORE_SAMPLE_MB"), "eval %.4f scan %.4f fix %.4f hist %.4f ms" % tuple(ts.mean(0)), "lds used", i.dfa_lds_used, "build %.1f ms" % i.dfa_build_ms,
          "relayouts", i.dfa_relayouts, "second hits", i.dfa_second_hits, flush=True)
    s.close()
