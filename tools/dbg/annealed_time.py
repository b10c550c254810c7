"""Full-evaluation time with the vocabulary the reference's annealer wrote (tests/golden/wiki_annealed) on 100 MB of
the text it was annealed on, automaton scan on / off (debugging aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth
tokens = O.load_golden("wiki_annealed")["tokens"]
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
corpus, off = synth.wiki_like(mb * 1_000_000, seed=1234, n_words=20_000)
for dfa in sys.argv[2:] or ["1", "0"]:
    os.environ["TOKSCORE_DFA"] = dfa
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    t0 = time.time()
    first = s.score()
    t1 = time.time()
    for _ in range(3):
        s.score()
    ts = []
    for _ in range(6):
        r = s.score()
        ts.append(s.last_timing())
    assert r[1] == first[1] and np.array_equal(r[0], first[0])
    ts = np.array(ts)
    i = s.info()
    print("dfa", dfa, "active", i.dfa_active, "scan_path", i.scan_path, "base", i.base_scan_path, "first %.1f ms" % ((t1 - t0) * 1e3),
          "eval %.4f scan %.4f fix %.4f hist %.4f ms" % tuple(ts.mean(0)), "total_tokens", r[1], "lds used", i.dfa_lds_used,
          "hot frac %.3f" % i.dfa_hot_fraction, "second hits", i.dfa_second_hits, flush=True)
    s.close()
