"""Time of a call of 256 moves with parts of search_items_kernel switched off (debugging aid; results are wrong with dbg != 0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(mb * 1_000_000, seed=20201120)
rng = np.random.default_rng(5)
multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8", "ignore")) > 1]
batches = [rng.choice(multi, size=K, replace=False) for _ in range(6)]
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens))
s.set_corpus(corpus, off)
s.score()
s.score_toggles(batches[0], 0)
for dbg in sys.argv[3:] or ["0", "1", "2", "4", "6"]:
    os.environ["TOKSCORE_LOCAL_BUDGET"] = dbg
    t0 = time.perf_counter()
    for b in batches:
        s.score_toggles(b, 0)
    dt = time.perf_counter() - t0
    print("budget", dbg, "ms per call %.3f" % (dt * 1e3 / len(batches)), flush=True)
