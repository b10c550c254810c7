"""Batched moves by the one-pass search against the occurrence-index path: same deltas, time per call (debugging aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(mb * 1_000_000, seed=20201120)
rng = np.random.default_rng(5)
multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8", "ignore")) > 1]
batches = [rng.choice(multi, size=K, replace=False) for _ in range(calls)]
res = {}
for mode in sys.argv[4:] or ["1", "0"]:
    os.environ["TOKSCORE_SEARCH"] = mode
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    s.score()
    t0 = time.perf_counter()
    s.score_toggles(batches[0], 0)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    deltas = [s.score_toggles(b, 0) for b in batches]
    dt = time.perf_counter() - t0
    res[mode] = np.array(deltas)
    print("search", mode, "first call %.1f ms" % (first * 1e3), "ms per call %.4f" % (dt * 1e3 / calls), "moves/s %.0f" % (K * calls / dt),
          "items per move %.0f" % (s.info().toggle_streams / (K * (calls + 1))), flush=True)
    s.close()
if len(res) == 2:
    a, b = res.values()
    print("same deltas:", bool(np.array_equal(a, b)), "differing", int((a != b).sum()), "of", a.size)
    if not np.array_equal(a, b):
        i = np.argwhere(a != b)[:10]
        print([(tuple(x), int(a[tuple(x)]), int(b[tuple(x)])) for x in i])
        sys.exit(1)
