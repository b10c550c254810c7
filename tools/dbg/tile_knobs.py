"""Tile scan on the annealed vocabulary under different wave counts / LDS trie shares (debugging aid)."""
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
os.environ["TOKSCORE_DFA"] = "0"
first = None
for spec in sys.argv[2:] or ["16", "12", "8", "6", "4"]:
    w, _, seg = spec.partition(":")
    os.environ["TOKSCORE_WAVES"] = w
    if seg: os.environ["TOKSCORE_SEG"] = seg
    else: os.environ.pop("TOKSCORE_SEG", None)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    r = s.score()
    if first is None: first = r
    for _ in range(2): s.score()
    ts = []
    for _ in range(5):
        r = s.score(); ts.append(s.last_timing())
    assert r[1] == first[1] and np.array_equal(r[0], first[0])
    ts = np.array(ts); i = s.info()
    print("waves", spec, "scan_path", i.scan_path, "eval %.4f scan %.4f fix %.4f hist %.4f ms" % tuple(ts.mean(0)), flush=True)
    s.close()
