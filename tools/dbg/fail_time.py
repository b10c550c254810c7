"""The failure form of the automaton scan (dfa.h: DfaImage::fail_code) against the tile scan and the flattened form on the
vocabulary the reference's annealer wrote, or on the bench vocabulary (debugging aid).
usage: fail_time.py MB [annealed|bench|cjk] form[:chains] ...   form: tile, flat, fail, auto"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
which = sys.argv[2] if len(sys.argv) > 2 else "annealed"
if which == "annealed":
    tokens = O.load_golden("wiki_annealed")["tokens"]
    corpus, off = synth.wiki_like(mb * 1_000_000, seed=1234, n_words=20_000)
elif which == "cjk":
    corpus, off = synth.cjk_heavy(mb * 1_000_000, seed=5)
    tokens = synth.build_vocab(corpus[:2_000_000], 32000)
else:
    corpus, off = synth.wiki_like(mb * 1_000_000, seed=1)
    tokens = synth.build_vocab(corpus[:2_000_000], 32000)
first = None
if mb <= 20:
    first = O.OracleVocab(*synth.pack_vocab(tokens)).score(corpus, off, threads=8)
for spec in sys.argv[3:] or ["tile", "fail"]:
    form, _, chains = spec.partition(":")
    os.environ["TOKSCORE_DFA"] = "0" if form == "tile" else "1"
    os.environ["TOKSCORE_DFA_FORM"] = {"tile": "0", "flat": "1", "fail": "2", "auto": "0"}[form]
    if form == "flat":
        os.environ["TOKSCORE_DFA_MAX_COLD_PPM"] = os.environ["TOKSCORE_DFA_MAX_VIRTUAL_PPM"] = "1000000"
    else:
        os.environ.pop("TOKSCORE_DFA_MAX_COLD_PPM", None), os.environ.pop("TOKSCORE_DFA_MAX_VIRTUAL_PPM", None)
    if chains: os.environ["TOKSCORE_DFA_CHAINS"] = chains
    else: os.environ.pop("TOKSCORE_DFA_CHAINS", None)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    t0 = time.time()
    r = s.score()
    t1 = time.time()
    if first is None: first = r
    ok = r[1] == first[1] and np.array_equal(r[0], first[0])
    for _ in range(3):
        s.score()
    ts = []
    for _ in range(6):
        r = s.score()
        ts.append(s.last_timing())
    ok = ok and r[1] == first[1] and np.array_equal(r[0], first[0])
    ts = np.array(ts)
    i = s.info()
    print(spec, "EXACT" if ok else "WRONG", "form", i.dfa_form, "active", i.dfa_active, "scan_path", i.scan_path, "first %.1f ms" % ((t1 - t0) * 1e3),
          "eval %.4f scan %.4f fix %.4f hist %.4f ms" % tuple(ts.mean(0)), "total_tokens", r[1], "lds used", i.dfa_lds_used,
          "hot frac %.3f" % i.dfa_hot_fraction, "second hits", i.dfa_second_hits, "relayouts", i.dfa_relayouts, flush=True)
    s.close()
