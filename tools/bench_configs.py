#!/usr/bin/env python3
"""The other BASELINE configurations at single-GPU size (parity-test cases, not the bench line):
   config 4 of SURVEY.md section 8 -- random UTF-8 over a 4k alphabet, V = 128k tokens cut from the
   corpus (deep trie, gather stress) -- and config 5's corpus (CJK-heavy, V = 32k), one full
   evaluation each, kernel times from the library's own events, checked against the CPU oracle on
   the first documents.  Usage: tools/bench_configs.py [mb]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402
import oracle_ffi as O  # noqa: E402  (the checker, never the thing measured)


def run(name, corpus, off, tokens):
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    out = s.score()
    ts = []
    for _ in range(8):
        s.score()
        ts.append(s.last_timing())
    ms = np.median(np.array(ts), axis=0)
    info = s.info()
    # parity on a bounded sample: the first documents (about 8 MB), scored on their own
    d = int(np.searchsorted(off, 8_000_000))
    d = max(1, min(d, len(off) - 1))
    sub_c, sub_o = corpus[:int(off[d])], off[:d + 1]
    s.set_corpus(sub_c, sub_o)
    got = s.score()
    exp = O.OracleVocab(*synth.pack_vocab(tokens)).score(sub_c, sub_o, threads=16)
    ok = bool(np.array_equal(got[0], exp[0]) and got[1:] == exp[1:])
    s.close()
    rec = {"config": name, "bytes": int(len(corpus)), "docs": int(len(off) - 1), "vocab": len(tokens),
           "trie_slots": int(info.trie_slots), "lds_slots": int(info.trie_lds_slots), "scan_path": int(info.scan_path),
           "total_tokens": int(out[1]), "eval_ms": float(ms[0]), "scan_ms": float(ms[1]), "hist_reduce_ms": float(ms[3]),
           "GBps_eval": len(corpus) / float(ms[0]) / 1e6, "sample_parity": ok}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
    corpus, off = synth.random_utf8(int(mb * 1e6), seed=355)
    run("config 4: random UTF-8, V=128k sampled from the corpus", corpus, off, synth.sample_vocab(corpus[:50_000_000], 128_000))
    corpus, off = synth.cjk_heavy(int(mb * 1e6), seed=5)
    run("config 5 corpus: CJK-heavy, V=32k", corpus, off, synth.build_vocab(corpus[:2_000_000], 32000))


if __name__ == "__main__":
    main()
