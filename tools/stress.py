#!/usr/bin/env python3
"""Scale checks on one GPU against the CPU oracle (tests/oracle_ffi.py), beyond what the unit
tests size for: a 1 GB wiki-like shard (32-bit index headroom), a CJK-heavy shard with
single-token toggles each followed by a re-score (scaled), and a
random-UTF-8 shard with a 128 k vocabulary (config 4, scaled; u32 stream entries).
Usage: tools/stress.py [wiki_mb cjk_mb utf8_mb]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O  # noqa: E402
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def check(name, s, voc, corpus, off, threads=16):
    t0 = time.time()
    got = s.score()
    t1 = time.time()
    exp = voc.score(corpus, off, threads=threads)
    t2 = time.time()
    ok = np.array_equal(got[0], exp[0]) and got[1:] == exp[1:]
    tm = s.last_timing()
    print("%-28s bytes=%d tokens=%d ok=%s gpu eval %.3f ms (%.1f GB/s), first call %.2f s, oracle %.1f s"
          % (name, got[2], got[1], ok, tm[0], got[2] / tm[0] / 1e6, t1 - t0, t2 - t1), flush=True)
    return ok


def main():
    wiki_mb, cjk_mb, utf8_mb = [float(x) for x in (sys.argv[1:4] + ["1000", "200", "200"][len(sys.argv) - 1:])]
    allok = True
    # 1. large wiki-like shard
    src, _ = synth.wiki_like(2_000_000, seed=20201120)
    tokens = synth.build_vocab(src, 32000)
    corpus, off = synth.wiki_like(int(wiki_mb * 1e6), seed=7)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    allok &= check("wiki %.0f MB V=32000" % wiki_mb, s, voc, corpus, off)
    s.score()
    allok &= check("  (second evaluation)", s, voc, corpus, off)
    s.close()
    del corpus
    # 2. CJK-heavy with toggles
    corpus, off = synth.cjk_heavy(int(cjk_mb * 1e6), seed=5)
    sample = corpus[: 2_000_000]
    tokens = synth.build_vocab(sample, 32000)
    s = ta.Scorer(0)
    tb, to = synth.pack_vocab(tokens)
    s.set_vocab(tb, to)
    s.set_corpus(corpus, off)
    voc = O.OracleVocab(tb, to)
    allok &= check("cjk %.0f MB V=%d" % (cjk_mb, len(tokens)), s, voc, corpus, off)
    rng = np.random.default_rng(5)
    enabled = np.ones(len(tokens), dtype=bool)
    multi = [i for i, t in enumerate(tokens) if len(t) > 3]
    t0 = time.time()
    n_tog = 200
    for k in range(n_tog):
        i = int(rng.choice(multi))
        enabled[i] = not enabled[i]
        s.toggle(i, bool(enabled[i]))
        got = s.score(incremental=True)
        if k < 3 or k == n_tog - 1:
            # a disabled entry = a string that can never match (keeps the ids of the rest)
            sub = [t if enabled[j] else b"\x00DISABLED" + str(j).encode() for j, t in enumerate(tokens)]
            v2 = O.OracleVocab(*synth.pack_vocab(sub))
            exp = v2.score(corpus, off, threads=16)
            ok = np.array_equal(got[0], exp[0]) and got[1:] == exp[1:]
            print("  toggle %d (token %d -> %s): ok=%s" % (k, i, enabled[i], ok), flush=True)
            allok &= ok
    print("  %d toggles + re-scores: %.2f ms each (host wall)" % (n_tog, (time.time() - t0 - 0) * 1e3 / n_tog), flush=True)
    s.close()
    del corpus
    # 3. random UTF-8, 128 k vocabulary (wide ids)
    corpus, off = synth.random_utf8(int(utf8_mb * 1e6), seed=355)
    tokens = synth.sample_vocab(corpus[: 50_000_000], 128000)  # deep trie: tokens cut from the corpus itself
    s = ta.Scorer(0)
    tb, to = synth.pack_vocab(tokens)
    s.set_vocab(tb, to)
    s.set_corpus(corpus, off)
    voc = O.OracleVocab(tb, to)
    allok &= check("utf8 %.0f MB V=%d" % (utf8_mb, len(tokens)), s, voc, corpus, off)
    info = s.info()
    print("  trie slots %d, scan path %d, chunks %d" % (info.trie_slots, info.scan_path, info.chunks))
    s.close()
    print("ALL OK" if allok else "MISMATCH")
    sys.exit(0 if allok else 1)


if __name__ == "__main__":
    main()
