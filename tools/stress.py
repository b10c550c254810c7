#!/usr/bin/env python3
"""Scale checks on one GPU against the CPU oracle (tests/oracle_ffi.py), beyond what the unit
tests size for: a 1 GB wiki-like shard (32-bit index headroom), a CJK-heavy shard with
single-token toggles each followed by a re-score (scaled), and a
random-UTF-8 shard with a 128 k vocabulary (config 4, scaled; u32 stream entries).
Usage: tools/stress.py [wiki_mb cjk_mb utf8_mb]
       tools/stress.py big [wiki_mb utf8_mb]   config 2's per-GPU share (2.5 GB wiki-like, V = 32 k) and
                                               config 4 at full size (1.25 GB random UTF-8, V = 128 k):
                                               one JSON record each (profiles/r02_other_configs.json)"""
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


def big(name, config, tokens, corpus, off, sample_bytes=8_000_000):
    """One shard at full per-GPU size: the first sample_bytes against the CPU oracle, then properties
    that hold at any size -- START/END once per document, the counts add up to total_tokens,
    total_bytes is the corpus, and a second evaluation gives the same vector.  -> a JSON record."""
    import json
    tb, to = synth.pack_vocab(tokens)
    V = len(tokens)
    D = len(off) - 1
    voc = O.OracleVocab(tb, to)
    d = int(np.searchsorted(off, sample_bytes, side="right")) - 1
    d = max(d, 1)
    s = ta.Scorer(0)
    s.set_vocab(tb, to)
    s.set_corpus(corpus[:int(off[d])], off[:d + 1])
    got = s.score()
    exp = voc.score(corpus[:int(off[d])], off[:d + 1], threads=16)
    sample_ok = bool(np.array_equal(got[0], exp[0]) and got[1:] == exp[1:])
    t0 = time.time()
    s.set_corpus(corpus, off)
    t1 = time.time()
    counts, tt, tby = s.score()
    first_s = time.time() - t1
    info = s.info()
    evals = []
    again = None
    for _ in range(3):  # (the automaton scan chooses again what LDS holds after its first two evaluations)
        again = s.score()
    for _ in range(5):
        again = s.score()
        evals.append(s.last_timing())
    tm = [float(np.median([e[k] for e in evals])) for k in range(4)]
    props = {"start_end_once_per_document": bool(int(counts[0]) == D and int(counts[1]) == D),
             "counts_add_up_to_total_tokens": bool(int(counts[2:].sum()) == tt),
             "total_bytes_is_the_corpus": bool(tby == int(off[-1])),
             "second_evaluation_identical": bool(np.array_equal(again[0], counts) and again[1:] == (tt, tby))}
    # incremental re-score and batched moves at this size: one rare token off and on again
    rare = int(np.argmin(np.where(counts[2:2 + V] > 0, counts[2:2 + V], np.iinfo(np.int64).max)))
    delta = int(s.score_toggles([rare], [0])[0])
    s.toggle(rare, False)
    inc = s.score(incremental=True)
    resc = int(s.info().rescored_chunks)
    s.toggle(rare, True)
    back = s.score(incremental=True)
    props["toggle_delta_equals_incremental_rescore"] = bool(inc[1] - tt == delta)
    props["toggled_back_identical"] = bool(np.array_equal(back[0], counts) and back[1] == tt)
    rec = {"config": config, "bytes": int(tby), "docs": D, "vocab": V, "trie_slots": int(info.trie_slots),
           "lds_slots": int(info.trie_lds_slots), "scan_path": int(info.scan_path), "chunks": int(info.chunks),
           "total_tokens": int(tt), "set_corpus_s": t1 - t0, "first_evaluation_s": first_s,
           "eval_ms": tm[0], "scan_ms": tm[1], "fixup_ms": tm[2], "hist_reduce_ms": tm[3],
           "GBps_eval": tby / tm[0] / 1e6 if tm[0] else 0.0, "roofline_frac_scan": (tby / tm[1] / 1e6 / 8000.0) if tm[1] else 0.0,
           "automaton_scan": int(info.dfa_active), "automaton_form": {0: "none", 1: "flattened", 2: "failure"}[int(info.dfa_form)], "sample_bytes": int(off[d]), "sample_parity": sample_ok,
           "rare_token_rescored_chunks": resc, **props}
    s.close()
    print(json.dumps(rec), flush=True)
    return sample_ok and all(props.values())


def main_big(wiki_mb, utf8_mb):
    import threading
    t_start = time.time()

    def heartbeat():  # (generating gigabytes of text takes minutes: say so)
        while True:
            time.sleep(45)
            print("  ... %.0f s" % (time.time() - t_start), file=sys.stderr, flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    ok = True
    if wiki_mb > 0:
        src, _ = synth.wiki_like(2_000_000, seed=20201120)
        tokens = synth.build_vocab(src, 32000)
        corpus, off = synth.wiki_like(int(wiki_mb * 1e6), seed=20201120)
        ok &= big("wiki", "config 2 per-GPU share: wiki-like %.0f MB, V=32k" % wiki_mb, tokens, corpus, off)
        del corpus
    if utf8_mb > 0:
        corpus, off = synth.random_utf8(int(utf8_mb * 1e6), seed=355)
        tokens = synth.sample_vocab(corpus[: 50_000_000], 128000)
        ok &= big("utf8", "config 4: random UTF-8 %.0f MB, V=128k sampled from the corpus" % utf8_mb, tokens, corpus, off)
    # config 5's corpus (CJK-heavy, V = 32 k) and the reference annealer's own vocabulary on the kind of text it was made from
    corpus, off = synth.cjk_heavy(400_000_000, seed=5)
    ok &= big("cjk", "config 5 corpus: CJK-heavy 400 MB, V=32k", synth.build_vocab(corpus[:2_000_000], 32000), corpus, off)
    corpus, off = synth.wiki_like(100_000_000, seed=1234, n_words=20_000)
    ok &= big("annealed", "tests/golden/wiki_annealed vocabulary (30,275 tokens, 124,789 trie nodes) on wiki-like 100 MB (seed 1234, 20,000 words)",
              O.load_golden("wiki_annealed")["tokens"], corpus, off)
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


def main():
    if sys.argv[1:2] == ["big"]:
        return main_big(*[float(x) for x in (sys.argv[2:4] + ["2500", "1250"][len(sys.argv) - 2:])])
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
