#!/usr/bin/env python3
"""Annealing moves/s on the host (moves/sec (i)): candidates of a wiki-like
slice (GetTokens, windows of up to 10 code points), TokenGenerator(pref 30,000), Generate --
moves = gen_cnt increments / wall time, as the reference counts them (TokenGenerator.cpp:157,184).
Usage: tools/bench_anneal.py [mb] [passes] [threads ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenizer_amd import annealing as A  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    threads = [int(x) for x in sys.argv[3:]] or [1, min(16, os.cpu_count() or 1)]
    corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
    t0 = time.perf_counter()
    toks = A.GetTokens(corpus, off, 10)
    build_s = time.perf_counter() - t0
    n = len(toks)
    arrays = toks.arrays()
    out = {"metric": "annealing moves/s (host DAG loop)", "corpus_bytes": int(len(corpus)), "max_len": 10,
           "candidates": n, "get_tokens_seconds": build_s, "pref_token_count": 30000, "passes": passes + 1, "runs": []}
    for th in threads:
        gen = A.TokenGenerator(A.Tokens.from_arrays(*arrays), 30000, th if th > 1 else 16)
        t0 = time.perf_counter()
        states = gen.Generate(passes, threads=th, seed=5)
        dt = time.perf_counter() - t0
        st = states[-1]
        out["runs"].append({"threads": th, "seconds": dt, "moves": int(st.gen_cnt), "moves_per_s": st.gen_cnt / dt,
                            "enabled_cnt": int(st.enabled_cnt), "score": st.score})
        gen.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
