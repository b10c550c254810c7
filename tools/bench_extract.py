#!/usr/bin/env python3
"""Row f3: candidate extraction (GetTokens: all windows of up to max_cp code points, counted, in trie
pre-order with both parents) on the GPU against the host builder of libtokanneal.so.
Usage: tools/bench_extract.py [mb] [max_cp] [host_mb]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import annealing as A  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    max_cp = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    host_mb = float(sys.argv[3]) if len(sys.argv) > 3 else min(mb, 4.0)
    corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
    s = ta.Scorer(0)
    s.set_corpus(corpus, off)
    n = ta.C.c_uint64()
    pos = ta.C.c_uint64()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        s._check(s._lib.tokscore_extract_candidates(s._h, max_cp, ta.C.byref(n), ta.C.byref(pos)))
        times.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    arrays = s.extract_candidates(max_cp)
    with_copy = time.perf_counter() - t0
    s.close()
    out = {"metric": "candidate extraction (GetTokens) GB/s", "corpus_bytes": int(len(corpus)), "max_cp": max_cp,
           "code_points": int(pos.value), "candidates": int(n.value), "gpu_seconds": min(times),
           "gpu_GBps": len(corpus) / min(times) / 1e9, "gpu_seconds_with_copy_to_host": with_copy}
    del arrays
    hc, ho = synth.wiki_like(int(host_mb * 1e6), seed=20201120)
    t0 = time.perf_counter()
    toks = A.GetTokens(hc, ho, max_cp)
    dt = time.perf_counter() - t0
    out.update({"host_bytes": int(len(hc)), "host_candidates": len(toks), "host_seconds": dt, "host_GBps": len(hc) / dt / 1e9,
                "host_threads": 1})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
