#!/usr/bin/env python3
"""f1 measurement: closed-vocabulary all-occurrence counting on the bench shard
(100 MB wiki-like, V = 32,000, windows of up to 10 code points), host wall time of the
synchronous entry point, with the CPU oracle on a bounded sample beside it.
Usage: tools/bench_count.py [mb] [max_cp] [--no-cpu]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    mb = float(args[0]) if args else 100.0
    max_cp = int(args[1]) if len(args) > 1 else 10
    src, _ = synth.wiki_like(2_000_000, seed=20201120)
    tokens = synth.build_vocab(src, int(os.environ.get("COUNT_VOCAB", "32000")))
    corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    counts, pos = s.count_occurrences(max_cp)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        s.count_occurrences(max_cp)
        ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) * 1e3
    tm = s.last_timing()  # [1] the counting kernel, [3] histogram + reduction (the stream-image path records them)
    rec = {"metric": "corpus GB/s, closed-vocabulary all-occurrence counting (f1)", "value": len(corpus) / ms / 1e6,
           "unit": "GB/s", "ms_per_call_host_wall": ms, "bytes": int(len(corpus)), "vocab": len(tokens), "max_cp": max_cp,
           "positions": pos, "increments": int(counts.sum())}
    if tm[1] > 0:
        rec["kernel"] = "count_ends_kernel"
        rec["kernel_ms"] = tm[1]
        rec["hist_reduce_ms"] = tm[3]
        # 1 algorithmic byte per corpus byte, one launch = the whole shard
        rec["roofline"] = {"bound": "hbm", "achieved": len(corpus) / tm[1] / 1e6, "peak": 8000.0, "unit": "GB/s",
                           "frac": len(corpus) / tm[1] / 1e6 / 8000.0, "traffic": None}

    if "--no-cpu" not in sys.argv:
        import oracle_ffi as O
        voc = O.OracleVocab(*synth.pack_vocab(tokens))
        d_end = int(np.searchsorted(off, 8_000_000, side="right")) - 1
        sub = off[:d_end + 1]
        t0 = time.time()
        exp = voc.count_occurrences(corpus[:int(sub[-1])], sub, max_cp, threads=16)
        dt = time.time() - t0
        rec["cpu_baseline"] = {"value": int(sub[-1]) / dt / 1e9, "unit": "GB/s", "cores": 16, "kind": "port",
                               "sample": "first %d documents (%d bytes), oracle/tokscore_oracle.c ora_count_occurrences" % (d_end, int(sub[-1]))}
        s2 = ta.Scorer(0)
        s2.set_vocab(*synth.pack_vocab(tokens))
        s2.set_corpus(corpus[:int(sub[-1])], sub)
        got = s2.count_occurrences(max_cp)
        rec["sample_parity"] = bool(np.array_equal(got[0], exp[0]) and got[1] == exp[1])
        s2.close()
    print(json.dumps(rec))
    s.close()


if __name__ == "__main__":
    main()
