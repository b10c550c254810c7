#!/usr/bin/env python3
"""Tuning sweep on one GPU: same resident workload as bench.py, several launch plans
(lane segment bytes, waves per workgroup, LDS trie slots, chunk bytes)."""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
    tokens = synth.build_vocab(vocab_src, int(os.environ.get("SWEEP_VOCAB", "32000")))
    corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
    tb, to = synth.pack_vocab(tokens)
    ref = None
    configs = []
    if not sys.argv[2:]:
        for seg, waves in itertools.product(("68", "132", "260"), ("16", "8", "4")):
            configs.append({"TOKSCORE_SEG": seg, "TOKSCORE_WAVES": waves})
    for extra in sys.argv[2:]:
        configs.append(dict(kv.split("=") for kv in extra.split(",")))
    for cfg in configs:
        for k in ("TOKSCORE_SEG", "TOKSCORE_WAVES", "TOKSCORE_LDS_SLOTS", "TOKSCORE_CHUNK", "TOKSCORE_SCAN", "TOKSCORE_BLOCKS_PER_CU", "TOKSCORE_PARK"):
            os.environ.pop(k, None)
        os.environ.update(cfg)
        s = ta.Scorer(0)
        s.set_vocab(tb, to)
        s.set_corpus(corpus, off)
        out = s.score()
        if ref is None:
            ref = out
        ok = np.array_equal(out[0], ref[0]) and out[1:] == ref[1:]
        for _ in range(3):
            s.score()
        s.drain_timings()
        for _ in range(10):
            s.score()
        tm = np.array(s.drain_timings())[-10:]
        info = s.info()
        ev, sc, fx, hs = tm.mean(axis=0)
        print("%-60s ok=%d lds_slots=%d chunks=%d eval %.3f ms scan %.3f fix %.3f hist %.3f -> scan %.1f GB/s eval %.1f GB/s"
              % (cfg, ok, info.trie_lds_slots, info.chunks, ev, sc, fx, hs, len(corpus) / sc / 1e6, len(corpus) / ev / 1e6),
              flush=True)
        s.close()


if __name__ == "__main__":
    main()
