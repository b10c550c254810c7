#!/usr/bin/env python3
"""ms per asynchronous evaluation of the bench shard, HIP events on and off (same process, alternating).
Usage: tools/dbg/ab_steps.py [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import tokenizer_amd as ta
from tokenizer_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(100_000_000, seed=20201120)
dev = torch.device("cuda", 0)
out = torch.zeros(len(tokens) + 5, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream()
scorers = {}
for timing in ("1", "0"):
    os.environ["TOKSCORE_TIMING"] = timing
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    s.score()
    scorers[timing] = s
for rep in range(3):
    for timing, s in scorers.items():
        for _ in range(5):
            s.score_async(out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.score_async(out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if timing == "1":
            s.drain_timings()
        print("timing", timing, "ms/step %.4f" % (dt * 1e3 / steps), flush=True)
