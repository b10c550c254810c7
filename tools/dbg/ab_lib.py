#!/usr/bin/env python3
"""Scan-kernel time of the bench shard under the library TOKSCORE_LIBRARY names (one process per
library: tools/dbg/ab_libs.sh).  Prints ms per asynchronous evaluation and the scan kernel's
average from the library's own HIP events.  Usage: tools/dbg/ab_lib.py [steps] [mb]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import tokenizer_amd as ta
from tokenizer_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mb = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
vocab_src, _ = synth.wiki_like(2_000_000, seed=20201120)
tokens = synth.build_vocab(vocab_src, 32000)
corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
dev = torch.device("cuda", 0)
out = torch.zeros(len(tokens) + 5, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream()
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens))
s.set_corpus(corpus, off)
r = s.score()
for rep in range(3):
    for _ in range(5):
        s.score_async(out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    s.drain_timings()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.score_async(out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = s.drain_timings()
    scan = sum(x[1] for x in t) / max(1, len(t))
    ev = sum(x[0] for x in t) / max(1, len(t))
    print(os.path.basename(ta.LIB_PATH), "ms/step %.4f" % (dt * 1e3 / steps), "scan %.4f eval %.4f (n=%d)" % (scan, ev, len(t)), "total_tokens", int(out[len(tokens) + 3]), flush=True)
