#!/usr/bin/env python3
"""Two evaluations in flight: two contexts (two candidate vocabularies over the same shard) on two
streams against the same two contexts on one stream.  What an annealer that scores two candidates
at a time would get; not the bench line (one evaluation per step).  Usage: tools/bench_two_in_flight.py [mb] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenizer_amd as ta  # noqa: E402
from tokenizer_amd import synth  # noqa: E402


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    src, _ = synth.wiki_like(2_000_000, seed=20201120)
    tokens = synth.build_vocab(src, 32000)
    corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
    dev = torch.device("cuda", 0)
    scorers, outs = [], []
    for k in range(2):
        s = ta.Scorer(0)
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        if k:  # the second candidate vocabulary: one token fewer
            s.toggle(100, False)
        scorers.append(s)
        outs.append(torch.zeros(len(tokens) + 5, dtype=torch.int64, device=dev))
    ref = [s.score() for s in scorers]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    res = {}
    for name, pick in (("one_stream", lambda k: streams[0]), ("two_streams", lambda k: streams[k])):
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3 if warm else steps):
                for k in range(2):
                    scorers[k].score_async(outs[k].data_ptr(), pick(k).cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res[name + "_ms_per_eval"] = dt * 1e3 / (2 * steps)
        res[name + "_GBps"] = len(corpus) * 2 * steps / dt / 1e9
        for k in range(2):
            got = outs[k].cpu().numpy().view(np.uint64)
            assert np.array_equal(got[:len(tokens) + 3], ref[k][0]) and int(got[len(tokens) + 3]) == ref[k][1], (name, k)
    res["bytes"] = int(len(corpus))
    res["exact"] = True
    print(json.dumps(res))


if __name__ == "__main__":
    main()
