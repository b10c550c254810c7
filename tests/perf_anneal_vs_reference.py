#!/usr/bin/env python3
"""Not a test (pytest does not collect it): times the COMPILED REFERENCE's annealer
(oracle/_ref/ref_harness `anneal`: GetTokens -> TokenGenerator -> Generate, hardware_concurrency
workers, batch = 30 as the harness passes it) on the corpus tools/bench_anneal.py uses, so the two
moves/s figures sit side by side.  Build container only (needs oracle/_ref).
Usage: python tests/perf_anneal_vs_reference.py [mb] [passes] [batch]"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tokenizer_amd import synth  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)  # default: the reference's (TokenGenerator.h:49)
    corpus, off = synth.wiki_like(int(mb * 1e6), seed=20201120)
    b = corpus.tobytes()
    texts = [b[int(off[i]):int(off[i + 1])].decode("utf-8") for i in range(len(off) - 1)]
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "data")
        os.makedirs(data)
        with open(os.path.join(data, "a.json"), "w", encoding="utf-8") as f:
            json.dump([{"id": str(i), "text": t, "title": ""} for i, t in enumerate(texts)], f, ensure_ascii=False)
        r = subprocess.run([HARNESS, "anneal", data, "10", "30000", str(batch), str(passes), os.path.join(tmp, "t.json")],
                           cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, text=True)
    j = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{")][-1])
    j["moves_per_s"] = j["gen_cnt"] / j["generate_seconds"]
    j["batch"] = batch
    print(json.dumps(j))


if __name__ == "__main__":
    main()
