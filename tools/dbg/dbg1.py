import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_ffi as O, tokenizer_amd as ta
from tokenizer_amd import synth
corpus, off = synth.wiki_like(3_000_000, seed=21)
tokens0 = synth.build_vocab(corpus[:600_000], 6000)
special = bytes(corpus[1_234_560:1_234_572]).lower()
print("special", special)
def run(tokens, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens)); s.set_corpus(corpus, off)
    out = s.score(); info = s.info()
    s.close()
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return out, info
for name, tokens in (("plain", tokens0), ("special", tokens0 + [special])):
    exp = O.OracleVocab(*synth.pack_vocab(tokens)).score(corpus, off, threads=8)
    for env in ({"TOKSCORE_SCAN": "stream"}, {"TOKSCORE_SCAN": "stream", "TOKSCORE_WAVES": "8"}, {"TOKSCORE_SCAN": "stream", "TOKSCORE_STREAM": "1024"}, {"TOKSCORE_SCAN": "stream", "TOKSCORE_STREAM": "100000000"}):
        (c, tt, tb), info = run(tokens, env)
        bad = np.nonzero(c != exp[0])[0]
        print(name, env, "path", info.scan_path, "chunks", info.chunks, "tt", tt, exp[1], "bad ids", len(bad), bad[:8], c[bad[:8]], exp[0][bad[:8]], flush=True)
