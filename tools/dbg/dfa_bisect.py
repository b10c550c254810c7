"""Which part of the automaton scan disagrees with the oracle: TOKSCORE_ASM 0/1, LDS sizes, chains (debugging aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth
corpus, off = synth.wiki_like(3_000_000, seed=21)
tokens = synth.build_vocab(corpus[:800_000], 12000)
voc = O.OracleVocab(*synth.pack_vocab(tokens))
want = voc.score(corpus, off, threads=8)
for asm in ("0", "1"):
    for lds in sys.argv[1].split(","):
        for ch in sys.argv[2].split(","):
            os.environ.update(TOKSCORE_ASM=asm, TOKSCORE_DFA_LDS=lds, TOKSCORE_DFA_CHAINS=ch)
            s = ta.Scorer(0)
            s.set_vocab(*synth.pack_vocab(tokens))
            s.set_corpus(corpus, off)
            out = []
            for it in range(3):
                got = s.score()
                d = got[0].astype(np.int64) - want[0].astype(np.int64)
                out.append((int(got[1]) - int(want[1]), int(np.abs(d).sum()), int((d != 0).sum())))
            print("asm", asm, "lds", lds, "chains", ch, "total diff / sum |diff| / ids:", out, flush=True)
            s.close()
