"""The CPU oracle (oracle/tokscore_oracle.c) against the goldens the COMPILED
REFERENCE produced (tests/golden/make_golden.py).  This is what pins the
oracle; everything GPU-side is then checked against the oracle."""
import numpy as np
import pytest

import oracle_ffi as O
from tokenizer_amd import synth


@pytest.mark.parametrize("case", O.golden_cases())
def test_oracle_matches_reference_golden(case):
    g = O.load_golden(case)
    tb, to = synth.pack_vocab(g["tokens"])
    voc = O.OracleVocab(tb, to)
    exp = g["expected"]
    corpus, off = synth.pack_docs(g["docs"])
    for threads in (1, 3):
        counts, tt, tby = voc.score(corpus, off, threads=threads)
        assert tt == exp["total_tokens"]
        assert tby == exp["total_bytes"]
        assert np.array_equal(counts, exp["counts_vec"])
    assert int(counts[0]) == len(g["docs"]) and int(counts[1]) == len(g["docs"])
    assert int(counts[2:].sum()) == exp["total_tokens"]
    per_doc = [len(voc.tokenize(d)) - 2 for d in g["docs"]]
    assert per_doc == exp["doc_tokens"]
    if "ids" in exp:
        for d, ids in zip(g["docs"], exp["ids"]):
            assert voc.tokenize(d) == ids


def test_oracle_get_id_and_max_len():
    g = O.load_golden("dup_tokens")
    tb, to = synth.pack_vocab(g["tokens"])
    voc = O.OracleVocab(tb, to)
    # tokens: a b ab a ab c b abc abc -> first id wins (SolutionFile.cpp:56)
    assert voc.get_id(b"a") == 2 and voc.get_id(b"b") == 3 and voc.get_id(b"ab") == 4
    assert voc.get_id(b"c") == 7 and voc.get_id(b"abc") == 9 and voc.get_id(b"zz") == -1
    assert voc.max_len == 3
