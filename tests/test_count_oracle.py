"""f1: closed-vocabulary all-occurrence counting.  The oracle
(oracle/tokscore_oracle.c: ora_count_occurrences) against what the compiled reference's
GetTokens produced: every candidate's `uses` is its number of occurrences as a code-point
substring of at most max_len code points (GetTokens.cpp:21-27, Trie.cpp:82,114-128)."""
import json
import os

import numpy as np

import oracle_ffi as O
from tokenizer_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))

DAG_TINY_DOCS = ["abracadabra abracadabra", "the cat sat on the mat", "日本語の本 日本"]  # tests/golden/make_golden.py (12)


def _check(names, uses, docs, max_len):
    voc = O.OracleVocab(*synth.pack_vocab(names))
    corpus, off = synth.pack_docs(docs)
    counts, positions = voc.count_occurrences(corpus, off, max_len)
    assert np.array_equal(counts[2:2 + len(names)], uses)
    assert counts[0] == counts[1] == counts[len(names) + 2] == 0
    assert positions == sum(len(d.decode("utf-8")) for d in docs)
    # threads do not change the result
    c2, p2 = voc.count_occurrences(corpus, off, max_len, threads=3)
    assert np.array_equal(c2, counts) and p2 == positions
    return counts


def test_oracle_counts_equal_reference_uses_mixed():
    g = O.load_cands("cands_mixed")
    assert len(g["names"]) > 5000 and len(set(g["names"])) == len(g["names"])
    _check(g["names"], g["uses"], g["docs"], g["max_len"])


def test_oracle_counts_equal_reference_uses_dag_tiny():
    j = json.load(open(os.path.join(HERE, "golden", "dag_tiny.json"), encoding="utf-8"))
    names = [t["name"].encode("utf-8") for t in j["tokens"]]
    uses = np.array([t["uses"] for t in j["tokens"]], dtype=np.uint64)
    _check(names, uses, [d.encode("utf-8") for d in DAG_TINY_DOCS], j["max_len"])


def test_window_is_in_code_points_and_not_lowercased():
    voc = O.OracleVocab(*synth.pack_vocab([b"ab", b"AB", "日本".encode(), "日本語".encode(), b"b"]))
    corpus, off = synth.pack_docs([b"abABab", "日本語日本".encode(), b"a", b"b"])
    c, pos = voc.count_occurrences(corpus, off, 2)
    assert [int(x) for x in c[2:7]] == [2, 1, 2, 0, 3]  # the 3-code-point token is outside a 2-code-point window
    assert pos == 6 + 5 + 1 + 1
    c3, _ = voc.count_occurrences(corpus, off, 3)
    assert int(c3[5]) == 1
    # occurrences never span two documents
    corpus, off = synth.pack_docs([b"a", b"b"])
    c, _ = voc.count_occurrences(corpus, off, 2)
    assert int(c[2]) == 0
