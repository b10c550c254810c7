"""f1 on the GPU: tokscore_count_occurrences against the compiled reference's candidate list
(golden) and against the CPU oracle on synthetic corpora.  Bit-exact."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gpu_counts(names, docs_or_packed, max_cp):
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(names))
        packed = docs_or_packed if isinstance(docs_or_packed, tuple) else synth.pack_docs(docs_or_packed)
        s.set_corpus(*packed)
        out = s.count_occurrences(max_cp)
        again = s.count_occurrences(max_cp)
        assert np.array_equal(out[0], again[0]) and out[1] == again[1]
    finally:
        s.close()
    return out


def test_counts_equal_reference_uses_mixed():
    g = O.load_cands("cands_mixed")
    counts, positions = _gpu_counts(g["names"], g["docs"], g["max_len"])
    bad = np.nonzero(counts[2:2 + len(g["names"])] != g["uses"])[0]
    assert bad.size == 0, (bad[:10], [g["names"][i] for i in bad[:10]], counts[2 + bad[:10]], g["uses"][bad[:10]])
    assert positions == sum(len(d.decode("utf-8")) for d in g["docs"])
    assert counts[0] == counts[1] == counts[-1] == 0


def test_counts_equal_reference_uses_dag_tiny():
    j = json.load(open(os.path.join(HERE, "golden", "dag_tiny.json"), encoding="utf-8"))
    names = [t["name"].encode("utf-8") for t in j["tokens"]]
    uses = np.array([t["uses"] for t in j["tokens"]], dtype=np.uint64)
    docs = ["abracadabra abracadabra", "the cat sat on the mat", "日本語の本 日本"]
    counts, _ = _gpu_counts(names, [d.encode("utf-8") for d in docs], j["max_len"])
    assert np.array_equal(counts[2:2 + len(names)], uses)


@pytest.mark.parametrize("kind,max_cp,V", [("wiki", 10, 8000), ("cjk", 6, 6000), ("utf8", 4, 40000), ("wiki", 2, 3000)])
def test_counts_equal_oracle_on_synthetic(kind, max_cp, V):
    if kind == "wiki":
        corpus, off = synth.wiki_like(1_500_000, seed=3)
        tokens = synth.build_vocab(corpus[:400_000], V, lowercase=False)
    elif kind == "cjk":
        corpus, off = synth.cjk_heavy(1_000_000, seed=4)
        tokens = synth.build_vocab(corpus[:300_000], V)
    else:
        corpus, off = synth.random_utf8(1_500_000, seed=6, doc_lo=64, doc_hi=60_000)
        tokens = synth.build_vocab(corpus[:600_000], V)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    exp = voc.count_occurrences(corpus, off, max_cp, threads=8)
    got = _gpu_counts(tokens, (corpus, off), max_cp)
    bad = np.nonzero(got[0] != exp[0])[0]
    assert bad.size == 0, (kind, bad[:10], got[0][bad[:10]], exp[0][bad[:10]])
    assert got[1] == exp[1]
    assert int(exp[0].sum()) > exp[1] // 2, "the workload is meant to hit often"


def test_edge_cases():
    # empty corpus, empty documents, a document shorter than a token, duplicates, windows at document ends
    names = [b"ab", b"b", b"ab", "語".encode(), b"abc"]
    for docs in ([], [b""], [b"", b"ab", b""], [b"a", b"b", b"ab", b"abab"], ["語語a".encode(), b"b" * 5000]):
        voc = O.OracleVocab(*synth.pack_vocab(names))
        corpus, off = synth.pack_docs(docs)
        exp = voc.count_occurrences(corpus, off, 3)
        got = _gpu_counts(names, (corpus, off), 3)
        assert np.array_equal(got[0], exp[0]) and got[1] == exp[1], docs


def test_many_tiny_documents():
    # hundreds of document starts inside one 1 KiB tile: the bitmap is built in several rounds
    rng = np.random.default_rng(11)
    alphabet = [b"a", b"b", b"ab", "é".encode(), "語".encode(), b" "]
    docs = []
    for _ in range(60_000):
        n = int(rng.integers(0, 4))
        docs.append(b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n)))
    names = [b"a", b"b", b"ab", b"ba", b"aab", "é".encode(), "語".encode(), "é語".encode(), b" a", b"a b", b"abab"]
    voc = O.OracleVocab(*synth.pack_vocab(names))
    corpus, off = synth.pack_docs(docs)
    exp = voc.count_occurrences(corpus, off, 4)
    got = _gpu_counts(names, (corpus, off), 4)
    assert np.array_equal(got[0], exp[0]) and got[1] == exp[1]


def test_long_tokens_take_the_unprefetched_tile_path():
    # a 1,500-byte candidate: the staged halo no longer fits the register prefetch
    rng = np.random.default_rng(12)
    base = bytes(rng.integers(97, 123, 1500).astype(np.uint8))
    docs = [base + b" tail", b"x" + base, base[:700], b"short", base + base]
    names = [base, base[:700], base[:3], b"x" + base[:10], b"tail"]
    voc = O.OracleVocab(*synth.pack_vocab(names))
    corpus, off = synth.pack_docs(docs)
    exp = voc.count_occurrences(corpus, off, 1600)
    got = _gpu_counts(names, (corpus, off), 1600)
    assert np.array_equal(got[0], exp[0]) and got[1] == exp[1]
    assert got[0][2] == 4  # the long one: documents 0 and 1, twice in document 4


def test_one_workgroup_sees_more_than_a_16_bit_row_holds(monkeypatch):
    # the slab rows of the slot histogram are 16 bits wide; a slot with more than 32,767 of one workgroup's
    # entries is booked directly (kernels_stream.hip hist_slots_kernel)
    monkeypatch.setenv("TOKSCORE_SLAB_BLOCKS", "1")
    names = [b"a", b"b", b"ab", b"ba", b"abab"]
    docs = [b"ab" * 50_000, b"b" * 40_000, b"ab"]
    voc = O.OracleVocab(*synth.pack_vocab(names))
    corpus, off = synth.pack_docs(docs)
    exp = voc.count_occurrences(corpus, off, 4)
    got = _gpu_counts(names, (corpus, off), 4)
    assert int(exp[0].max()) > 40_000
    assert np.array_equal(got[0], exp[0]) and got[1] == exp[1]


@pytest.mark.gpu
@pytest.mark.parametrize("longest,extra", [(24, 300), (45, 300), (60, 40), (90, 40)])
def test_longer_tokens_take_the_wider_rings(longest, extra):
    """Tokens of up to 24 / 45 / 90 bytes: 64- and 128-byte lane rings and other round lengths in the
    assembly loops of the scan and of the counting kernel (the bench vocabulary only sees 32-byte
    rings with rounds of 7)."""
    corpus, off = synth.wiki_like(1_200_000, seed=41)
    tokens = synth.build_vocab(corpus[:300_000], 3000)
    rng = np.random.default_rng(longest)
    raw = bytes(corpus)
    for _ in range(extra):  # strings cut from the corpus, so that they do occur
        n = int(rng.integers(12, longest + 1))
        at = int(rng.integers(0, len(raw) - n))
        piece = raw[at:at + n]
        if b"\0" not in piece:
            tokens.append(piece.lower())
    tokens = list(dict.fromkeys(tokens))
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        exp = voc.score(corpus, off, threads=8)
        got = s.score()
        assert np.array_equal(got[0], exp[0]) and got[1:] == exp[1:]
        if longest <= 60:
            assert s.info().scan_path == 2
        expc = voc.count_occurrences(corpus, off, longest, threads=8)
        gotc = s.count_occurrences(longest)
        assert np.array_equal(gotc[0], expc[0]) and gotc[1] == expc[1]
    finally:
        s.close()


_ASM0_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth
corpus, off = synth.wiki_like(1_000_000, seed=43)
tokens = synth.build_vocab(corpus[:300_000], 3000, lowercase=False)
voc = O.OracleVocab(*synth.pack_vocab(tokens))
s = ta.Scorer(0)
s.set_vocab(*synth.pack_vocab(tokens))
s.set_corpus(corpus, off)
exp = voc.score(corpus, off, threads=8)
for _ in range(2):
    got = s.score()
    assert np.array_equal(got[0], exp[0]) and got[1:] == exp[1:], "score"
assert s.info().scan_path == 2
expc = voc.count_occurrences(corpus, off, 8, threads=8)
gotc = s.count_occurrences(8)
assert np.array_equal(gotc[0], expc[0]) and gotc[1] == expc[1], "count"
s.close()
print("asm0 ok")
"""


def test_cpp_loops_behind_tokscore_asm_0():
    """TOKSCORE_ASM=0 (read once per process, hence a child process: one, for this test only) puts the
    C++ loops in the place of the assembly ones in scan_ranges_kernel and count_ends_kernel; same
    results."""
    import subprocess
    import sys
    env = dict(os.environ, TOKSCORE_ASM="0", TOKSCORE_SCAN="stream")
    r = subprocess.run([sys.executable, "-c", _ASM0_CHILD, os.path.dirname(HERE), HERE], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "asm0 ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
