"""The scan automaton's builder (tokenizer_amd/csrc/dfa.cpp) on the CPU: the image run over the
documents on the host (tokscore_hostdfa_check, the same step rule as the kernel) must give what the
compiled reference's goldens and the oracle give.  No GPU involved."""
import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth


def _host_scorer(tokens):
    s = ta.Scorer(host_only=True)
    s.set_vocab(*synth.pack_vocab(tokens))
    return s


@pytest.fixture(params=["flattened", "failure"])
def form(request, monkeypatch):
    """Both forms of the automaton (dfa.h): every byte taken at once (virtual entries), or f followed state by state."""
    monkeypatch.setenv("TOKSCORE_HOSTDFA_FAIL", "1" if request.param == "failure" else "0")
    return request.param


@pytest.mark.parametrize("case", O.golden_cases())
def test_automaton_matches_reference_golden(case, form):
    g = O.load_golden(case)
    corpus, off = synth.pack_docs(g["docs"])
    if any(b"\x00" in d for d in g["docs"]):
        pytest.skip("documents with NUL bytes never reach the automaton (the tile scan serves them)")
    s = _host_scorer(g["tokens"])
    try:
        counts, tt, stats = s.hostdfa_check(corpus, off, max_slots=8 << 20)
    except ta.TokscoreError as e:
        assert e.status == ta.E_VOCAB  # too many codes / slots: the other scan paths serve it
        pytest.skip(str(e))
    exp = g["expected"]
    assert tt == exp["total_tokens"]
    assert np.array_equal(counts, exp["counts_vec"])
    s.close()


@pytest.mark.parametrize("kind", ["wiki", "cjk"])
def test_automaton_matches_oracle_on_synthetic(kind, form):
    if kind == "wiki":
        corpus, off = synth.wiki_like(1_500_000, seed=11)
        tokens = synth.build_vocab(corpus[:600_000], 8000)
    else:
        corpus, off = synth.cjk_heavy(1_000_000, seed=12)
        tokens = synth.build_vocab(corpus[:400_000], 6000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    want = voc.score(corpus, off, threads=4)
    s = _host_scorer(tokens)
    counts, tt, stats = s.hostdfa_check(corpus, off)
    assert tt == want[1]
    assert np.array_equal(counts, want[0])
    assert int(stats[0]) >= int(stats[1]) + int(stats[2])
    # toggles: the automaton is rebuilt from the vocabulary as it is
    rng = np.random.default_rng(3)
    multi = [i for i, t in enumerate(tokens) if len(t) > 1]
    off_tokens = sorted(set(int(x) for x in rng.choice(multi, size=40)))
    for i in off_tokens:
        s.toggle(i, False)
    kept = [t if i not in set(off_tokens) else b"" for i, t in enumerate(tokens)]
    voc2 = O.OracleVocab(*synth.pack_vocab(kept))
    want2 = voc2.score(corpus, off, threads=4)
    counts2, tt2, _ = s.hostdfa_check(corpus, off)
    assert tt2 == want2[1]
    assert np.array_equal(counts2, want2[0])
    s.close()


def test_failure_form_has_no_virtual_entries(monkeypatch):
    # the vocabulary the reference's annealer wrote: 2.8 M virtual entries flattened, none in the failure form, whose
    # rows hold the trie's own edges and at most one failure entry per state
    g = O.load_golden("wiki_annealed")
    corpus, off = synth.pack_docs(g["docs"])
    s = _host_scorer(g["tokens"])
    nodes = int(s.info().trie_nodes)
    monkeypatch.setenv("TOKSCORE_HOSTDFA_FAIL", "0")
    _, _, flat = s.hostdfa_check(corpus, off, max_slots=8 << 20)
    monkeypatch.setenv("TOKSCORE_HOSTDFA_FAIL", "1")
    counts, tt, fail = s.hostdfa_check(corpus, off, max_slots=8 << 20)
    assert int(flat[2]) > 1_000_000 and int(fail[2]) == 0
    assert nodes - 1 <= int(fail[1]) <= 2 * nodes + 256  # edges (+ the root row's entries for bytes without an edge) + failure entries
    assert int(fail[0]) < int(flat[0]) // 8
    assert tt == g["expected"]["total_tokens"] and np.array_equal(counts, g["expected"]["counts_vec"])
    s.close()


def test_automaton_refuses_what_it_cannot_hold():
    corpus, off = synth.random_utf8(300_000, seed=355)
    tokens = synth.sample_vocab(corpus, 4000, seed=355)
    s = _host_scorer(tokens)
    with pytest.raises(ta.TokscoreError) as e:
        s.hostdfa_check(corpus, off, max_slots=20_000)
    assert e.value.status == ta.E_VOCAB
    s.close()


# ---- the search automaton of a batch of strings (tokscore_score_toggles' one-pass search) ----
def _brute_occurrences(strings, text):
    tb = text.tobytes()
    found = set()
    for k, s in enumerate(strings):
        if len(s) < 2:
            continue
        i = tb.find(s)
        while i >= 0:
            found.add((k, i))
            i = tb.find(s, i + 1)
    return found


@pytest.mark.parametrize("seed,ncodes,nstr,alphabet", [(1, 50, 64, 6), (2, 12, 200, 3), (3, 254, 64, 40), (4, 5, 30, 2)])
def test_search_automaton_finds_every_occurrence(seed, ncodes, nstr, alphabet):
    """Aho-Corasick in the scan's LDS-array layout (dfa.h: BuildSearchImage), stepped on the host as the
    kernel steps it, against str.find: overlapping strings, strings that are suffixes / prefixes of each
    other, duplicates, one-code strings (left out by contract), separators in the text."""
    rng = np.random.default_rng(seed)
    text = rng.integers(1, min(ncodes, alphabet + 1), size=20_000).astype(np.uint8)
    text[rng.integers(0, len(text), size=200)] = 0  # document separators
    strings = []
    for _ in range(nstr):
        if rng.random() < 0.5:  # drawn from the text: occurs
            L = int(rng.integers(1, 9))
            at = int(rng.integers(0, len(text) - L))
            s = text[at:at + L].tobytes()
            if 0 in s:
                s = s.replace(b"\x00", b"\x01")
        else:
            s = rng.integers(1, min(ncodes, alphabet + 1), size=int(rng.integers(1, 7))).astype(np.uint8).tobytes()
        strings.append(s)
    strings.append(strings[0])  # a duplicate: both are reported
    strings.append(strings[1][:1] + strings[1])  # an extension to the left: the shorter one ends where it ends
    pairs, (slots, states, entries) = ta.hostsearch_check(strings, ncodes, text)
    got = set(map(tuple, pairs.tolist()))
    assert len(got) == len(pairs)  # nothing twice
    assert got == _brute_occurrences(strings, text)
    assert slots <= 40832 and states >= 1


def test_search_automaton_that_does_not_fit_is_refused():
    rng = np.random.default_rng(9)
    strings = [rng.integers(1, 200, size=12).astype(np.uint8).tobytes() for _ in range(400)]
    with pytest.raises(ta.TokscoreError) as e:
        ta.hostsearch_check(strings, 201, np.zeros(16, np.uint8), max_slots=4096)
    assert e.value.status == ta.E_CAPACITY
