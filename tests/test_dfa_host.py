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


@pytest.mark.parametrize("case", O.golden_cases())
def test_automaton_matches_reference_golden(case):
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
def test_automaton_matches_oracle_on_synthetic(kind):
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


def test_automaton_refuses_what_it_cannot_hold():
    corpus, off = synth.random_utf8(300_000, seed=355)
    tokens = synth.sample_vocab(corpus, 4000, seed=355)
    s = _host_scorer(tokens)
    with pytest.raises(ta.TokscoreError) as e:
        s.hostdfa_check(corpus, off, max_slots=20_000)
    assert e.value.status == ta.E_VOCAB
    s.close()
