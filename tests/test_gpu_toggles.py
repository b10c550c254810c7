"""tokscore_score_toggles: K independent annealing moves judged in one call.  Each delta must equal
total_tokens(vocabulary with the move) - total_tokens(vocabulary as it is) from the CPU oracle, and
the call must leave the scorer's state as it found it."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_total(tokens, enabled, corpus, off):
    # reference semantics of a disabled entry: a string that can never match
    sub = [t if enabled[k] else b"\x00DISABLED" + str(k).encode() for k, t in enumerate(tokens)]
    return O.OracleVocab(*synth.pack_vocab(sub)).score(corpus, off, threads=8)[1]


def _check_moves(s, tokens, enabled, corpus, off, idx, en):
    base = _oracle_total(tokens, enabled, corpus, off)
    got = s.score_toggles(idx, en)
    assert got.dtype == np.int64 and got.shape == (len(idx),)
    for k, (i, e) in enumerate(zip(idx, en)):
        moved = enabled.copy()
        moved[i] = bool(e)
        exp = _oracle_total(tokens, moved, corpus, off) - base
        assert int(got[k]) == exp, (k, i, tokens[i], e, int(got[k]), exp)
    return got


@pytest.mark.parametrize("env", [{}, {"TOKSCORE_SEARCH": "0"},
                                 {"TOKSCORE_SEARCH": "0", "TOKSCORE_TOGGLE_ITEMS": "2000", "TOKSCORE_TOGGLE_LIST": "4"}, {"TOKSCORE_SCAN": "tile"},
                                 {"TOKSCORE_NOMARKRANGES": "1"}, {"TOKSCORE_SEARCH_SLOTS": "1024"}, {"TOKSCORE_SEARCH_MOVES": "3"},
                                 {"TOKSCORE_DFA_CHAINS": "1"}, {"TOKSCORE_DFA_CHAINS": "4"}, {"TOKSCORE_LOCAL_BOTH": "1"},
                                 {"TOKSCORE_DFA_FORM": "2"}])
def test_deltas_equal_oracle_differences(env, monkeypatch):
    """env: the moves' strings found by one pass of a search automaton over the corpus (the default where the
    automaton scan has a plan) / through the occurrence index and a search per (move, chunk): one batch / several
    batches (a small budget of pairs) whose item lists start too short and have to be provided again / the
    move-by-move fallback on a scan path without the batched kernel / token starts and tokens per stream always
    from the walk over streams (after a full evaluation they come from its ranges otherwise) / the search's
    (move, stream) table too small at first / three moves per pass / the search loop for 1 and 4 ranges per lane / both chains walked (by default the
    chain of the vocabulary as it is comes from the token-start map) / full evaluations on the automaton's failure
    form (the search runs on that plan's transposed corpus, four ranges per lane)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    corpus, off = synth.wiki_like(1_500_000, seed=31)
    tokens = synth.build_vocab(corpus[:500_000], 5000)
    tokens.append(tokens[100])                       # a duplicate string: disabling one copy changes no boundary
    tokens.append(b"The")                            # never matches the lower-cased corpus
    tokens.append(bytes(corpus[700_000:700_014]).lower())  # rare and long
    V = len(tokens)
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        base = s.score()
        assert s.info().dfa_active == 1 and s.info().base_scan_path == 0  # (the automaton scan needs no plan; moves do: below)
        enabled = np.ones(V, dtype=bool)
        rng = np.random.default_rng(7)
        multi = [i for i, t in enumerate(tokens) if len(t) >= 2]
        single = [i for i, t in enumerate(tokens) if len(t) == 1]
        idx = [int(i) for i in rng.choice(multi, size=10, replace=False)]
        idx += [100, V - 3, V - 2, V - 1, single[0], tokens.index(b" ") if b" " in tokens else single[-1]]
        idx += [idx[0]]                               # the same move twice: independent, same answer
        en = [0] * len(idx)
        en[3] = 1                                     # enabling what is enabled: no move
        got = _check_moves(s, tokens, enabled, corpus, off, idx, en)
        assert got[3] == 0 and got[-1] == got[0]
        assert s.info().base_scan_path == (1 if env.get("TOKSCORE_SCAN") == "tile" else 2)
        assert (s.info().toggle_streams > 0) == (env.get("TOKSCORE_SCAN") != "tile"), "which path served the call"
        by_search = env.get("TOKSCORE_SCAN") != "tile" and env.get("TOKSCORE_SEARCH") != "0"
        assert (s.info().search_passes > 0) == by_search, "which path served the call"
        if "TOKSCORE_SEARCH_MOVES" in env:
            assert s.info().search_passes >= 4
        assert any(int(d) != 0 for d in got)
        # nothing was changed by asking
        again = s.score(incremental=True)
        assert np.array_equal(again[0], base[0]) and again[1:] == base[1:]
        # moves against a vocabulary with entries disabled (two of them queued, not evaluated yet)
        off_now = idx[:4]
        for i in off_now[:2]:
            s.toggle(i, False)
            enabled[i] = False
        s.score(incremental=True)
        for i in off_now[2:]:
            s.toggle(i, False)
            enabled[i] = False
        idx2 = off_now + [int(i) for i in rng.choice(multi, size=4, replace=False)]
        en2 = [1, 1, 1, 0] + [0 if enabled[i] else 1 for i in idx2[4:]]
        _check_moves(s, tokens, enabled, corpus, off, idx2, en2)
        after = s.score()
        assert after[1] == _oracle_total(tokens, enabled, corpus, off)
    finally:
        s.close()


@pytest.mark.parametrize("forced", [False, True])
def test_deltas_on_cjk_and_many_moves(forced, monkeypatch):
    """forced: the automaton scan's plan although its policy would not take this vocabulary -- the moves then go
    through the one-pass search (multi-byte code points, 100+ codes); otherwise through the occurrence index."""
    if forced:
        monkeypatch.setenv("TOKSCORE_DFA_MAX_COLD_PPM", "1000000")
        monkeypatch.setenv("TOKSCORE_DFA_MAX_VIRTUAL_PPM", "1000000")
    else:  # (no automaton plan at all: the library would take the failure form for this vocabulary, and search with it)
        monkeypatch.setenv("TOKSCORE_DFA_FORM", "1")
    corpus, off = synth.cjk_heavy(400_000, seed=9)
    tokens = synth.build_vocab(corpus[:200_000], 9000)
    V = len(tokens)
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        base = s.score()
        assert s.info().scan_path == 2 and s.info().dfa_active == (1 if forced else 0)
        enabled = np.ones(V, dtype=bool)
        rng = np.random.default_rng(11)
        multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8")) > 1]
        idx = [int(i) for i in rng.choice(multi, size=64, replace=False)]
        got = s.score_toggles(idx, 0)
        assert s.info().toggle_streams > 0 and (s.info().search_passes > 0) == forced
        # the oracle on a sample of the 64, the rest through the scorer's own full evaluations
        for k in range(0, 64, 8):
            moved = enabled.copy()
            moved[idx[k]] = False
            assert int(got[k]) == _oracle_total(tokens, moved, corpus, off) - base[1], (k, tokens[idx[k]])
        for k in range(64):
            s.toggle(idx[k], False)
            tt = s.score()[1]
            s.toggle(idx[k], True)
            assert int(got[k]) == tt - base[1], (k, tokens[idx[k]])
        assert np.all(got >= 0), "removing a token cannot shorten a greedy segmentation here"
    finally:
        s.close()


def test_empty_and_invalid_arguments():
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab([b"a", b"ab"]))
        s.set_corpus(*synth.pack_docs([b"abab", b"", b"b"]))
        assert s.score_toggles([], []).shape == (0,)
        assert list(s.score_toggles([1, 0, 1], [0, 0, 1])) == [2, 0, 0]  # "ab" off: ab ab -> a ? a ? (UNKNOWN b twice)
        with pytest.raises(ta.TokscoreError):
            s.score_toggles([2], [0])
    finally:
        s.close()


def test_incremental_rescore_on_the_default_cost_model(monkeypatch):
    """Without TOKSCORE_INCR_DIV the library chooses between two partial passes and a full
    evaluation by estimated cost (a small corpus: always the full evaluation): the result is the
    same either way."""
    monkeypatch.delenv("TOKSCORE_INCR_DIV", raising=False)
    corpus, off = synth.wiki_like(800_000, seed=33)
    tokens = synth.build_vocab(corpus[:300_000], 3000)
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        s.score()
        enabled = np.ones(len(tokens), dtype=bool)
        for i in (len(tokens) // 2, len(tokens) // 3):
            enabled[i] = False
            s.toggle(i, False)
            got = s.score(incremental=True)
            assert got[1] == _oracle_total(tokens, enabled, corpus, off)
            assert s.info().rescored_chunks == 0xFFFFFFFF, "a full evaluation was expected to be cheaper here"
    finally:
        s.close()


def test_annealing_like_sequence_takes_token_starts_from_the_ranges():
    """judge K moves, apply one, take the exact score, again: every batch after a full evaluation gets
    its token starts and tokens per stream from that evaluation's ranges (mark_ranges_kernel), with
    entries that the previous toggle's repairs have just moved.  Deltas against the oracle, and the
    applied move's delta against the next exact score."""
    corpus, off = synth.wiki_like(1_200_000, seed=37)
    tokens = synth.build_vocab(corpus[:400_000], 4000)
    V = len(tokens)
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        total = s.score()[1]
        assert s.info().scan_path == 2
        enabled = np.ones(V, dtype=bool)
        rng = np.random.default_rng(9)
        multi = [i for i, t in enumerate(tokens) if len(t) >= 2]
        for batch in range(6):
            idx = [int(i) for i in rng.choice(multi, size=6, replace=False)]
            en = [0 if enabled[i] else 1 for i in idx]
            got = _check_moves(s, tokens, enabled, corpus, off, idx, en)
            k = int(np.argmax(np.abs(got)))
            s.toggle(idx[k], bool(en[k]))
            enabled[idx[k]] = bool(en[k])
            now = s.score()[1]
            assert now - total == int(got[k]), (batch, idx[k], now - total, int(got[k]))
            total = now
    finally:
        s.close()


def test_range_exit_tags_survive_the_wrap_of_their_16_bit_epoch(monkeypatch):
    """The ranges scan tags a repaired range exit with a 16-bit evaluation number (kernels_stream.hip:
    exit_now); when the numbers go round, the words are cleared so that a repair of exactly one cycle ago
    cannot pass for this evaluation's.  Toggles before every evaluation make the ordered fix-up pass run."""
    monkeypatch.setenv("TOKSCORE_SCAN", "stream")
    monkeypatch.setenv("TOKSCORE_DFA", "0")
    monkeypatch.setenv("TOKSCORE_FIX_EPOCH_START", "65530")
    corpus, off = synth.wiki_like(1_200_000, seed=33)
    tokens = synth.build_vocab(corpus[:400_000], 4000)
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        rng = np.random.default_rng(3)
        multi = [i for i, t in enumerate(tokens) if len(t) > 1]
        on = np.ones(len(tokens), dtype=bool)
        for it in range(12):
            for i in rng.choice(multi, size=6, replace=False):
                on[i] = not on[i]
                s.toggle(int(i), bool(on[i]))
            got = s.score()
            assert s.info().base_scan_path == 2
            kept = [t if on[i] else b"" for i, t in enumerate(tokens)]
            want = O.OracleVocab(*synth.pack_vocab(kept)).score(corpus, off, threads=8)
            assert got[1:] == want[1:] and np.array_equal(got[0], want[0]), it
    finally:
        s.close()


@pytest.mark.parametrize("incr_div", [None, "3"])
def test_incremental_parity_under_the_default_cost_model_and_under_the_test_rule(monkeypatch, incr_div):
    """The same toggle / incremental re-score sequence with TOKSCORE_INCR_DIV unset (what users get: the cost
    model of tokscore_score_incremental decides between two partial passes and a full evaluation) and with the
    test suite's plain rule; a corpus large enough that the model has a choice.  Which way it went is in
    tokscore_info.rescored_chunks; the counts are the oracle's either way."""
    if incr_div is None:
        monkeypatch.delenv("TOKSCORE_INCR_DIV", raising=False)
    else:
        monkeypatch.setenv("TOKSCORE_INCR_DIV", incr_div)
    monkeypatch.setenv("TOKSCORE_SCAN", "tile")
    monkeypatch.setenv("TOKSCORE_DFA", "0")
    corpus, off = synth.wiki_like(24_000_000, seed=41)
    tokens = synth.build_vocab(corpus[:1_000_000], 8000)
    tokens.append(bytes(corpus[12_345_600:12_345_613]).lower())   # occurs in very few chunks
    V = len(tokens)
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        s.score()
        enabled = np.ones(V, dtype=bool)
        took_partial = []
        for i in (V - 1, V // 2, V - 1):
            enabled[i] = not enabled[i]
            s.toggle(int(i), bool(enabled[i]))
            got = s.score(incremental=True)
            kept = [t if enabled[k] else b"" for k, t in enumerate(tokens)]
            want = O.OracleVocab(*synth.pack_vocab(kept)).score(corpus, off, threads=8)
            assert got[1:] == want[1:] and np.array_equal(got[0], want[0])
            took_partial.append(s.info().rescored_chunks != 0xFFFFFFFF)
        if incr_div == "3":
            assert took_partial[0] and took_partial[2], "a token in a handful of chunks: the partial passes were expected"
    finally:
        s.close()


@pytest.mark.parametrize("both", [False, True])
def test_moves_on_repetitive_text(both, monkeypatch):
    """Three letters, strings that overlap themselves: occurrences of a move's string inside the stretch an earlier
    one changed, walks that go on for dozens of tokens, occurrences that begin a longer token (dropped before the table),
    and the string as the corpus' last bytes (the token-start map has no bit behind them).  both: the chain of the
    vocabulary as it is walked as well, instead of read from the map."""
    if both:
        monkeypatch.setenv("TOKSCORE_LOCAL_BOTH", "1")
    rng = np.random.default_rng(11)
    docs = []
    for _ in range(400):
        n = int(rng.integers(20, 1500))
        runs = []
        while sum(len(r) for r in runs) < n:
            unit = [b"ab", b"aba", b"abc", b"c", b"ba", b"bab", b"abab", b"cab"][int(rng.integers(0, 8))]
            runs.append(unit * int(rng.integers(1, 12)))
        docs.append(b"".join(runs)[:n])
    docs.append(b"cabcababab" * 3 + b"abab")     # ends with strings of the vocabulary
    corpus = np.frombuffer(b"".join(docs), dtype=np.uint8).copy()
    off = np.zeros(len(docs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(d) for d in docs])
    tokens = [b"a", b"b", b"c", b"ab", b"ba", b"aba", b"bab", b"abab", b"abc", b"ca", b"cab", b"bc", b"ababab", b"cc", b"bca"]
    V = len(tokens)
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        s.score()
        enabled = np.ones(V, dtype=bool)
        idx = list(range(3, V))
        _check_moves(s, tokens, enabled, corpus, off, idx, [0] * len(idx))
        assert s.info().search_passes > 0, "the one-pass search served the call"
        for i in (5, 7, 10):                      # "aba", "abab", "cab" off: moves that enable, moves against a changed vocabulary
            s.toggle(i, False)
            enabled[i] = False
        s.score()
        got = _check_moves(s, tokens, enabled, corpus, off, idx, [0 if enabled[i] else 1 for i in idx])
        assert any(int(d) != 0 for d in got)
    finally:
        s.close()
