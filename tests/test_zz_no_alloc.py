"""tokscore_count_occurrences_static (the entry point that allocates nothing: its scratch is a
static array in the library's device image) against tokscore_count_occurrences, through the
C-ABI, bit for bit -- on two calls in a row, the second on a smaller corpus, so that whatever the
first call left in the shared array would show."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth

pytestmark = pytest.mark.gpu
E_STATE, E_CAPACITY = 4, 6  # include/tokscore.h


class _Hip:
    """hipMalloc / hipMemcpy through ctypes: torch's bundled HIP runtime must not be loaded second
    into a process that already runs libtokscore.so's."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")

    def malloc(self, nbytes):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p

    def fill(self, p, byte, nbytes):
        assert self.lib.hipMemset(p, C.c_int(byte), C.c_size_t(nbytes)) == 0
        assert self.lib.hipDeviceSynchronize() == 0  # (hipMemset does not wait; the scorer's stream is non-blocking)

    def to_host(self, p, n_u64):
        assert self.lib.hipDeviceSynchronize() == 0
        host = np.zeros(n_u64, dtype=np.uint64)
        assert self.lib.hipMemcpy(host.ctypes.data_as(C.c_void_p), p, C.c_size_t(n_u64 * 8), 2) == 0  # D2H
        return host

    def free(self, p):
        self.lib.hipFree(p)


def test_static_equals_allocating_entry_point_twice():
    hip = _Hip()
    max_cp = 8
    big, big_off = synth.wiki_like(3_000_000, seed=21)
    small, small_off = synth.cjk_heavy(150_000, seed=22)
    tokens = synth.build_vocab(np.concatenate([big[:400_000], small[:100_000]]), 6000, lowercase=False)
    V = len(tokens)
    s = ta.Scorer(0)
    d_out = hip.malloc((V + 4) * 8)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(big, big_off)
        # before any evaluation the trie is not on the device: refused, not allocated
        with pytest.raises(ta.TokscoreError) as ei:
            s.count_occurrences_static(max_cp, d_out.value)
        assert ei.value.status == E_STATE
        for corpus, off in ((big, big_off), (small, small_off)):
            s.set_corpus(corpus, off)
            exp_counts, exp_pos = s.count_occurrences(max_cp)
            hip.fill(d_out, 0xEE, (V + 4) * 8)  # stale output must not survive
            s.count_occurrences_static(max_cp, d_out.value)
            got = hip.to_host(d_out, V + 4)
            assert np.array_equal(got[:V + 3], exp_counts)
            assert int(got[V + 3]) == exp_pos
            assert int(exp_counts.sum()) > 0
        # and the oracle agrees with both on the small one
        voc = O.OracleVocab(*synth.pack_vocab(tokens))
        ora_counts, ora_pos = voc.count_occurrences(small, small_off, max_cp)
        assert np.array_equal(got[:V + 3], ora_counts) and int(got[V + 3]) == ora_pos
    finally:
        s.close()
        hip.free(d_out)


def test_static_with_few_rows_and_refusal(monkeypatch):
    """With less of the array usable (TOKSCORE_STATIC_LIMIT) the call runs fewer workgroups and stays
    exact; when not even one row of partial counts fits it refuses with a clear error."""
    hip = _Hip()
    max_cp = 6
    corpus, off = synth.wiki_like(1_200_000, seed=23)
    tokens = synth.build_vocab(corpus[:300_000], 5000, lowercase=False)
    V = len(tokens)
    row = 4 * (V + 4)
    s = ta.Scorer(0)
    d_out = hip.malloc((V + 4) * 8)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        exp_counts, exp_pos = s.count_occurrences(max_cp)
        monkeypatch.setenv("TOKSCORE_STATIC_LIMIT", str(16384 + 3 * row + 100))  # heads + firsts + 3 rows
        s.count_occurrences_static(max_cp, d_out.value)
        got = hip.to_host(d_out, V + 4)
        assert np.array_equal(got[:V + 3], exp_counts) and int(got[V + 3]) == exp_pos
        monkeypatch.setenv("TOKSCORE_STATIC_LIMIT", str(row))
        with pytest.raises(ta.TokscoreError) as ei:
            s.count_occurrences_static(max_cp, d_out.value)
        assert ei.value.status == E_CAPACITY
        assert "static buffer holds" in str(ei.value)
    finally:
        s.close()
        hip.free(d_out)


def test_static_call_leaves_pending_toggles_to_the_next_evaluation():
    """score -> toggle -> count_occurrences_static -> score(incremental): the static call must not swallow the
    pending toggle (it refuses while one is pending: it never changes the context's state), and the
    incremental re-score that follows equals a full evaluation of the toggled vocabulary."""
    hip = _Hip()
    max_cp = 6
    corpus, off = synth.wiki_like(1_500_000, seed=27)
    tokens = synth.build_vocab(corpus[:400_000], 5000)
    V = len(tokens)
    s = ta.Scorer(0)
    d_out = hip.malloc((V + 4) * 8)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        s.score()
        multi = [i for i, t in enumerate(tokens) if len(t) > 2]
        s.toggle(multi[7], False)
        with pytest.raises(ta.TokscoreError) as ei:
            s.count_occurrences_static(max_cp, d_out.value)
        assert ei.value.status == E_STATE and "pending" in str(ei.value)
        got = s.score(incremental=True)
        kept = [b"" if i == multi[7] else t for i, t in enumerate(tokens)]
        want = O.OracleVocab(*synth.pack_vocab(kept)).score(corpus, off, threads=8)
        assert got[1:] == want[1:] and np.array_equal(got[0], want[0])
        # with nothing pending the static call runs, and counts the vocabulary as it now is
        exp_counts, exp_pos = s.count_occurrences(max_cp)
        s.count_occurrences_static(max_cp, d_out.value)
        host = hip.to_host(d_out, V + 4)
        assert np.array_equal(host[:V + 3], exp_counts) and int(host[V + 3]) == exp_pos
        full = s.score()
        assert full[1:] == want[1:] and np.array_equal(full[0], want[0])
    finally:
        s.close()
        hip.free(d_out)
