"""The automaton scan (kernels_dfa.hip; the default path of full evaluations since round 3) through the
C-ABI against the CPU oracle: ragged and tiny corpora, ranges that are mostly padding, every number
of ranges per lane, LDS prefixes from "everything" to "almost nothing", toggles (the image is
rebuilt from the vocabulary as it is), and the vocabulary the reference's annealer wrote."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


FORCE = {"TOKSCORE_DFA_MAX_COLD_PPM": "1000000", "TOKSCORE_DFA_MAX_VIRTUAL_PPM": "1000000"}


def _check(s, voc, corpus, off, what=""):
    want = voc.score(corpus, off, threads=8)
    got = s.score()
    assert got[1:] == want[1:], (what, got[1:], want[1:])
    bad = np.nonzero(got[0] != want[0])[0]
    assert bad.size == 0, (what, bad[:8], got[0][bad[:8]], want[0][bad[:8]])


@pytest.mark.parametrize("chains", ["1", "2", "4"])
@pytest.mark.parametrize("lds", ["0", "6000", "3000"])
def test_automaton_scan_matches_oracle(chains, lds):
    corpus, off = synth.wiki_like(3_000_000, seed=21)
    tokens = synth.build_vocab(corpus[:800_000], 12000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    # (a small LDS array sends a third of the lookups to the global array: the library would leave such a
    # vocabulary to the plan's scan -- FORCE lifts that, so that the paths around the global array are run)
    with _Env(TOKSCORE_DFA_CHAINS=chains, TOKSCORE_DFA_LDS=lds, **FORCE):
        s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    _check(s, voc, corpus, off, "first evaluation (every range entry repaired)")
    info = s.info()
    assert info.dfa_active == 1 and info.scan_path == 2 and info.dfa_chains == int(chains)
    if lds != "0":
        assert info.dfa_lds_slots == int(lds) and info.dfa_slots > info.dfa_lds_slots
    _check(s, voc, corpus, off, "second evaluation (what LDS holds chosen again from the first one's counters)")
    _check(s, voc, corpus, off, "third evaluation")
    _check(s, voc, corpus, off, "fourth evaluation (entries as the third left them)")
    assert s.info().dfa_relayouts == 1
    s.close()


def test_automaton_scan_ragged_and_tiny_documents():
    rng = np.random.default_rng(5)
    base, boff = synth.wiki_like(400_000, seed=22)
    tokens = synth.build_vocab(base[:200_000], 3000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    raw = base.tobytes()
    docs = [b"", b"a", b"", raw[:7], raw[1000:1003], b"", raw[5000:5000 + 70_001], b"Q", b""]
    docs += [raw[int(p):int(p) + int(n)] for p, n in zip(rng.integers(0, 300_000, 300), rng.integers(0, 40, 300))]
    for sub in (docs, docs[:3], [b""], [b"xyz"], [raw[:100_000]]):
        corpus, off = synth.pack_docs(sub)
        s = ta.Scorer(0)
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        _check(s, voc, corpus, off, len(sub))
        if len(corpus):
            assert s.info().dfa_active == 1
        _check(s, voc, corpus, off, len(sub))
        s.close()


def test_automaton_scan_after_toggles():
    corpus, off = synth.wiki_like(2_000_000, seed=23)
    tokens = synth.build_vocab(corpus[:600_000], 6000)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    _check(s, voc, corpus, off)
    assert s.info().dfa_active == 1
    rng = np.random.default_rng(9)
    multi = [i for i, t in enumerate(tokens) if len(t) > 1]
    gone = set()
    for rnd in range(3):
        for i in rng.choice(multi, size=5, replace=False):
            if int(i) in gone:
                gone.discard(int(i))
                s.toggle(int(i), True)
            else:
                gone.add(int(i))
                s.toggle(int(i), False)
        kept = [b"" if i in gone else t for i, t in enumerate(tokens)]
        voc2 = O.OracleVocab(*synth.pack_vocab(kept))
        # the first evaluation after a toggle runs on the plan's own scan (one patched slot) ...
        _check(s, voc2, corpus, off, "after toggles, round %d" % rnd)
        assert s.info().dfa_active == 0
        # ... and once two evaluations in a row met no new toggle, the image is rebuilt
        _check(s, voc2, corpus, off)
        assert s.info().dfa_active == 1
        _check(s, voc2, corpus, off)
        # incremental re-scores start from the automaton scan's counts like from any other full evaluation
        j = int(rng.choice(multi))
        s.toggle(j, j in gone)
        gone.symmetric_difference_update({j})
        kept = [b"" if i in gone else t for i, t in enumerate(tokens)]
        want = O.OracleVocab(*synth.pack_vocab(kept)).score(corpus, off, threads=8)
        got = s.score(incremental=True)
        assert got[1:] == want[1:] and np.array_equal(got[0], want[0])
    s.close()


FAIL_FORM = {"TOKSCORE_DFA_FORM": "2"}


@pytest.mark.parametrize("chains", ["1", "2", "4"])
@pytest.mark.parametrize("lds", ["0", "3000"])
def test_failure_form_matches_oracle(chains, lds):
    # the same vocabulary in the failure form (f followed state by state, lanes with byte windows of their own): with
    # all of LDS nearly every lookup is answered there, with 3,000 slots most rows are only in the global array
    corpus, off = synth.wiki_like(3_000_000, seed=21)
    tokens = synth.build_vocab(corpus[:800_000], 12000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    with _Env(TOKSCORE_DFA_CHAINS=chains, TOKSCORE_DFA_LDS=lds, **FAIL_FORM):
        s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    _check(s, voc, corpus, off, "first evaluation (every range entry repaired)")
    info = s.info()
    assert info.dfa_active == 1 and info.scan_path == 2 and info.dfa_chains == int(chains) and info.dfa_form == 2
    assert info.dfa_virtual_slots == 0
    _check(s, voc, corpus, off, "second evaluation (what LDS holds chosen again from the first one's counters)")
    _check(s, voc, corpus, off, "third evaluation (the global array's hits no longer counted)")
    _check(s, voc, corpus, off, "fourth evaluation")
    s.close()


def test_failure_form_ragged_documents_and_toggles():
    rng = np.random.default_rng(6)
    base, boff = synth.wiki_like(400_000, seed=22)
    tokens = synth.build_vocab(base[:200_000], 3000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    raw = base.tobytes()
    docs = [b"", b"a", b"", raw[:7], raw[1000:1003], b"", raw[5000:5000 + 70_001], b"Q", b""]
    docs += [raw[int(p):int(p) + int(n)] for p, n in zip(rng.integers(0, 300_000, 300), rng.integers(0, 40, 300))]
    for sub in (docs, docs[:3], [b""], [b"xyz"], [raw[:100_000]]):
        corpus, off = synth.pack_docs(sub)
        with _Env(**FAIL_FORM):
            s = ta.Scorer(0)
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        _check(s, voc, corpus, off, len(sub))
        if len(corpus):
            assert s.info().dfa_active == 1 and s.info().dfa_form == 2
        _check(s, voc, corpus, off, len(sub))
        s.close()
    # toggles: the plan's scan serves the evaluations until the image has been rebuilt, in the same form
    corpus, off = synth.wiki_like(1_500_000, seed=23)
    with _Env(**FAIL_FORM):
        s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    _check(s, voc, corpus, off)
    multi = [i for i, t in enumerate(tokens) if len(t) > 1]
    gone = set(int(i) for i in rng.choice(multi, size=6, replace=False))
    for i in gone:
        s.toggle(i, False)
    voc2 = O.OracleVocab(*synth.pack_vocab([b"" if i in gone else t for i, t in enumerate(tokens)]))
    _check(s, voc2, corpus, off, "after toggles")
    assert s.info().dfa_active == 0
    _check(s, voc2, corpus, off)
    assert s.info().dfa_active == 1 and s.info().dfa_form == 2
    _check(s, voc2, corpus, off)
    s.close()


def test_reference_annealer_vocabulary():
    # tests/golden/wiki_annealed: 30,275 tokens the reference's own TokenGenerator wrote, 124,789 trie
    # nodes, most prefixes not tokens -- flattened that is 2.8 M virtual slots, a third of the steps arrive on one
    # and almost half of the lookups need the global array, so the library takes the failure form for it
    # (tokscore.hip: ensure_dfa); the flattened form, when forced, still counts exactly
    g = O.load_golden("wiki_annealed")
    corpus, off = synth.pack_docs(g["docs"])
    for force in (False, True):
        with _Env(**(FORCE if force else {})):
            s = ta.Scorer(0)
        s.set_vocab(*synth.pack_vocab(g["tokens"]))
        s.set_corpus(corpus, off)
        for _ in range(3):
            counts, tt, tb = s.score()
            assert tt == g["expected"]["total_tokens"] and tb == g["expected"]["total_bytes"]
            assert np.array_equal(counts, g["expected"]["counts_vec"])
        info = s.info()
        assert info.dfa_active == 1 and info.scan_path == 2
        assert info.dfa_form == (1 if force else 2)
        if force:
            assert info.dfa_virtual_slots > 1_000_000 and info.dfa_slots > info.dfa_lds_slots
        else:
            assert info.dfa_virtual_slots == 0
        s.close()
    # ... and on a corpus it was not made from, against the oracle
    corpus, off = synth.wiki_like(4_000_000, seed=24)
    voc = O.OracleVocab(*synth.pack_vocab(g["tokens"]))
    with _Env(**FORCE):
        s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(g["tokens"]))
    s.set_corpus(corpus, off)
    for _ in range(3):  # (the second and third evaluation run on re-laid-out images)
        _check(s, voc, corpus, off)
        assert s.info().dfa_active == 1
    s.close()


def test_cjk_vocabulary():
    # multibyte text: flattened, a tenth of the steps arrive on a virtual entry and more than a tenth of the lookups
    # need the global array -- the failure form (a twentieth) unless the flattened one is forced; exact either way
    corpus, off = synth.cjk_heavy(3_000_000, seed=5)
    tokens = synth.build_vocab(corpus[:1_000_000], 16000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    for force in (False, True):
        with _Env(**(FORCE if force else {})):
            s = ta.Scorer(0)
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        for _ in range(3):
            _check(s, voc, corpus, off)
            assert s.info().dfa_active == 1 and s.info().dfa_form == (1 if force else 2)
        s.close()


def test_other_paths_trie_and_plan_are_built_on_demand():
    # evaluations on the automaton need neither the packed trie of the other scan paths nor their plan (100 ms and
    # 50 ms per GB of host time): both appear with the first call that does -- and change no result
    corpus, off = synth.wiki_like(2_000_000, seed=26)
    tokens = synth.build_vocab(corpus[:600_000], 6000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    _check(s, voc, corpus, off)
    _check(s, voc, corpus, off)
    info = s.info()
    assert info.dfa_active == 1 and info.trie_slots == 0 and info.chunks == 0
    multi = [i for i, t in enumerate(tokens) if len(t) > 2]
    s.toggle(multi[7], False)
    assert s.info().trie_slots > 0
    voc2 = O.OracleVocab(*synth.pack_vocab([b"" if i == multi[7] else t for i, t in enumerate(tokens)]))
    want = voc2.score(corpus, off, threads=8)
    got = s.score(incremental=True)
    assert got[1:] == want[1:] and np.array_equal(got[0], want[0])
    assert s.info().chunks > 0
    s.toggle(multi[7], True)
    for _ in range(3):  # (the plan's scan, then the rebuilt image)
        _check(s, voc, corpus, off)
    assert s.info().dfa_active == 1
    s.close()


def test_async_evaluations_on_a_caller_stream():
    # (device memory and the stream through ctypes: torch's bundled HIP runtime must not be loaded second
    # into a process that already runs libtokscore.so's)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    corpus, off = synth.wiki_like(2_000_000, seed=25)
    tokens = synth.build_vocab(corpus[:600_000], 6000)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    want = voc.score(corpus, off, threads=8)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    n = len(tokens) + 5
    d_out, st = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_out), C.c_size_t(n * 8)) == 0
    assert hip.hipStreamCreate(C.byref(st)) == 0
    try:
        for _ in range(4):
            s.score_async(d_out.value, st.value)
        assert not s.async_status(st.value)
        got = np.zeros(n, dtype=np.uint64)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d_out, C.c_size_t(n * 8), 2) == 0
        assert np.array_equal(got[:n - 2], want[0]) and int(got[n - 2]) == want[1] and int(got[n - 1]) == want[2]
        assert s.info().dfa_active == 1
    finally:
        s.close()
        hip.hipStreamDestroy(st)
        hip.hipFree(d_out)
