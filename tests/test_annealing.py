"""Rows a5 / f2 / f4: libtokanneal.so -- candidate DAG, `.candidates-*.bin`,
move primitives and the annealing loop -- against what the COMPILED REFERENCE produced:

    tests/golden/anneal_mixed/candidates.bin.gz   GetTokens' own WriteTokens output (Token.cpp:163-193)
    tests/golden/anneal_mixed/trace.json.gz       ref_harness `trace`: a seeded single-threaded run of
                                                  TokenGenerator::WorkerTask, the state after every task,
                                                  GetSolution at the end; plus the documents
    tests/golden/dag_tiny.json                    ref_harness `dag`: scripted SimulateStep / ApplyStep

All host code: these run without a GPU.  Doubles are compared for equality -- the library does the
reference's operations in the reference's order with the same libstdc++ generator/distributions."""
import ctypes as C
import gzip
import json
import os
import re

import numpy as np
import pytest

import oracle_ffi as O
from tokenizer_amd import annealing as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(O.GOLDEN, "anneal_mixed")


@pytest.fixture(scope="module")
def trace():
    return json.loads(gzip.open(os.path.join(CASE, "trace.json.gz")).read().decode("utf-8"))


@pytest.fixture(scope="module")
def golden_bin(tmp_path_factory):
    p = tmp_path_factory.mktemp("cands") / "golden.bin"
    p.write_bytes(gzip.open(os.path.join(CASE, "candidates.bin.gz")).read())
    return str(p)


def pack_docs(texts):
    docs = [t.encode("utf-8") for t in texts]
    off = np.zeros(len(docs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint64)
    corpus = np.frombuffer(b"".join(docs), dtype=np.uint8)
    return corpus, off


def test_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "tokanneal.h")).read()
    declared = sorted(set(re.findall(r"\b(tokanneal_[a-z_]+)\s*\(", text)))
    assert declared == sorted(A.EXPORTED_SYMBOLS)
    lib = C.CDLL(A.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert A.load_library().tokanneal_abi_version() == 100


def test_read_tokens_parses_the_reference_file(trace, golden_bin):
    code, toks = A.ReadTokens(golden_bin)
    assert code == A.READ_OK
    assert len(toks) == trace["n_tokens"]
    chr_, uses, lpar, rpar = toks.arrays()
    roots = rpar == A.NONE
    assert np.array_equal(roots, lpar == A.NONE)
    assert int((~roots).sum()) == trace["tot_cand"]
    names = toks.names()
    # the DAG's shape (Trie.cpp:92-103): l = the string minus its first code point, r = minus its last
    for i in range(0, len(toks), 7):
        s = names[i].decode("utf-8")
        assert toks.GetName(i) == names[i] and toks.size(i) == len(s)
        assert chr(int(chr_[i])) == s[0]
        if len(s) > 1:
            assert names[lpar[i]].decode("utf-8") == s[1:] and names[rpar[i]].decode("utf-8") == s[:-1]
    # pre-order of a trie whose children are sorted by code point = sorted by code point sequence
    cps = [tuple(ord(c) for c in n.decode("utf-8")) for n in names]
    assert cps == sorted(cps) and len(set(cps)) == len(cps)


def test_get_tokens_builds_the_reference_candidates(trace, golden_bin, tmp_path):
    # FileCandidates + BuildTokens restated (hash trie, children ordered afterwards) against the
    # reference's pointer trie: same tokens, same order, same counts, same parents ...
    built = A.GetTokens(*pack_docs(trace["docs"]), max_len=trace["max_len"])
    _, ref = A.ReadTokens(golden_bin)
    for a, b in zip(built.arrays(), ref.arrays()):
        assert np.array_equal(a, b)
    # ... and WriteTokens writes the very file the reference wrote
    out = tmp_path / A.candidates_file_name(None, trace["max_len"])
    assert out.name == ".candidates-all-5.bin"
    A.WriteTokens(str(out), built)
    assert out.read_bytes() == open(golden_bin, "rb").read()
    # uses are occurrence counts: check a few against a plain count over the documents
    names = built.names()
    uses = built.arrays()[1]
    for i in (0, 11, 500, 4000, len(names) - 1):
        s = names[i].decode("utf-8")
        want = sum(sum(1 for k in range(len(d) - len(s) + 1) if d[k:k + len(s)] == s) for d in trace["docs"])
        assert int(uses[i]) == want, s


def test_read_tokens_error_codes(golden_bin, tmp_path):
    blob = open(golden_bin, "rb").read()

    def code_of(data):
        p = tmp_path / "x.bin"
        p.write_bytes(data)
        return A.ReadTokens(str(p))[0]

    assert A.ReadTokens(str(tmp_path / "missing.bin"))[0] == A.BAD_STREAM      # Token.cpp:110
    assert code_of(b"0.3.9\x00" + blob[6:]) == A.INVALID_VERSION               # :115-116
    assert code_of(b"0.4.0") == A.INVALID_VERSION
    assert code_of(blob[:len(blob) // 2]) == A.BAD_FILE                        # :146
    assert code_of(blob[:-3]) == A.BAD_FILE
    assert code_of(blob + b"\x00") == A.NOT_DONE                               # :159
    code, toks = A.ReadTokens(golden_bin)
    assert code == A.READ_OK and len(toks) > 0
    p = tmp_path / "empty.bin"
    p.write_bytes(b"0.4.0\x00" + (0).to_bytes(8, "little"))                    # :121
    code, toks = A.ReadTokens(str(p))
    assert code == A.READ_OK and len(toks) == 0
    # a parent index past the end is a corrupt file, not a crash
    bad = bytearray(blob)
    bad[-4:] = (0x7FFFFFFF).to_bytes(4, "little")
    assert code_of(bytes(bad)) == A.BAD_FILE


def test_varint_is_read_at_64_bits(tmp_path):
    # the reference's reader shifts an int (Token.cpp:136) and breaks above 2^31 uses; the writer is fine
    big = (1 << 40) + 12345
    t = A.Tokens.from_arrays([ord("a"), ord("a")], [big, big - 7], [A.NONE, 0], [A.NONE, 0])
    p = tmp_path / "big.bin"
    A.WriteTokens(str(p), t)
    code, back = A.ReadTokens(str(p))
    assert code == A.READ_OK
    assert [int(x) for x in back.arrays()[1]] == [big, big - 7]
    assert back.GetName(1) == b"aa" and back.size(1) == 2


def test_merge_of_two_corpora_is_the_set_of_both(trace, golden_bin):
    # GetTokens over several files is one trie over all of them (GetTokens.cpp:31-90): the merge of the
    # per-part sets has to be the builder's set for the parts together -- and the builder is pinned by
    # the compiled reference's candidates.bin above
    docs = [d.encode("utf-8") for d in trace["docs"]]
    max_len = trace["max_len"]

    def build(part):
        off = np.zeros(len(part) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(d) for d in part], dtype=np.uint64)
        return A.GetTokens(np.frombuffer(b"".join(part), dtype=np.uint8) if part else np.zeros(0, np.uint8), off, max_len)

    whole = build(docs)
    _, ref = A.ReadTokens(golden_bin)
    for cut in (0, 1, len(docs) // 3, len(docs) - 1, len(docs)):
        a, b = build(docs[:cut]), build(docs[cut:])
        for m in (a.merge(b), b.merge(a)):
            for x, y, z in zip(m.arrays(), whole.arrays(), ref.arrays()):
                assert np.array_equal(x, y) and np.array_equal(x, z)
    # three parts, and a set merged with itself: the same strings, twice the uses
    third = len(docs) // 3
    m3 = build(docs[:third]).merge(build(docs[third:2 * third])).merge(build(docs[2 * third:]))
    assert all(np.array_equal(x, y) for x, y in zip(m3.arrays(), whole.arrays()))
    twice = whole.merge(whole).arrays()
    assert np.array_equal(twice[1], 2 * whole.arrays()[1])
    assert all(np.array_equal(twice[k], whole.arrays()[k]) for k in (0, 2, 3))


def test_merge_rejects_sets_that_are_not_in_builder_order():
    # "a", "ab", "b" in pre-order ...
    good = A.Tokens.from_arrays([97, 97, 98], [3, 1, 2], [A.NONE, 2, A.NONE], [A.NONE, 0, A.NONE])
    assert len(good.merge(good)) == 3
    # ... and with the roots swapped: "b", "a", "ab" is a valid DAG but not the builder's order
    bad = A.Tokens.from_arrays([98, 97, 97], [2, 3, 1], [A.NONE, A.NONE, 0], [A.NONE, A.NONE, 1])
    for pair in ((good, bad), (bad, good), (bad, bad)):
        with pytest.raises(A.TokannealError) as e:
            pair[0].merge(pair[1])
        assert e.value.status == A.E_DATA
    # a token before its prefix
    bad2 = A.Tokens.from_arrays([97, 97, 98], [1, 3, 2], [2, A.NONE, A.NONE], [1, A.NONE, A.NONE])
    with pytest.raises(A.TokannealError):
        good.merge(bad2)


def test_from_arrays_rejects_what_is_not_a_dag():
    with pytest.raises(A.TokannealError):
        A.Tokens.from_arrays([97, 97], [1, 1], [1, 0], [1, 0])              # a cycle
    with pytest.raises(A.TokannealError):
        A.Tokens.from_arrays([97, 97], [1, 1], [A.NONE, 5], [A.NONE, 0])    # out of range
    with pytest.raises(A.TokannealError):
        A.Tokens.from_arrays([97, 97], [1, 1], [A.NONE, A.NONE], [A.NONE, 0])  # a root on one side only


def test_move_primitives_match_the_reference_dump():
    g = json.load(open(os.path.join(O.GOLDEN, "dag_tiny.json"), encoding="utf-8"))
    toks = g["tokens"]
    none = lambda v: A.NONE if v < 0 else v  # noqa: E731
    t = A.Tokens.from_arrays([x["chr"] for x in toks], [x["uses"] for x in toks], [none(x["l"]) for x in toks],
                             [none(x["r"]) for x in toks])
    assert [n.decode("utf-8") for n in t.names()] == [x["name"] for x in toks]
    gen = A.TokenGenerator(t, 30000, 30)
    assert len(t) == 0  # moved from
    assert gen.state().tot_cand == g["tot_cand"]
    for i, x in enumerate(toks):
        assert gen.token_state(i) == (x["uses"], x["uses"], bool(x["enabled"]))
    for s in g["steps"]:
        assert gen.SimulateStep(s["i"]) == s["sim"], s
        assert gen.ApplyStep(s["i"], bool(s["enable"])) == s["apply"], s
    for i, (lu, ru, en) in enumerate(g["final"]):
        assert gen.token_state(i) == (lu, ru, bool(en))
    root = next(i for i, x in enumerate(toks) if x["size"] == 1)
    with pytest.raises(A.TokannealError):
        gen.SimulateStep(root)  # single code points never move (TokenGenerator.cpp:35-39)


def _calc_score(s, pref):
    """TokenGenerator::CalcScore through the ORACLE's restatement (oracle/dag_oracle.c)."""
    from test_dag_oracle import Lomax, _lib
    L = _lib()
    d = Lomax(0.0, s["moment1"], s["moment2"], s["beta"], s["sigma"])
    return L.ora_calc_score(C.byref(d), s["raw_score"], s["enabled_cnt"], s["tot_cand"], pref)


def test_worker_tasks_follow_the_reference_step_by_step(trace, golden_bin):
    _, toks = A.ReadTokens(golden_bin)
    gen = A.TokenGenerator(toks, trace["pref"], trace["batch"])
    gen.seed(trace["seed"])
    keys = ("gen_cnt", "enabled_cnt", "raw_score", "temp", "moment1", "moment2", "beta", "sigma")
    assert len(trace["states"]) >= 500
    for k, want in enumerate(trace["states"]):
        gen.WorkerTask()
        got = gen.state().as_dict()
        assert {x: got[x] for x in keys} == {x: want[x] for x in keys}, "task %d" % k
        if k % 50 == 0:
            assert got["score"] == _calc_score(got, trace["pref"])
    assert got["tot_cand"] == trace["tot_cand"]
    assert [s.decode("utf-8") for s in gen.GetSolution()] == trace["solution"]
    b, off = gen.solution_packed()
    assert len(off) == len(trace["solution"]) + 1 and int(off[-1]) == len(b)


def test_generate_single_worker_is_the_same_sequence_of_tasks(trace, golden_bin):
    # Generate(pass_cnt) = (pass_cnt + 1) x ceil(tot_cand / batch) WorkerTasks (TokenGenerator.cpp:178-183)
    per_pass = -(-trace["tot_cand"] // trace["batch"])
    assert per_pass <= len(trace["states"])
    _, toks = A.ReadTokens(golden_bin)
    gen = A.TokenGenerator(toks, trace["pref"], trace["batch"])
    states = gen.Generate(0, threads=1, seed=trace["seed"])
    assert len(states) == 1
    want = trace["states"][per_pass - 1]
    got = states[0].as_dict()
    for k in ("gen_cnt", "enabled_cnt", "raw_score", "temp", "moment1", "moment2", "beta", "sigma"):
        assert got[k] == want[k], k
    assert got["gen_cnt"] == per_pass * trace["batch"]


def _enabled(gen, rpar):
    return [i for i in range(len(rpar)) if rpar[i] != A.NONE and gen.token_state(i)[2]]


def test_generate_with_several_workers_keeps_the_books(trace, golden_bin):
    _, toks = A.ReadTokens(golden_bin)
    rpar = toks.arrays()[3]
    gen = A.TokenGenerator(toks, trace["pref"], trace["batch"])
    states = gen.Generate(3, threads=4, seed=7)
    assert len(states) == 4
    per_pass = -(-trace["tot_cand"] // trace["batch"])
    assert [s.gen_cnt for s in states] == [per_pass * trace["batch"] * (k + 1) for k in range(4)]
    st = gen.state()
    assert len(_enabled(gen, rpar)) == st.enabled_cnt
    sol = gen.GetSolution()
    assert len(sol) == st.enabled_cnt + int((rpar == A.NONE).sum()) and len(set(sol)) == len(sol)
    # the pull towards pref_token_count (TokenGenerator.cpp:108-117) holds the count in its neighbourhood
    assert 0.3 * trace["pref"] < st.enabled_cnt < 3 * trace["pref"]


def test_undoing_every_move_restores_the_counts(trace, golden_bin):
    # `uses` is a function of the enabled set alone (what an enabled descendant claims is taken off
    # every ancestor up to the next enabled one): taking all moves back, in any order, must leave
    # every token with its occurrence count and the raw score's bookkeeping at zero
    _, toks = A.ReadTokens(golden_bin)
    n = len(toks)
    _, uses, _, rpar = toks.arrays()
    gen = A.TokenGenerator(toks, trace["pref"], trace["batch"])
    gen.Generate(1, threads=1, seed=99)
    raw = gen.state().raw_score
    enabled = _enabled(gen, rpar)
    assert len(enabled) > 20
    rng = np.random.default_rng(3)
    for i in rng.permutation(enabled):
        sim = gen.SimulateStep(int(i))
        assert gen.ApplyStep(int(i), False) == sim
        raw -= sim
    assert raw == 0
    for i in range(n):
        assert gen.token_state(i) == (int(uses[i]), int(uses[i]), rpar[i] == A.NONE)


# ---- moves judged by an evaluator instead of the DAG (tokanneal_set_evaluator) ----------------

def test_evaluator_that_returns_the_dag_estimate_changes_nothing(trace, golden_bin):
    """With one move per batch, an evaluator that answers with Token::SimulateStep's own value
    must walk the very same trajectory as the DAG path (same seed, same draws, same decisions)."""
    _, toks = A.ReadTokens(golden_bin)
    plain = A.TokenGenerator(toks, trace["pref"], 1)
    _, toks = A.ReadTokens(golden_bin)
    judged = A.TokenGenerator(toks, trace["pref"], 1)

    class Dag:
        calls = 0

        def evaluate(self, cands, enable):
            Dag.calls += 1
            return [judged.SimulateStep(int(c)) for c in cands]

    judged.set_evaluator(Dag())
    plain.seed(7)
    judged.seed(7)
    for k in range(400):
        plain.WorkerTask()
        judged.WorkerTask()
        a, b = plain.state().as_dict(), judged.state().as_dict()
        assert a == b, k
    assert Dag.calls == 400
    assert plain.GetSolution() == judged.GetSolution()
    judged.set_evaluator(None)          # back on the DAG
    plain.WorkerTask()
    judged.WorkerTask()
    assert plain.state().as_dict() == judged.state().as_dict()


def test_corpus_judged_moves_keep_the_raw_score_exact():
    """The GPU evaluator's protocol with the CPU oracle in the scorer's place: values are exact
    token-count differences of the corpus, accepted moves are reported, and after every batch the
    raw score is the tokens saved against the single-code-point start."""
    texts = ["the cat sat on the mat. the cat ate the rat.", "a cat, a mat, a rat and a bat",
             "日本語の本 日本の本", "that hat that sat"] * 3
    corpus, off = pack_docs(texts)
    toks = A.GetTokens(corpus, off, 4)
    names = toks.names()
    rpar = toks.arrays()[3]
    n = len(names)
    enabled = rpar == A.NONE

    def total(mask):
        sub = [t if mask[k] else b"\x00OFF" + str(k).encode() for k, t in enumerate(names)]
        b = np.frombuffer(b"".join(sub), dtype=np.uint8)
        o = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(t) for t in sub], out=o[1:])
        return O.OracleVocab(b, o).score(corpus, off)[1]

    class Judge:
        start = total(enabled)
        log = []

        def evaluate(self, cands, enable):
            base = total(enabled)
            vals = []
            for c in cands:
                assert bool(enabled[c]) != enable
                m = enabled.copy()
                m[c] = enable
                d = total(m) - base
                vals.append(-d if enable else d)
            Judge.log.append((len(cands), enable))
            return vals

        def applied(self, cand, enable):
            enabled[cand] = enable

        def resync(self):
            return float(Judge.start - total(enabled))

    gen = A.TokenGenerator(toks, 40, 8, evaluator=Judge())
    gen.seed(3)
    for k in range(30):
        gen.WorkerTask()
        st = gen.state()
        assert st.raw_score == Judge.start - total(enabled), k
        on = [i for i in range(n) if rpar[i] != A.NONE and gen.token_state(i)[2]]
        assert on == [i for i in range(n) if rpar[i] != A.NONE and enabled[i]], k
        assert st.enabled_cnt == len(on)
    assert sum(c for c, _ in Judge.log) == 30 * 8
    assert st.raw_score > 0, "judged on the corpus the annealer must find tokens that shorten it"


def test_a_failing_evaluator_surfaces_as_its_own_exception(golden_bin):
    _, toks = A.ReadTokens(golden_bin)

    class Broken:
        def evaluate(self, cands, enable):
            raise KeyError("no corpus")

    gen = A.TokenGenerator(toks, 50, 4, evaluator=Broken())
    with pytest.raises(KeyError):
        gen.WorkerTask()
