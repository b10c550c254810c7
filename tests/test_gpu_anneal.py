"""The reference's whole flow (src/main.cpp:20-46) with the annealer on the host and the score on
the GPU: GetTokens -> TokenGenerator -> Generate -> GetSolution -> SolutionFile -> score of the
shard, checked against the CPU oracle for the vocabulary the annealer found; and the candidates'
`uses` (host builder) against the GPU's closed-vocabulary counting (row f1)."""
import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import annealing as A
from tokenizer_amd import files, synth

pytestmark = pytest.mark.gpu


def test_anneal_then_score_matches_oracle(tmp_path):
    corpus, off = synth.wiki_like(300_000, seed=31, n_words=3000, median_doc=1500)
    path = tmp_path / "solution.tokens.json"
    sol, counts, total_tokens, total_bytes = A.anneal_and_score(
        corpus, off, max_len=6, pref_token_count=800, pass_cnt=3, batch_size=16, threads=1, seed=11,
        solution_path=str(path))
    try:
        # the file is a valid .tokens.json holding the solution in GetSolution's order
        back = files.SolutionFile(str(path))
        assert back.IsValid() and back.tokens == sol.tokens
        assert 300 < len(sol.tokens) < 5000
        assert total_bytes == int(off[-1])
        exp_counts, exp_tt, exp_tb = O.OracleVocab(*synth.pack_vocab(sol.tokens)).score(corpus, off, threads=4)
        assert (total_tokens, total_bytes) == (exp_tt, exp_tb)
        assert np.array_equal(counts, exp_counts)
        # the annealed vocabulary compresses (main.cpp:45 prints this ratio)
        assert total_bytes / total_tokens > 1.1
        # same seed, one worker: the same solution again
        again, _, tt2, _ = A.anneal_and_score(corpus, off, max_len=6, pref_token_count=800, pass_cnt=3, batch_size=16,
                                              threads=1, seed=11)
        assert again.tokens == sol.tokens and tt2 == total_tokens
        again.close()
        # ... and again with the candidates extracted on the GPU instead of the host
        gpu, _, tt3, _ = A.anneal_and_score(corpus, off, max_len=6, pref_token_count=800, pass_cnt=3, batch_size=16,
                                            threads=1, seed=11, gpu_candidates=True)
        assert gpu.tokens == sol.tokens and tt3 == total_tokens
        gpu.close()
    finally:
        sol.close()


def test_trajectory_reports_the_measured_compression_per_pass():
    corpus, off = synth.wiki_like(200_000, seed=32, n_words=2500, median_doc=1200)
    traj = []
    sol, counts, tt, tb = A.anneal_and_score(corpus, off, max_len=6, pref_token_count=600, pass_cnt=2, batch_size=16,
                                             threads=2, seed=3, trajectory=traj)
    sol.close()
    assert [t["pass"] for t in traj] == [0, 1, 2]
    assert traj[-1]["total_tokens"] == tt and abs(traj[-1]["bytes_per_token"] - tb / tt) < 1e-12
    assert all(t["bytes_per_token"] > 1.0 for t in traj)


@pytest.mark.parametrize("kind,max_cp", [("wiki", 5), ("cjk", 4)])
def test_candidate_uses_equal_gpu_occurrence_counts(kind, max_cp):
    # two independent routes to Trie::AddString's freq (Trie.cpp:114-128): the host builder counts
    # while it builds the candidate trie, the GPU counts the finished candidate list over the corpus
    if kind == "wiki":
        corpus, off = synth.wiki_like(120_000, seed=41, n_words=2000, median_doc=900)
    else:
        corpus, off = synth.cjk_heavy(60_000, seed=42, doc_lo=64, doc_hi=4000)
    toks = A.GetTokens(corpus, off, max_cp)
    names = toks.names()
    uses = toks.arrays()[1]
    assert len(names) > 20_000
    s = ta.Scorer(0)
    try:
        s.set_vocab(*synth.pack_vocab(names))
        s.set_corpus(corpus, off)
        counts, positions = s.count_occurrences(max_cp)
    finally:
        s.close()
    assert np.array_equal(counts[2:2 + len(names)], uses)
    roots = toks.arrays()[3] == A.NONE
    assert positions == int(uses[roots].sum())


def _gpu_candidates(corpus, off, max_cp):
    s = ta.Scorer(0)
    try:
        s.set_corpus(corpus, off)
        return s.extract_candidates(max_cp)
    finally:
        s.close()


def test_gpu_candidates_equal_the_reference_file(tmp_path):
    # row f3: the GPU's sort-based extraction against the file the compiled reference's GetTokens wrote
    import gzip
    import json
    import os
    case = os.path.join(O.GOLDEN, "anneal_mixed")
    trace = json.loads(gzip.open(os.path.join(case, "trace.json.gz")).read().decode("utf-8"))
    docs = [t.encode("utf-8") for t in trace["docs"]]
    corpus, off = synth.pack_docs(docs)
    chr_, uses, lpar, rpar, positions = _gpu_candidates(corpus, off, trace["max_len"])
    assert len(chr_) == trace["n_tokens"]
    assert positions == sum(len(t) for t in trace["docs"])
    out = tmp_path / "gpu.bin"
    A.WriteTokens(str(out), A.Tokens.from_arrays(chr_, uses, lpar, rpar))
    assert out.read_bytes() == gzip.open(os.path.join(case, "candidates.bin.gz")).read()


@pytest.mark.parametrize("kind,nbytes,max_cp", [("wiki", 400_000, 10), ("cjk", 150_000, 6), ("utf8", 200_000, 5),
                                                ("wiki", 50_000, 1), ("short", 0, 7), ("wiki", 3_000_000, 4)])
def test_gpu_candidates_equal_host_builder(kind, nbytes, max_cp):
    if kind == "wiki":
        corpus, off = synth.wiki_like(nbytes, seed=51, n_words=4000, median_doc=700)
    elif kind == "cjk":
        corpus, off = synth.cjk_heavy(nbytes, seed=52, doc_lo=16, doc_hi=3000)
    elif kind == "utf8":
        corpus, off = synth.random_utf8(nbytes, seed=53, doc_lo=8, doc_hi=5000)
    else:  # empty documents, documents shorter than the window, repeats, 4-byte characters at document ends
        docs = ["", "a", "", "ab", "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", "𝄞", "x𝄞", "", "日本語", "abcabcabc", "é", ""]
        corpus, off = synth.pack_docs([d.encode("utf-8") for d in docs])
    want = A.GetTokens(corpus, off, max_cp).arrays()
    got = _gpu_candidates(corpus, off, max_cp)
    for name, a, b in zip(("chr", "uses", "lpar", "rpar"), got[:4], want):
        assert len(a) == len(b), name
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, (name, bad[:5], a[bad[:5]], b[bad[:5]])
    roots = want[3] == A.NONE
    assert got[4] == int(want[1][roots].sum())


@pytest.mark.parametrize("group_bytes", [1 << 30, 1, 450_000])
def test_gpu_candidates_shard_by_shard_equal_host_builder_on_all(group_bytes):
    # three shards = one trie over all files: extracted together by one sort on the GPU (the default: they fit one
    # group) / one after the other and merged on the host / two groups (the first two shards together, then the third)
    shards = [synth.wiki_like(300_000, seed=61, n_words=3000, median_doc=500),
              synth.cjk_heavy(120_000, seed=62, doc_lo=16, doc_hi=2000),
              synth.wiki_like(200_000, seed=63, n_words=3000, median_doc=900)]
    docs = []
    for b, o in shards:
        raw = b.tobytes()
        docs += [raw[int(o[i]):int(o[i + 1])] for i in range(len(o) - 1)]
    want = A.GetTokens(*synth.pack_docs(docs), 6).arrays()
    got = A.GetTokensGpu(shards, max_len=6, group_bytes=group_bytes).arrays()
    for name, a, b in zip(("chr", "uses", "lpar", "rpar"), got, want):
        assert np.array_equal(a, b), name
    one = A.GetTokensGpu(*shards[0], max_len=6).arrays()
    assert all(np.array_equal(a, b) for a, b in zip(one, A.GetTokens(*shards[0], 6).arrays()))


def test_extract_candidates_rejects_what_it_cannot_take():
    s = ta.Scorer(0)
    try:
        with pytest.raises(ta.TokscoreError):
            s.extract_candidates(5)  # no corpus
        corpus, off = synth.pack_docs([b"abc", b"a\x00b"])
        s.set_corpus(corpus, off)
        with pytest.raises(ta.TokscoreError):
            s.extract_candidates(5)  # NUL inside a document
        with pytest.raises(ta.TokscoreError):
            s.extract_candidates(0)
        with pytest.raises(ta.TokscoreError):
            s.extract_candidates(33)
        corpus, off = synth.pack_docs([b"", b""])
        s.set_corpus(corpus, off)
        assert len(s.extract_candidates(5)[0]) == 0
    finally:
        s.close()


def test_moves_judged_on_the_gpu_keep_the_books():
    """evaluator="gpu": GpuMoveEvaluator behind tokanneal_set_evaluator.  After every batch the raw
    score must be the tokens saved against the start, the scorer's enabled set the generator's, and
    the scorer's token count what the CPU oracle gets for that set."""
    corpus, off = synth.wiki_like(120_000, seed=41)
    corpus = np.frombuffer(bytes(corpus).lower(), dtype=np.uint8)
    toks = A.GetTokens(corpus, off, 3)
    names = toks.names()
    rpar = toks.arrays()[3]
    n = len(names)
    judge = A.GpuMoveEvaluator(toks, corpus, off)
    gen = A.TokenGenerator(toks, 400, 32, evaluator=judge)
    try:
        gen.seed(5)
        for k in range(12):
            gen.WorkerTask()
            st = gen.state()
            assert st.raw_score == judge.start_tokens - judge.total_tokens, k
        assert judge.calls >= 12 and judge.moves == 12 * 32
        on = np.array([rpar[i] == A.NONE or gen.token_state(i)[2] for i in range(n)])
        sub = [t if on[k] else b"\x00OFF" + str(k).encode() for k, t in enumerate(names)]
        exp = O.OracleVocab(*synth.pack_vocab(sub)).score(corpus, off, threads=8)
        assert judge.scorer.score()[1] == exp[1] == judge.total_tokens
        assert st.raw_score > 0 and st.enabled_cnt == int(on.sum()) - int((rpar == A.NONE).sum())
    finally:
        gen.close()
        judge.close()


def test_anneal_and_score_with_the_gpu_evaluator():
    corpus, off = synth.wiki_like(60_000, seed=42)
    corpus = np.frombuffer(bytes(corpus).lower(), dtype=np.uint8)
    sol, counts, tt, tb = A.anneal_and_score(corpus, off, max_len=3, pref_token_count=300, pass_cnt=0, batch_size=64,
                                             threads=1, seed=9, evaluator="gpu")
    try:
        assert tb == len(corpus) and 0 < tt < tb
        voc = O.OracleVocab(*synth.pack_vocab(sol.tokens))
        exp = voc.score(corpus, off, threads=8)
        assert (tt, tb) == exp[1:] and np.array_equal(counts, exp[0])
    finally:
        sol.close()
