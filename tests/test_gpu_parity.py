"""GPU parity: the HIP path (through the C-ABI) against the reference-generated goldens
and against the CPU oracle on seeded synthetic corpora.  Bit-exact: integer work only."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O
import tokenizer_amd as ta
from tokenizer_amd import synth

pytestmark = pytest.mark.gpu


def _scorer(env):
    # a test that names one of the plan's scan paths means that path: the automaton scan (the default for
    # full evaluations since round 3) is switched off unless the test asks for it
    if "TOKSCORE_SCAN" in env and "TOKSCORE_DFA" not in env:
        env = dict(env, TOKSCORE_DFA="0")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = ta.Scorer(0)
    finally:
        pass
    return s, old


def _restore(old):
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _score_with(env, tokens, corpus, off):
    """Plan parameters are read from the environment when the first score() plans the launch."""
    s, old = _scorer(env)
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        out = s.score()
        info = s.info()
        # a second evaluation on the resident corpus must reproduce the first
        out2 = s.score()
        assert np.array_equal(out[0], out2[0]) and out[1:] == out2[1:]
    finally:
        _restore(old)
        s.close()
    return out, info


CONFIGS = [
    {"TOKSCORE_SCAN": "generic", "TOKSCORE_CHUNK": "64"},     # boundary repair on nearly every chunk
    {"TOKSCORE_SCAN": "generic", "TOKSCORE_CHUNK": "2048"},
    {"TOKSCORE_SCAN": "tile", "TOKSCORE_CHUNK": "8192"},
    {"TOKSCORE_SCAN": "tile", "TOKSCORE_CHUNK": "300"},       # many mid-document chunk starts
    {"TOKSCORE_SCAN": "tile", "TOKSCORE_SEG": "132", "TOKSCORE_WAVES": "8"},
    {"TOKSCORE_SCAN": "tile", "TOKSCORE_LDS_SLOTS": "300"},   # most trie lookups fall through to L2
    {"TOKSCORE_SCAN": "generic", "TOKSCORE_CHUNK": "64", "TOKSCORE_NOSNAP": "1"},  # boundaries left mid-token: fix-up kernels
    {"TOKSCORE_SCAN": "tile", "TOKSCORE_CHUNK": "300", "TOKSCORE_NOSNAP": "1"},
    {"TOKSCORE_SCAN": "tile", "TOKSCORE_CHUNK": "300", "TOKSCORE_TILE_BLOCKS": "8"},   # fewer workgroups than queue heads (a small partition)
    {"TOKSCORE_SCAN": "tile", "TOKSCORE_CHUNK": "2048", "TOKSCORE_TILE_BLOCKS": "1"},
    {"TOKSCORE_SCAN": "stream"},                               # one byte stream per lane, trie image in LDS
    {"TOKSCORE_SCAN": "stream", "TOKSCORE_STREAM": "64", "TOKSCORE_SPC": "3"},    # shortest streams, ragged chunks
    {"TOKSCORE_SCAN": "stream", "TOKSCORE_STREAM": "1000", "TOKSCORE_WAVES": "4"},
]
# the plan's own scan paths above run with the automaton scan (round 3, the default for full evaluations)
# switched off; these run with it on: the library's defaults, fewer ranges per lane, and an LDS prefix so
# short that most lookups and events go the way of slots beyond it
for _c in CONFIGS:
    _c["TOKSCORE_DFA"] = "0"
# (FORCE: the library leaves vocabularies whose sample sends many lookups to the global array, or arrives often
# on virtual entries, to the plan's scan; the goldens are to run on the automaton whatever their shape)
FORCE = {"TOKSCORE_DFA_MAX_COLD_PPM": "1000000", "TOKSCORE_DFA_MAX_VIRTUAL_PPM": "1000000"}
DFA_CONFIGS = [
    dict(FORCE),
    dict(FORCE, TOKSCORE_DFA_CHAINS="1"),
    dict(FORCE, TOKSCORE_DFA_CHAINS="2", TOKSCORE_DFA_LDS="600"),
    dict(FORCE, TOKSCORE_SCAN="tile", TOKSCORE_DFA="1", TOKSCORE_DFA_LDS="2000"),
    # the failure form of the automaton (dfa.h: DfaImage::fail_code; dfa_fscan_kernel), whatever the vocabulary's shape:
    # four / one / two ranges per lane, and an LDS array so short that most rows are only in the global array
    {"TOKSCORE_DFA_FORM": "2"},
    {"TOKSCORE_DFA_FORM": "2", "TOKSCORE_DFA_CHAINS": "1"},
    {"TOKSCORE_DFA_FORM": "2", "TOKSCORE_DFA_CHAINS": "2", "TOKSCORE_DFA_LDS": "600"},
]
# goldens the automaton does not take (the plan's scan serves them): none any more (random_utf8's 190 distinct
# edge bytes fit the label's byte since it became one)
DFA_EXCEPT = set()
CONFIGS = CONFIGS + DFA_CONFIGS
# goldens whose text has sync points and whose trie fits LDS: the stream scan must be the path taken
STREAM_CASES = {"ascii_caps", "cjk_heavy", "short_docs", "prefix_greedy"}  # (wiki_annealed: 125k trie nodes, random_utf8: > 125 edge labels: tile scan)


@pytest.mark.parametrize("cfg", range(len(CONFIGS)))
@pytest.mark.parametrize("case", O.golden_cases())
def test_score_matches_reference_golden(case, cfg):
    g = O.load_golden(case)
    corpus, off = synth.pack_docs(g["docs"])
    (counts, tt, tb), info = _score_with(CONFIGS[cfg], g["tokens"], corpus, off)
    exp = g["expected"]
    assert tb == exp["total_bytes"]
    assert tt == exp["total_tokens"]
    bad = np.nonzero(counts != exp["counts_vec"])[0]
    assert bad.size == 0, (case, CONFIGS[cfg], bad[:10], counts[bad[:10]], exp["counts_vec"][bad[:10]])
    if CONFIGS[cfg].get("TOKSCORE_DFA") == "0":
        assert info.dfa_active == 0
        if CONFIGS[cfg]["TOKSCORE_SCAN"] == "tile":
            assert info.scan_path == 1, "tile scan was expected to be selected"
        if CONFIGS[cfg]["TOKSCORE_SCAN"] == "stream" and case in STREAM_CASES:
            assert info.scan_path == 2, "stream scan was expected to be selected"
    elif case in DFA_EXCEPT:
        assert info.dfa_active == 0
    elif "TOKSCORE_DFA_LDS" not in CONFIGS[cfg]:
        # every other reference-generated golden, wiki_annealed (125 k trie nodes) included; with an LDS array of a
        # few hundred slots the depth-1 rows of the larger vocabularies do not fit, and the plan's scan serves them
        assert info.dfa_active == 1 and info.scan_path == 2, "the automaton scan was expected to serve the evaluation"
        assert info.dfa_form == (2 if CONFIGS[cfg].get("TOKSCORE_DFA_FORM") == "2" else 1)


@pytest.mark.parametrize("case", [c for c in O.golden_cases() if "ids" in O.load_golden(c)["expected"]])
def test_tokenize_matches_reference_ids(case):
    g = O.load_golden(case)
    s = ta.Scorer(0)
    s.set_vocab(*synth.pack_vocab(g["tokens"]))
    for d, ids in zip(g["docs"], g["expected"]["ids"]):
        assert s.tokenize(d) == ids
    s.close()


@pytest.mark.parametrize("kind,nbytes,V", [("wiki", 6_000_000, 32000), ("cjk", 1_500_000, 8000),
                                            ("utf8", 1_500_000, 20000)])
def test_score_matches_oracle_on_synthetic(kind, nbytes, V):
    gen = {"wiki": synth.wiki_like, "cjk": synth.cjk_heavy, "utf8": synth.random_utf8}[kind]
    corpus, off = gen(nbytes)
    tokens = synth.build_vocab(corpus[:min(nbytes, 1_000_000)], V)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    exp_counts, exp_tt, exp_tb = voc.score(corpus, off, threads=8)
    for cfg in ({"TOKSCORE_SCAN": "stream"}, {"TOKSCORE_SCAN": "tile"}, {"TOKSCORE_SCAN": "generic"}):
        (counts, tt, tb), info = _score_with(cfg, tokens, corpus, off)
        assert (tt, tb) == (exp_tt, exp_tb), (kind, cfg)
        assert np.array_equal(counts, exp_counts), (kind, cfg)


@pytest.mark.parametrize("kind", ["utf8", "wiki"])
def test_wide_vocabulary_uses_32bit_stream(kind):
    # V + 3 > 32767 forces 32-bit stream entries, counted in byte-wide LDS bins (kernels_tile.hip hist_kernel)
    if kind == "utf8":
        corpus, off = synth.random_utf8(1_000_000, seed=11, doc_lo=512, doc_hi=100_000)
    else:
        corpus, off = synth.wiki_like(3_000_000, seed=12)
        # hot ids: some thousands of one token per workgroup, so that byte bins wrap (and wrap downwards
        # when a re-score takes the old tokens out again)
        docs = [corpus[off[i]:off[i + 1]].tobytes() for i in range(len(off) - 1)]
        # (the third: the four ids of one bin word counted in step, so that a carry out of one byte finds
        # its neighbour at the edge as well)
        docs[3:3] = [b"zq" * 40000, b"z" * 30000 + b"qzq" * 9000, b"zqzq\x01zq\x01q\x01z\x01" * 20000]
    tokens = synth.build_vocab(corpus[:1_000_000], 40000)
    if kind == "wiki":
        corpus, off = synth.pack_docs(docs)
        tokens = [t for t in tokens if t not in (b"zq", b"zqzq", b"z", b"q")][:39990] + [b"z", b"q", b"zq", b"zqzq"]
        assert (len(tokens) - 4 + 2) % 4 == 0   # ids of the last four: one word of byte bins
    # random text has too few repeated n-grams: top the list up with slices of the corpus
    text = corpus.tobytes().decode("utf-8")
    seen = set(tokens)
    k = 0
    while len(tokens) < 36000:
        t = text[k * 7:k * 7 + 2 + k % 4].encode("utf-8")
        k += 1
        if t and t not in seen:
            seen.add(t)
            tokens.append(t)
    assert len(tokens) > 33000
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    exp = voc.score(corpus, off, threads=8)
    # byte bins in one id range, and in several narrow ones (range boundaries inside a word's neighbours)
    for cfg in ({"TOKSCORE_SCAN": "tile"}, {"TOKSCORE_SCAN": "tile", "TOKSCORE_HIST_RANGE": "9001"}):
        (counts, tt, tb), info = _score_with(cfg, tokens, corpus, off)
        assert info.scan_path == 1
        assert (tt, tb) == exp[1:], cfg
        assert np.array_equal(counts, exp[0]), cfg
    # wiki: the hot ids pass 128 per workgroup many times over, the byte bins wrap and the wraps are booked
    assert kind != "wiki" or int(exp[0][2:].max()) >= 20000
    # re-scores after toggles subtract the old tokens of the selected chunks: bins move down and wrap downwards
    s, old = _scorer({"TOKSCORE_SCAN": "tile"})
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        s.score()
    finally:
        _restore(old)
    by_count = np.argsort(exp[0][2:len(tokens) + 2])[::-1]
    hot = [int(k) for k in by_count if len(tokens[int(k)].decode("utf-8")) > 1][:3]
    assert kind != "wiki" or len(tokens) - 1 in hot   # "zqzq": without it the same text is twice as many "zq"
    assert hot
    for i in hot:
        s.toggle(i, False)
    got = s.score(incremental=True)
    sub = [b"\x00DISABLED" + str(k).encode() if k in hot else t for k, t in enumerate(tokens)]
    exp2 = O.OracleVocab(*synth.pack_vocab(sub)).score(corpus, off, threads=8)
    assert got[1:] == exp2[1:] and np.array_equal(got[0], exp2[0])
    for i in hot:
        s.toggle(i, True)
    got = s.score(incremental=True)
    assert got[1:] == exp[1:] and np.array_equal(got[0], exp[0])
    s.close()


def test_deep_128k_vocabulary_on_random_utf8():
    # SURVEY section 8 config 4 at test size: random UTF-8 over a 4k alphabet, V = 128k tokens of 2..10 code
    # points cut from the corpus itself (deep trie, most of it outside LDS, four histogram id ranges)
    corpus, off = synth.random_utf8(3_000_000, seed=355, doc_lo=4096, doc_hi=400_000)
    tokens = synth.sample_vocab(corpus, 128_000)
    assert len(tokens) > 128_000
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    exp = voc.score(corpus, off, threads=8)
    # ... and the automaton's failure form (3 M slots, half the lookups beyond LDS: never chosen for it, exact all the same)
    for cfg in ({"TOKSCORE_SCAN": "tile"}, {"TOKSCORE_SCAN": "generic"}, {"TOKSCORE_DFA_FORM": "2"}):
        (counts, tt, tb), info = _score_with(cfg, tokens, corpus, off)
        assert (tt, tb) == exp[1:], cfg
        assert np.array_equal(counts, exp[0]), cfg
        if "TOKSCORE_DFA_FORM" in cfg:
            assert info.dfa_active == 1 and info.dfa_form == 2 and info.scan_path == 2


def test_one_workgroup_sees_more_than_a_16_bit_row_holds():
    # stream scan: slab rows are int16; bins that do not fit are booked directly, whichever way they count
    tokens = [b"a", b"b", b"ab", b"abab", b"c"]
    docs = [b"ab" * 70_000 + b"c", b"b" * 40_000, b"abab" * 3, b""]
    corpus, off = synth.pack_docs(docs)
    exp = O.OracleVocab(*synth.pack_vocab(tokens)).score(corpus, off)
    assert int(exp[0].max()) > 32_767
    (counts, tt, tb), info = _score_with({"TOKSCORE_SCAN": "stream", "TOKSCORE_SLAB_BLOCKS": "1"}, tokens, corpus, off)
    assert info.scan_path == 2
    assert (tt, tb) == exp[1:] and np.array_equal(counts, exp[0])
    # and the subtracting pass of a re-score: "abab" off makes 70,000 "ab" of 35,000 "abab"
    s, old = _scorer({"TOKSCORE_SCAN": "stream", "TOKSCORE_SLAB_BLOCKS": "1"})
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        s.score()
        s.toggle(3, False)
        got = s.score(incremental=True)
    finally:
        _restore(old)
    sub = [t if k != 3 else b"\x00OFF" for k, t in enumerate(tokens)]
    exp2 = O.OracleVocab(*synth.pack_vocab(sub)).score(corpus, off)
    assert got[1:] == exp2[1:] and np.array_equal(got[0], exp2[0])
    s.close()


def test_empty_and_tiny_corpora():
    tokens = [b"a", b"b", b"ab"]
    for docs in ([], [b""], [b"", b"", b""], [b"a"], [b"", b"ab", b"", b"zab"]):
        corpus, off = synth.pack_docs(docs)
        voc = O.OracleVocab(*synth.pack_vocab(tokens))
        exp = voc.score(corpus, off)
        for cfg in ({}, {"TOKSCORE_SCAN": "tile"}, {"TOKSCORE_SCAN": "generic"}):
            (counts, tt, tb), _ = _score_with(cfg, tokens, corpus, off)
            assert (tt, tb) == exp[1:], (docs, cfg)
            assert np.array_equal(counts, exp[0]), (docs, cfg)


def test_long_tokens_fall_back_to_generic_scan():
    # a token longer than any lane segment cannot use the LDS tile kernel
    long_tok = b"x" * 1500
    tokens = [b"x", b"y", long_tok, b"xy"]
    docs = [b"x" * 4000 + b"y" + b"x" * 1499 + b"yxy", b"xyxy"]
    corpus, off = synth.pack_docs(docs)
    voc = O.OracleVocab(*synth.pack_vocab(tokens))
    exp = voc.score(corpus, off)
    (counts, tt, tb), info = _score_with({}, tokens, corpus, off)
    assert info.base_scan_path == 0  # (the automaton scan, which has no such limit, serves the full evaluations all the same)
    assert (tt, tb) == exp[1:] and np.array_equal(counts, exp[0])


@pytest.mark.parametrize("path", ["stream", "tile"])
def test_toggle_equals_full_rescore_of_the_toggled_vocabulary(path):
    corpus, off = synth.cjk_heavy(400_000, seed=9)
    tokens = synth.build_vocab(corpus[:200_000], 9000)
    V = len(tokens)
    s, old = _scorer({"TOKSCORE_SCAN": path})
    s.set_vocab(*synth.pack_vocab(tokens))
    s.set_corpus(corpus, off)
    base = s.score()
    _restore(old)
    assert s.info().scan_path == (2 if path == "stream" else 1)
    rng = np.random.default_rng(5)
    enabled = np.ones(V, dtype=bool)
    multi = [i for i, t in enumerate(tokens) if len(t.decode("utf-8")) > 1]
    assert multi, "vocabulary has no multi-character tokens"
    for step in range(12):
        i = int(rng.choice(multi))
        enabled[i] = not enabled[i]
        s.toggle(i, bool(enabled[i]))
        got = s.score(incremental=True)
        # reference semantics of a disabled entry: a string that can never match
        # (upper-case survives in the vocabulary but never in the lower-cased input)
        sub = [t if enabled[k] else b"\x00DISABLED" + str(k).encode() for k, t in enumerate(tokens)]
        voc = O.OracleVocab(*synth.pack_vocab(sub))
        exp = voc.score(corpus, off, threads=8)
        assert got[1:] == exp[1:], step
        assert np.array_equal(got[0], exp[0]), step
    for i in np.nonzero(~enabled)[0].tolist():
        s.toggle(i, True)
    again = s.score()
    assert np.array_equal(again[0], base[0]) and again[1:] == base[1:]
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ascii_caps", "dup_tokens", "nul_doc", "cjk_unknown", "invalid_utf8"])
def test_reference_flow_through_file_mirrors(case):
    """main.cpp:26-46 as the reference writes it: SolutionFile + DataFile + Tokenize per
    article, against what the compiled reference printed for the same two files."""
    from tokenizer_amd import files
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case)
    tkn = files.SolutionFile(os.path.join(d, "tokens.json"))
    test = files.DataFile(os.path.join(d, "shard.json"))
    assert tkn.IsValid() and test.IsValid()
    exp = json.load(open(os.path.join(d, "expected.json")))
    init_size = comp_size = 0
    for article, ids in zip(test.GetEntries(), exp["ids"]):
        init_size += len(article.text)
        got = tkn.Tokenize(article.text)
        comp_size += len(got) - 2
        assert got == [files.UNKNOWN if i < 0 else i for i in ids]
    assert init_size == exp["total_bytes"] and comp_size == exp["total_tokens"]
    counts, tt, tb = files.score_shard(tkn, test)
    assert (tt, tb) == (exp["total_tokens"], exp["total_bytes"])
    for k, v in exp["counts"].items():
        idx = len(tkn.tokens) + 2 if int(k) < 0 else int(k)
        assert int(counts[idx]) == v
    assert int(counts.sum()) == sum(exp["counts"].values())
    tkn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["stream", "tile"])
@pytest.mark.parametrize("index_mb", ["2048", "0"])  # with the occurrence index / by searching the corpus
def test_incremental_rescore_paths(index_mb, path):
    """tokscore_score_incremental: several toggles per re-score, a token that occurs nearly
    everywhere (falls back to a full evaluation), a tokenize call between toggle and re-score,
    re-score without toggles -- always equal to a full evaluation of the toggled vocabulary."""
    corpus, off = synth.wiki_like(3_000_000, seed=21)
    tokens = synth.build_vocab(corpus[:600_000], 6000)
    tokens.append(bytes(corpus[1_234_560:1_234_572]).lower())   # a string that occurs in very few chunks
    assert tokens[-1] not in tokens[:-1]
    V = len(tokens)

    def expect(enabled):
        sub = [t if enabled[k] else b"\x00DISABLED" + str(k).encode() for k, t in enumerate(tokens)]
        return O.OracleVocab(*synth.pack_vocab(sub)).score(corpus, off, threads=8)

    def same(a, b):
        return np.array_equal(a[0], b[0]) and a[1:] == b[1:]

    old_env = os.environ.get("TOKSCORE_INDEX_MB")
    os.environ["TOKSCORE_INDEX_MB"] = index_mb
    s, old_scan = _scorer({"TOKSCORE_SCAN": path})
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        enabled = np.ones(V, dtype=bool)
        base = s.score()
        _restore(old_scan)
        assert s.info().scan_path == (2 if path == "stream" else 1)
        assert same(base, expect(enabled))
        assert same(s.score(incremental=True), base)              # nothing toggled
        rare = V - 1
        enabled[rare] = False
        s.toggle(rare, False)
        assert same(s.score(incremental=True), expect(enabled))
        assert s.info().rescored_chunks < s.info().chunks, "the fast path was expected to be taken"
        rng = np.random.default_rng(3)
        long_tokens = [i for i, t in enumerate(tokens) if len(t) >= 5]
        for step in range(4):                                       # three toggles per re-score
            for i in rng.choice(long_tokens, size=3, replace=False):
                enabled[i] = not enabled[i]
                s.toggle(int(i), bool(enabled[i]))
            assert same(s.score(incremental=True), expect(enabled)), step
        space = tokens.index(b" ") if b" " in tokens else None        # occurs in every chunk
        if space is not None:
            enabled[space] = False
            s.toggle(space, False)
            assert same(s.score(incremental=True), expect(enabled))
            enabled[space] = True
            s.toggle(space, True)
            assert same(s.score(incremental=True), expect(enabled))
        i = long_tokens[0]                                          # tokenize in between
        enabled[i] = not enabled[i]
        s.toggle(i, bool(enabled[i]))
        ids = s.tokenize(bytes(corpus[:2000]))
        sub = [t if enabled[k] else b"\x00DISABLED" + str(k).encode() for k, t in enumerate(tokens)]
        assert ids == O.OracleVocab(*synth.pack_vocab(sub)).tokenize(bytes(corpus[:2000]))
        assert same(s.score(incremental=True), expect(enabled))
        j = long_tokens[1]                                          # toggled there and back before a re-score
        s.toggle(j, not enabled[j])
        s.toggle(j, bool(enabled[j]))
        assert same(s.score(incremental=True), expect(enabled))
        assert same(s.score(), expect(enabled))
    finally:
        s.close()
        if old_env is None:
            os.environ.pop("TOKSCORE_INDEX_MB", None)
        else:
            os.environ["TOKSCORE_INDEX_MB"] = old_env


class _DeviceBuffer:
    """A device buffer without torch (whose bundled HIP runtime must not be loaded second into a
    process that already runs libtokscore.so's): hipMalloc / hipMemcpy through ctypes."""

    def __init__(self, nbytes):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0

    def to_host(self):
        host = np.zeros(self.nbytes // 8, dtype=np.uint64)
        assert self.hip.hipMemcpy(host.ctypes.data_as(self.C.c_void_p), self.ptr, self.C.c_size_t(self.nbytes), 2) == 0  # D2H
        return host

    def free(self):
        self.hip.hipFree(self.ptr)


def test_stream_overflow_is_reported_or_rescued():
    """The tile scan's token stream has a capacity (the stream scan's cannot be exceeded).  With the
    capacity shrunk (TOKSCORE_STREAM_DIV) the periodic golden overflows it: the synchronous call
    must redo the evaluation on the generic path and still be exact, the asynchronous one must
    raise the flag tokscore_async_status reads -- never a silently short result."""
    g = O.load_golden("periodic")
    corpus, off = synth.pack_docs(g["docs"])
    exp = g["expected"]
    V = len(g["tokens"])
    s, old = _scorer({"TOKSCORE_SCAN": "tile", "TOKSCORE_STREAM_DIV": "4000"})
    out = None
    try:
        s.set_vocab(*synth.pack_vocab(g["tokens"]))
        s.set_corpus(corpus, off)
        out = _DeviceBuffer((V + 5) * 8)
        s.score_async(out.ptr.value)  # on the context's own stream
        assert s.info().scan_path == 1
        assert s.async_status() is True, "an overflowed asynchronous evaluation must say so"
        counts, tt, tb = s.score()  # falls back by itself
        assert s.info().scan_path == 0
        assert (tt, tb) == (exp["total_tokens"], exp["total_bytes"])
        assert np.array_equal(counts, exp["counts_vec"])
        s.score_async(out.ptr.value)
        assert s.async_status() is False
        got = out.to_host()
        assert np.array_equal(got[:V + 3], exp["counts_vec"]) and int(got[V + 3]) == exp["total_tokens"]
    finally:
        _restore(old)
        s.close()
        if out:
            out.free()
    # the stream scan on the same input: no capacity to exceed
    s, old = _scorer({"TOKSCORE_SCAN": "stream", "TOKSCORE_STREAM_DIV": "4000"})
    out = None
    try:
        s.set_vocab(*synth.pack_vocab(g["tokens"]))
        s.set_corpus(corpus, off)
        out = _DeviceBuffer((V + 5) * 8)
        s.score_async(out.ptr.value)
        assert s.async_status() is False
        got = out.to_host()
        assert np.array_equal(got[:V + 3], exp["counts_vec"])
    finally:
        _restore(old)
        s.close()
        if out:
            out.free()


@pytest.mark.parametrize("path", ["stream", "tile"])
def test_sampled_timing_marks_every_nth_evaluation_and_changes_no_result(path):
    """tokscore_set_timing: mode 2 records the two events around the scan kernel on every n-th full
    evaluation only; the results are what they are with every kernel marked, or with none."""
    corpus, off = synth.wiki_like(600_000, seed=21)
    tokens = synth.build_vocab(corpus[:300_000], 3000)
    V = len(tokens)
    exp = O.OracleVocab(*synth.pack_vocab(tokens)).score(corpus, off, threads=8)
    s, old = _scorer({"TOKSCORE_SCAN": path})
    out = None
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        out = _DeviceBuffer((V + 5) * 8)
        s.score_async(out.ptr.value)
        assert s.async_status() is False
        s.drain_timings()
        for mode, every, marked in ((2, 4, 3), (0, 1, 0), (1, 1, 10), (2, 1, 10)):
            s.set_timing(mode, every)
            for _ in range(10):
                s.score_async(out.ptr.value)
            assert s.async_status() is False
            got = out.to_host()
            assert np.array_equal(got[:V + 3], exp[0]) and int(got[V + 3]) == exp[1] and int(got[V + 4]) == exp[2]
            tm = s.drain_timings()
            assert len(tm) == marked, (mode, every, len(tm))
            for t in tm:
                assert t[1] > 0
                if mode == 2:
                    assert t[0] == t[1] and t[2] == 0 and t[3] == 0
                else:
                    assert t[0] >= t[1]
    finally:
        _restore(old)
        s.close()
        if out:
            out.free()


@pytest.mark.gpu
def test_full_evaluations_between_other_users_of_the_stream_state():
    """A full evaluation over ranges that follows another one has no prepare launch (the scan clears
    the output vector, the previous reduction left the cursor words ready); anything else that uses
    the token stream in between -- occurrence counting, toggles with incremental re-scores, batched
    moves -- must put the next evaluation back on the prepared path.  Two output buffers in turn:
    each holds a whole result whatever it held before."""
    corpus, off = synth.wiki_like(1_500_000, seed=33)
    tokens = synth.build_vocab(corpus[:400_000], 4000)
    V = len(tokens)
    ora = O.OracleVocab(*synth.pack_vocab(tokens))
    exp = ora.score(corpus, off, threads=8)
    s, old = _scorer({"TOKSCORE_SCAN": "stream"})
    a = b = None
    try:
        s.set_vocab(*synth.pack_vocab(tokens))
        s.set_corpus(corpus, off)
        a, b = _DeviceBuffer((V + 5) * 8), _DeviceBuffer((V + 5) * 8)

        def check(buf):
            assert s.async_status() is False  # (waits for the context's stream)
            got = buf.to_host()
            assert np.array_equal(got[:V + 3], exp[0]) and int(got[V + 3]) == exp[1] and int(got[V + 4]) == exp[2]

        for buf in (a, b, a, a, b):
            s.score_async(buf.ptr.value)
            check(buf)
        counts, _ = s.count_occurrences(6)
        assert np.array_equal(np.asarray(counts), np.asarray(ora.count_occurrences(corpus, off, 6)[0]))
        for buf in (b, a):
            s.score_async(buf.ptr.value)
            check(buf)
        pick = [i for i, t in enumerate(tokens) if len(t) >= 3][:3]
        for i in pick:  # off and on again through the incremental path, a full evaluation after each
            s.toggle(i, False)
            s.score(incremental=True)
            s.toggle(i, True)
            r = s.score(incremental=True)
            assert r[1] == exp[1]
            s.score_async(a.ptr.value)
            check(a)
        s.score_toggles(pick, [False] * len(pick))
        for buf in (a, b):
            s.score_async(buf.ptr.value)
            check(buf)
        r = s.score()
        assert np.array_equal(r[0], exp[0]) and r[1] == exp[1]
    finally:
        _restore(old)
        s.close()
        for buf in (a, b):
            if buf:
                buf.free()


@pytest.mark.gpu
def test_bench_exchange_runs_over_rccl_on_one_rank():
    """bench.py's N > 1 exchange (score_async, then all_reduce of the u64 vector on the same stream,
    barrier, max over ranks) executed with RCCL on the hardware that is there: a communicator of one
    rank (BENCH_RCCL_SELFTEST=1).  The line must still carry an exact in-run parity check."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_RCCL_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mb", "8", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "RCCL SELFTEST" in line["data"] and line["n_gpus"] == 1
    assert line["parity_checked"]["exact"] is True and line["parity_checked"]["bytes"] == 8_000_000 or line["parity_checked"]["exact"] is True
