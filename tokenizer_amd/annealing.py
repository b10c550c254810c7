"""Host-side mirror of the reference's `annealing` namespace
over libtokanneal.so (include/tokanneal.h):

    GetTokens / ReadTokens / WriteTokens  -- src/tokenizer/GetTokens.cpp:92-115, Token.cpp:109-193
    Tokens                                -- std::vector<annealing::Token> (Token.h:38-93)
    TokenGenerator                        -- src/tokenizer/TokenGenerator.{h,cpp}
    anneal_and_score                      -- src/main.cpp:20-46: candidates -> annealing -> solution
                                             -> .tokens.json -> score of a shard (on the GPU)

Same names and argument meaning as the reference.  All arithmetic runs in the C++ library; this
file only moves buffers.  The annealing loop is host code, as in the reference; the corpus
score of its result comes from the GPU scorer (tokenizer_amd.Scorer).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtokanneal.so")

OK = 0
E_ARG, E_IO, E_DATA, E_SPACE = 1, 2, 3, 4
NONE = 0xFFFFFFFF  # "no parent" (Token.cpp:187-190)
# annealing::TokenReadErrCode (Token.h:13-19)
READ_OK, BAD_STREAM, INVALID_VERSION, BAD_FILE, NOT_DONE = range(5)

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class State(C.Structure):
    """tokanneal_state: what Generate prints after a pass (TokenGenerator.cpp:184-185) + the fit."""
    _fields_ = [("gen_cnt", C.c_uint64), ("enabled_cnt", C.c_uint64), ("tot_cand", C.c_uint64),
                ("raw_score", C.c_double), ("score", C.c_double), ("temp", C.c_double),
                ("moment1", C.c_double), ("moment2", C.c_double), ("beta", C.c_double), ("sigma", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
# every symbol include/tokanneal.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ("tokanneal_abi_version", C.c_int, []),
    ("tokanneal_strerror", C.c_char_p, [C.c_int]),
    ("tokanneal_cands_build", C.c_int, [_vpp, _u8p, _u64p, C.c_uint64, C.c_uint32]),
    ("tokanneal_cands_from_arrays", C.c_int, [_vpp, C.c_uint64, _u32p, _u64p, _u32p, _u32p]),
    ("tokanneal_cands_merge", C.c_int, [_vpp, C.c_void_p, C.c_void_p]),
    ("tokanneal_cands_read", C.c_int, [_vpp, C.c_char_p, C.POINTER(C.c_int)]),
    ("tokanneal_cands_write", C.c_int, [_vp, C.c_char_p]),
    ("tokanneal_cands_size", C.c_uint64, [_vp]),
    ("tokanneal_cands_arrays", C.c_int, [_vp, _u32p, _u64p, _u32p, _u32p]),
    ("tokanneal_cands_token_size", C.c_uint64, [_vp, C.c_uint64]),
    ("tokanneal_cands_name", C.c_int, [_vp, C.c_uint64, _u8p, C.c_uint64, _u64p]),
    ("tokanneal_cands_names", C.c_int, [_vp, _u8p, C.c_uint64, _u64p, _u64p]),
    ("tokanneal_cands_free", None, [_vp]),
    ("tokanneal_gen_create", C.c_int, [_vpp, _vpp, C.c_uint64, C.c_uint64]),
    ("tokanneal_gen_free", None, [_vp]),
    ("tokanneal_simulate_step", C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_double)]),
    ("tokanneal_apply_step", C.c_int, [_vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    ("tokanneal_token_state", C.c_int, [_vp, C.c_uint64, _u64p, _u64p, C.POINTER(C.c_int)]),
    ("tokanneal_seed", C.c_int, [_vp, C.c_uint32]),
    ("tokanneal_set_evaluator", C.c_int, [_vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("tokanneal_worker_task", C.c_int, [_vp]),
    ("tokanneal_generate", C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_int64, C.POINTER(State), C.c_uint64, _u64p]),
    ("tokanneal_get_state", C.c_int, [_vp, C.POINTER(State)]),
    ("tokanneal_solution", C.c_int, [_vp, _u8p, C.c_uint64, _u64p, C.c_uint64, _u64p, _u64p]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]
# include/tokfiles.h (the shard reader lives in the same library; files.ShardFile is its mirror)
_FILES_SIGNATURES = [
    ("tokfiles_shard_read", C.c_int, [_vpp, C.c_char_p, C.POINTER(C.c_int)]),
    ("tokfiles_shard_docs", C.c_uint64, [_vp]),
    ("tokfiles_shard_bytes", C.c_uint64, [_vp, C.c_int]),
    ("tokfiles_shard_copy", C.c_int, [_vp, C.c_int, _u8p, _u64p]),
    ("tokfiles_shard_free", None, [_vp]),
]

_lib = None


class TokannealError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("tokanneal status %d: %s" % (status, message))
        self.status = status


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C tokenizer_amd/csrc`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, res, args in _SIGNATURES + _FILES_SIGNATURES:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(st):
    if st != OK:
        raise TokannealError(st, load_library().tokanneal_strerror(st).decode())


def _unpack(bytes_, off):
    b = bytes_.tobytes()
    return [b[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


class Tokens:
    """The candidate set: std::vector<annealing::Token>.  Built by GetTokens / ReadTokens /
    from_arrays; TokenGenerator takes it over (it is empty afterwards, like the moved-from vector)."""

    def __init__(self, handle):
        self._lib = load_library()
        self._h = C.c_void_p(handle)

    @classmethod
    def from_arrays(cls, chr_, uses, lpar, rpar):
        lib = load_library()
        a = [np.ascontiguousarray(chr_, dtype=np.uint32), np.ascontiguousarray(uses, dtype=np.uint64),
             np.ascontiguousarray(lpar, dtype=np.uint32), np.ascontiguousarray(rpar, dtype=np.uint32)]
        h = C.c_void_p()
        _check(lib.tokanneal_cands_from_arrays(C.byref(h), len(a[0]), a[0].ctypes.data_as(_u32p), a[1].ctypes.data_as(_u64p),
                                               a[2].ctypes.data_as(_u32p), a[3].ctypes.data_as(_u32p)))
        return cls(h.value)

    def merge(self, other: "Tokens") -> "Tokens":
        """The candidate set of both corpora together (each string once, uses added up, the
        reference's order): what GetTokens builds over several files (GetTokens.cpp:31-90)."""
        h = C.c_void_p()
        _check(self._lib.tokanneal_cands_merge(C.byref(h), self._h, other._h))
        return Tokens(h.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tokanneal_cands_free(self._h)
            self._h = None

    __del__ = close

    def __len__(self):
        return int(self._lib.tokanneal_cands_size(self._h)) if self._h else 0

    def arrays(self):
        """-> (chr, uses, l_parent, r_parent); chr is the FIRST code point of the string (Trie.cpp:81-89)."""
        n = len(self)
        chr_ = np.zeros(n, dtype=np.uint32)
        uses = np.zeros(n, dtype=np.uint64)
        lpar = np.zeros(n, dtype=np.uint32)
        rpar = np.zeros(n, dtype=np.uint32)
        if n:
            _check(self._lib.tokanneal_cands_arrays(self._h, chr_.ctypes.data_as(_u32p), uses.ctypes.data_as(_u64p),
                                                    lpar.ctypes.data_as(_u32p), rpar.ctypes.data_as(_u32p)))
        return chr_, uses, lpar, rpar

    def size(self, i: int) -> int:  # Token::size() (Token.cpp:56-64)
        return int(self._lib.tokanneal_cands_token_size(self._h, i))

    def GetName(self, i: int) -> bytes:  # Token::GetName() (Token.cpp:65-74)
        buf = (C.c_uint8 * 1024)()
        n = C.c_uint64()
        _check(self._lib.tokanneal_cands_name(self._h, i, buf, 1024, C.byref(n)))
        return bytes(buf[:n.value])

    def names(self):
        need = C.c_uint64()
        _check(self._lib.tokanneal_cands_names(self._h, None, 0, None, C.byref(need)))
        b = np.zeros(max(1, need.value), dtype=np.uint8)
        off = np.zeros(len(self) + 1, dtype=np.uint64)
        _check(self._lib.tokanneal_cands_names(self._h, b.ctypes.data_as(_u8p), len(b), off.ctypes.data_as(_u64p), None))
        return _unpack(b, off)


def GetTokens(corpus: np.ndarray, doc_off: np.ndarray, max_len: int = 255) -> Tokens:
    """annealing::GetTokens(metadata, max_len, ..., rebuild=true) (GetTokens.cpp:92-115) without
    the file walk: FileCandidates + Trie::BuildTokens over documents packed back to back (what
    DataFile.packed() returns).  max_len is in code points, as there."""
    lib = load_library()
    b = np.ascontiguousarray(corpus, dtype=np.uint8)
    if b.size == 0:
        b = np.zeros(1, dtype=np.uint8)
    o = np.ascontiguousarray(doc_off, dtype=np.uint64)
    h = C.c_void_p()
    _check(lib.tokanneal_cands_build(C.byref(h), b.ctypes.data_as(_u8p), o.ctypes.data_as(_u64p), len(o) - 1, max_len))
    return Tokens(h.value)


def GetTokensGpu(corpus, doc_off=None, max_len: int = 255, device: int = 0, group_bytes: int = 1 << 30) -> Tokens:
    """The same candidate set as GetTokens, computed on the GPU (tokscore_extract_candidates: one
    sort of all windows instead of a trie; max_len <= 32, valid UTF-8 without NUL bytes).
    `corpus` may also be a list of (bytes, offsets) shards, like the files of one metadata.json.
    Documents are independent and the extraction is one sort of all their windows, so shards are not
    merged: as many as fit `group_bytes` (below 4 GiB: 32-bit positions; about 80 bytes of HBM per
    corpus byte while it runs) are put behind one another and extracted TOGETHER -- the sort does the
    merging, on the GPU.  Only the sets of different groups go through the host's merge
    (tokanneal_cands_merge, 28 M tokens/s)."""
    from . import Scorer
    shards = [(corpus, doc_off)] if doc_off is not None else list(corpus)
    group_bytes = max(1, min(int(group_bytes), (1 << 32) - (1 << 24)))
    groups, cur, cur_bytes = [], [], 0
    for b, o in shards:
        b = np.ascontiguousarray(b, dtype=np.uint8)
        o = np.ascontiguousarray(o, dtype=np.uint64)
        n = int(o[-1] - o[0])
        if cur and cur_bytes + n > group_bytes:
            groups.append(cur)
            cur, cur_bytes = [], 0
        cur.append((b, o))
        cur_bytes += n
    if cur:
        groups.append(cur)
    s = Scorer(device)
    total = None
    try:
        for g in groups:
            if len(g) == 1:
                b, o = g[0]
            else:  # one corpus: the shards' bytes behind one another, the offsets shifted
                b = np.concatenate([x[int(y[0]):int(y[-1])] for x, y in g])
                parts, base = [np.zeros(1, dtype=np.uint64)], 0
                for _, y in g:
                    parts.append((y[1:] - y[0]) + np.uint64(base))
                    base += int(y[-1] - y[0])
                o = np.concatenate(parts)
            s.set_corpus(b, o)
            chr_, uses, lpar, rpar, _ = s.extract_candidates(max_len)
            part = Tokens.from_arrays(chr_, uses, lpar, rpar)
            if total is None:
                total = part
            else:
                merged = total.merge(part)
                total.close()
                part.close()
                total = merged
    finally:
        s.close()
    return total if total is not None else Tokens.from_arrays([], [], [], [])


def candidates_file_name(file_cnt=None, max_len: int = 255) -> str:
    """The cache file GetTokens uses under the corpus root (GetTokens.cpp:94-96)."""
    return ".candidates-" + ("all" if file_cnt is None else str(file_cnt)) + ("" if max_len == 255 else "-%d" % max_len) + ".bin"


def ReadTokens(path):
    """annealing::ReadTokens (Token.cpp:109-161) -> (TokenReadErrCode, Tokens or None)."""
    lib = load_library()
    h = C.c_void_p()
    code = C.c_int(BAD_STREAM)
    st = lib.tokanneal_cands_read(C.byref(h), os.fsencode(path), C.byref(code))
    if st != OK:
        return code.value, None
    return READ_OK, Tokens(h.value)


def WriteTokens(path, tokens: Tokens):
    """annealing::WriteTokens (Token.cpp:163-193), byte for byte."""
    _check(load_library().tokanneal_cands_write(tokens._h, os.fsencode(path)))


_EVALUATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _u32p, C.c_uint64, C.c_int, C.POINTER(C.c_double))
_APPLIED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_int)
_RESYNC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))


class GpuMoveEvaluator:
    """Judges the annealer's moves on a corpus resident on the GPU instead of on the candidate DAG:
    a Scorer whose vocabulary is the whole candidate set (entry i = candidate i), with exactly the
    generator's enabled candidates enabled.  The value of a move is its exact effect on the corpus'
    token count (Scorer.score_toggles, one call per batch); the raw score is the tokens saved
    against the single-code-point vocabulary the generator starts from."""

    def __init__(self, tokens: Tokens, corpus, doc_off, device: int = 0):
        from . import Scorer
        names = tokens.names()
        _, _, _, rpar = tokens.arrays()
        self.scorer = Scorer(device)
        b = np.frombuffer(b"".join(names), dtype=np.uint8) if names else np.zeros(0, dtype=np.uint8)
        off = np.zeros(len(names) + 1, dtype=np.uint64)
        np.cumsum([len(n) for n in names], out=off[1:])
        self.scorer.set_vocab(b, off)
        self.scorer.set_corpus(corpus, doc_off)
        for i in np.nonzero(rpar != NONE)[0]:  # everything but the single code points starts disabled (TokenGenerator.cpp:27-48)
            self.scorer.toggle(int(i), False)
        self.start_tokens = self.scorer.score()[1]
        self.total_tokens = self.start_tokens
        self.calls = 0
        self.moves = 0

    def evaluate(self, cands, enable: bool):
        d = self.scorer.score_toggles(cands, 1 if enable else 0).astype(np.float64)
        self.calls += 1
        self.moves += len(cands)
        return -d if enable else d

    def applied(self, cand: int, enable: bool):
        self.scorer.toggle(int(cand), bool(enable))

    def resync(self) -> float:
        self.total_tokens = self.scorer.score(incremental=True)[1]
        return float(self.start_tokens - self.total_tokens)

    def close(self):
        self.scorer.close()


class TokenGenerator:
    """annealing::TokenGenerator (TokenGenerator.h:15-53).  `tokens` is consumed.
    evaluator: None = the reference's estimate on the candidate DAG (Token::SimulateStep);
    an object with evaluate(cands, enable) -> values, applied(cand, enable), resync() -> raw score
    (GpuMoveEvaluator) = moves judged by it, a batch per call (include/tokanneal.h)."""

    def __init__(self, tokens: Tokens, pref_token_count: int, batch_size: int = None, evaluator=None):
        self._lib = load_library()
        if batch_size is None:  # the reference's default argument (TokenGenerator.h:49)
            batch_size = os.cpu_count() or 1
        self._h = C.c_void_p()
        _check(self._lib.tokanneal_gen_create(C.byref(self._h), C.byref(tokens._h), pref_token_count, batch_size))
        tokens._h = None
        self._evaluator = None
        self._callback_error = None
        if evaluator is not None:
            self.set_evaluator(evaluator)

    def set_evaluator(self, evaluator):
        self._evaluator = evaluator
        if evaluator is None:
            self._callbacks = None
            _check(self._lib.tokanneal_set_evaluator(self._h, None, None, None, None))
            return

        def guard(f):
            def run(*a):
                try:
                    f(*a)
                    return 0
                except BaseException as e:  # noqa: B902 -- must not unwind through the C frames
                    self._callback_error = e
                    return 1
            return run

        def evaluate(_user, cand, n, enable, out):
            vals = evaluator.evaluate(np.ctypeslib.as_array(cand, shape=(int(n),)).copy(), bool(enable))
            for k in range(int(n)):
                out[k] = float(vals[k])

        def applied(_user, cand, enable):
            evaluator.applied(int(cand), bool(enable))

        def resync(_user, raw):
            raw[0] = float(evaluator.resync())

        self._callbacks = (_EVALUATE_FN(guard(evaluate)),
                           _APPLIED_FN(guard(applied)) if hasattr(evaluator, "applied") else None,
                           _RESYNC_FN(guard(resync)) if hasattr(evaluator, "resync") else None)
        _check(self._lib.tokanneal_set_evaluator(self._h, *[C.cast(f, C.c_void_p) if f else None for f in self._callbacks], None))

    def _checked(self, status):
        err, self._callback_error = self._callback_error, None
        if err is not None:
            raise err
        _check(status)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tokanneal_gen_free(self._h)
            self._h = None

    __del__ = close

    # Token::SimulateStep / Token::ApplyStep<Enable> (Token.cpp:83-105) of token i
    def SimulateStep(self, i: int) -> float:
        d = C.c_double()
        _check(self._lib.tokanneal_simulate_step(self._h, i, C.byref(d)))
        return d.value

    def ApplyStep(self, i: int, enable: bool) -> float:
        d = C.c_double()
        _check(self._lib.tokanneal_apply_step(self._h, i, 1 if enable else 0, C.byref(d)))
        return d.value

    def token_state(self, i: int):
        """-> (l uses, r uses, enabled) of token i as they stand."""
        lu, ru, en = C.c_uint64(), C.c_uint64(), C.c_int()
        _check(self._lib.tokanneal_token_state(self._h, i, C.byref(lu), C.byref(ru), C.byref(en)))
        return int(lu.value), int(ru.value), bool(en.value)

    def seed(self, seed: int):
        """Deterministic seeding of the generator WorkerTask draws from on this thread."""
        _check(self._lib.tokanneal_seed(self._h, seed & 0xFFFFFFFF))

    def WorkerTask(self):  # TokenGenerator.cpp:107-158
        self._checked(self._lib.tokanneal_worker_task(self._h))

    def state(self) -> State:
        s = State()
        _check(self._lib.tokanneal_get_state(self._h, C.byref(s)))
        return s

    def Generate(self, pass_cnt: int, threads: int = 0, seed: int = -1):
        """TokenGenerator::Generate(pass_cnt) (TokenGenerator.cpp:174-188): pass_cnt + 1 passes.
        -> the state after every pass.  threads = 0: as many workers as the reference starts;
        seed >= 0 with threads = 1 is reproducible."""
        states = (State * (pass_cnt + 1))()
        n = C.c_uint64()
        self._checked(self._lib.tokanneal_generate(self._h, pass_cnt, threads, seed, states, pass_cnt + 1, C.byref(n)))
        return [states[i] for i in range(int(n.value))]

    def GetSolution(self):
        """TokenGenerator::GetSolution (TokenGenerator.cpp:190-217) -> list of token byte strings."""
        return _unpack(*self.solution_packed())

    def solution_packed(self):
        """The solution as (bytes, offsets): the exact arguments of Scorer.set_vocab."""
        n, need = C.c_uint64(), C.c_uint64()
        _check(self._lib.tokanneal_solution(self._h, None, 0, None, 0, C.byref(n), C.byref(need)))
        b = np.zeros(max(1, need.value), dtype=np.uint8)
        off = np.zeros(n.value + 1, dtype=np.uint64)
        _check(self._lib.tokanneal_solution(self._h, b.ctypes.data_as(_u8p), len(b), off.ctypes.data_as(_u64p), len(off),
                                            None, None))
        return b[:need.value], off


def anneal_and_score(corpus, doc_off, score_corpus=None, score_off=None, max_len=10, pref_token_count=30000,
                     pass_cnt=20, batch_size=None, threads=0, seed=-1, solution_path=None, device=0,
                     trajectory=None, gpu_candidates=False, evaluator=None):
    """main.cpp:20-46 end to end: GetTokens(max_len) -> TokenGenerator(pref) -> Generate(pass_cnt)
    -> GetSolution -> SolutionFile(solution, path) -> score of a shard.  The shard that is scored
    defaults to the one the candidates came from.  -> (SolutionFile, counts, total_tokens,
    total_bytes); total_bytes / total_tokens is the ratio main.cpp:45 prints.

    trajectory: a list that receives, after every pass, the annealer's own state together with
    what the GPU measures for the solution as it stands ({"pass", "score", "enabled_cnt",
    "total_tokens", "bytes_per_token"}) -- the reference only ever sees its DAG estimate.  The
    passes then run as pass_cnt + 1 calls of Generate(0) (worker k of pass p seeded seed + p + k).
    gpu_candidates: build the candidate set on the GPU (GetTokensGpu) instead of the host.
    evaluator="gpu": every move is judged by its exact effect on the token count of the shard the
    candidates came from (GpuMoveEvaluator) instead of by the DAG estimate (TokenGenerator.cpp:87);
    one worker.  The default is the reference's DAG path."""
    from . import Scorer
    from .files import SolutionFile
    sc, so = (corpus, doc_off) if score_corpus is None else (score_corpus, score_off)
    tokens = GetTokensGpu(corpus, doc_off, max_len, device) if gpu_candidates else GetTokens(corpus, doc_off, max_len)
    judge = None
    if evaluator == "gpu":
        judge = GpuMoveEvaluator(tokens, corpus, doc_off, device)
    elif evaluator is not None:
        raise ValueError("evaluator must be None or 'gpu'")
    gen = TokenGenerator(tokens, pref_token_count, batch_size, evaluator=judge)
    if trajectory is None:
        gen.Generate(pass_cnt, threads, seed)
    else:
        probe = Scorer(device)
        probe.set_corpus(sc, so)
        for p in range(pass_cnt + 1):
            st = gen.Generate(0, threads, seed if seed < 0 else seed + p)[0]
            probe.set_vocab(*gen.solution_packed())
            _, tt, tb = probe.score()
            trajectory.append({"pass": p, "score": st.score, "enabled_cnt": int(st.enabled_cnt), "total_tokens": tt,
                               "bytes_per_token": tb / tt if tt else 0.0})
        probe.close()
    solution = gen.GetSolution()
    gen.close()
    if judge is not None:
        judge.close()
    sol = SolutionFile(solution, solution_path or os.devnull, device=device)
    s = sol.scorer()
    s.set_corpus(sc, so)
    counts, total_tokens, total_bytes = s.score()
    return sol, counts, total_tokens, total_bytes
