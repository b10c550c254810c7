"""tokenizer_amd -- MI355X-native corpus scorer for the annealing tokenizer.

Thin ctypes binding over the C-ABI in include/tokscore.h (libtokscore.so, built in-tree
by tokenizer_amd/csrc/Makefile).  The binding mirrors the reference's scoring seam --
SolutionFile::{GetId, Tokenize} (/root/reference/src/files/SolutionFile.h:19-28) and the
accumulation loop of /root/reference/src/main.cpp:31-46 -- and nothing else.  There is no
CPU fallback: without the HIP library or without a GPU every scoring call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TOKSCORE_LIBRARY: another build of the same library (csrc/Makefile `single`), for build checks
LIB_PATH = os.environ.get("TOKSCORE_LIBRARY") or os.path.join(_HERE, "libtokscore.so")

OK = 0
E_ARG, E_NO_DEVICE, E_HIP, E_STATE, E_VOCAB, E_CAPACITY, E_INTERNAL = 1, 2, 3, 4, 5, 6, 7
UNKNOWN = 2 ** 64 - 1  # (size_t)-1, SolutionFile.cpp:76

_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)


class TokscoreInfo(C.Structure):
    _fields_ = [("vocab_size", C.c_uint64), ("max_len", C.c_uint64), ("trie_nodes", C.c_uint64),
                ("trie_slots", C.c_uint64), ("trie_lds_slots", C.c_uint64), ("corpus_bytes", C.c_uint64),
                ("corpus_docs", C.c_uint64), ("chunks", C.c_uint64), ("slot_bits", C.c_uint32),
                ("id_bits", C.c_uint32), ("scan_path", C.c_uint32), ("rescored_chunks", C.c_uint32),
                ("toggle_streams", C.c_uint64), ("dfa_slots", C.c_uint64), ("dfa_virtual_slots", C.c_uint64),
                ("dfa_lds_slots", C.c_uint32), ("dfa_active", C.c_uint32), ("dfa_chains", C.c_uint32),
                ("dfa_range_bytes", C.c_uint32), ("base_scan_path", C.c_uint32), ("dfa_lds_used", C.c_uint32),
                ("dfa_hot_fraction", C.c_double), ("dfa_build_ms", C.c_double), ("dfa_plan_ms", C.c_double),
                ("dfa_second_hits", C.c_uint64), ("dfa_relayouts", C.c_uint32), ("search_passes", C.c_uint32),
                ("dfa_form", C.c_uint32), ("reserved0", C.c_uint32)]


# every symbol include/tokscore.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ("tokscore_abi_version", C.c_int, []),
    ("tokscore_device_count", C.c_int, []),
    ("tokscore_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    ("tokscore_create_hostonly", C.c_int, [C.POINTER(C.c_void_p)]),
    ("tokscore_destroy", None, [C.c_void_p]),
    ("tokscore_strerror", C.c_char_p, [C.c_int]),
    ("tokscore_last_error", C.c_char_p, [C.c_void_p]),
    ("tokscore_set_vocab", C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint64]),
    ("tokscore_vocab_max_len", C.c_int, [C.c_void_p, _u64p]),
    ("tokscore_toggle", C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
    ("tokscore_set_corpus", C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint64]),
    ("tokscore_set_corpus_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    ("tokscore_score", C.c_int, [C.c_void_p, _u64p, _u64p, _u64p]),
    ("tokscore_score_async", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("tokscore_async_status", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    ("tokscore_score_incremental", C.c_int, [C.c_void_p, _u64p, _u64p, _u64p]),
    ("tokscore_score_toggles", C.c_int, [C.c_void_p, _u64p, C.POINTER(C.c_int32), C.c_uint64, C.POINTER(C.c_int64)]),
    ("tokscore_count_occurrences", C.c_int, [C.c_void_p, C.c_uint32, _u64p, _u64p]),
    ("tokscore_count_occurrences_static", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("tokscore_extract_candidates", C.c_int, [C.c_void_p, C.c_uint32, _u64p, _u64p]),
    ("tokscore_candidates_copy", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), _u64p, C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32)]),
    ("tokscore_tokenize", C.c_int, [C.c_void_p, _u8p, C.c_uint64, _u64p, C.c_uint64, _u64p]),
    ("tokscore_get_info", C.c_int, [C.c_void_p, C.POINTER(TokscoreInfo)]),
    ("tokscore_last_timing", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    ("tokscore_drain_timings", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    ("tokscore_set_timing", C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    ("tokscore_hosttrie_longest", C.c_int, [C.c_void_p, _u8p, C.c_uint64, _u64p, _u64p]),
    ("tokscore_hostsearch_check", C.c_int, [_u8p, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, _u8p, C.c_uint64, _u64p, C.c_uint64, _u64p, _u64p]),
    ("tokscore_hostdfa_check", C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint64, C.c_uint64, C.c_uint32, _u64p, _u64p, _u64p]),
    ("tokscore_probe_read_bandwidth", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


class TokscoreError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("tokscore status %d: %s" % (status, message))
        self.status = status


def load_library():
    """dlopen libtokscore.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C tokenizer_amd/csrc` "
                              "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                              % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, res, args in _SIGNATURES:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def device_count() -> int:
    return int(load_library().tokscore_device_count())


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _p64(a):
    return a.ctypes.data_as(_u64p)


def hostsearch_check(strings, ncodes: int, text: np.ndarray, max_slots: int = 40832):
    """Host self-check of the search automaton of a batch of strings (tokscore_hostsearch_check; no GPU):
    strings = byte strings over the codes 1 .. ncodes - 1, text = uint8 codes.  -> (pairs[n, 2] of (string,
    position of its first byte) in the order found, (LDS slots, states, entries))."""
    lib = load_library()
    codes = np.frombuffer(b"".join(strings), dtype=np.uint8).copy() if strings else np.zeros(0, np.uint8)
    off = np.zeros(len(strings) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(x) for x in strings])
    text = np.ascontiguousarray(text, dtype=np.uint8)
    n_found = C.c_uint64(0)
    stats = np.zeros(3, dtype=np.uint64)
    cap = max(16, 4 * len(text))
    buf = codes if len(codes) else np.zeros(1, np.uint8)
    tbuf = text if len(text) else np.zeros(1, np.uint8)
    while True:
        pairs = np.zeros(2 * cap, dtype=np.uint64)
        rc = lib.tokscore_hostsearch_check(_p8(buf), off.ctypes.data_as(C.POINTER(C.c_uint32)), len(strings), ncodes, max_slots,
                                           _p8(tbuf), len(text), _p64(pairs), cap, C.byref(n_found), _p64(stats))
        if rc != 0:
            raise TokscoreError(rc, "hostsearch_check")
        if n_found.value <= cap:
            break
        cap = n_found.value
    return pairs[: 2 * n_found.value].reshape(-1, 2).astype(np.int64), tuple(int(x) for x in stats)


class Scorer:
    """One GPU, one vocabulary, one resident corpus (a tokscore_ctx)."""

    def __init__(self, device: int = 0, host_only: bool = False):
        self._lib = load_library()
        self._h = C.c_void_p()
        st = self._lib.tokscore_create_hostonly(C.byref(self._h)) if host_only else \
            self._lib.tokscore_create(C.byref(self._h), device)
        if st != OK:
            self._h = None
            why = self._lib.tokscore_last_error(None).decode()
            raise TokscoreError(st, self._lib.tokscore_strerror(st).decode() + (": " + why if why else ""))
        self.V = 0
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tokscore_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, st):
        if st != OK:
            msg = self._lib.tokscore_last_error(self._h).decode() or self._lib.tokscore_strerror(st).decode()
            raise TokscoreError(st, msg)

    # vocabulary -- SolutionFile(tokens, path) id assignment (SolutionFile.cpp:59-72)
    def set_vocab(self, tok_bytes: np.ndarray, tok_off: np.ndarray):
        b = np.ascontiguousarray(tok_bytes, dtype=np.uint8)
        if b.size == 0:
            b = np.zeros(1, dtype=np.uint8)
        o = np.ascontiguousarray(tok_off, dtype=np.uint64)
        self._check(self._lib.tokscore_set_vocab(self._h, _p8(b), _p64(o), len(o) - 1))
        self.V = len(o) - 1

    def set_vocab_list(self, tokens):
        from .synth import pack_vocab
        self.set_vocab(*pack_vocab(tokens))

    @property
    def max_len(self) -> int:
        v = C.c_uint64()
        self._check(self._lib.tokscore_vocab_max_len(self._h, C.byref(v)))
        return int(v.value)

    def toggle(self, token_index: int, enable: bool):
        self._check(self._lib.tokscore_toggle(self._h, token_index, 1 if enable else 0))

    # corpus -- DataFile::GetEntries() packed (DataFile.cpp:24-31)
    def set_corpus(self, corpus: np.ndarray, doc_off: np.ndarray):
        b = np.ascontiguousarray(corpus, dtype=np.uint8)
        if b.size == 0:
            b = np.zeros(1, dtype=np.uint8)
        o = np.ascontiguousarray(doc_off, dtype=np.uint64)
        self._check(self._lib.tokscore_set_corpus(self._h, _p8(b), _p64(o), len(o) - 1))

    def set_corpus_device(self, d_bytes_ptr: int, d_off_ptr: int, n_docs: int, total_bytes: int, keep=None):
        self._keep = [keep]
        self._check(self._lib.tokscore_set_corpus_device(self._h, d_bytes_ptr, d_off_ptr, n_docs, total_bytes))

    # scoring -- main.cpp:38-43 accumulation + histogram of the ids Tokenize emits
    def score(self, incremental: bool = False, out: np.ndarray = None):
        """-> (counts[V+3], total_tokens, total_bytes); `out` (uint64[V+3]) is filled and returned
        when given, so that a loop of moves allocates nothing."""
        counts = np.zeros(self.V + 3, dtype=np.uint64) if out is None else out
        if counts.dtype != np.uint64 or counts.size != self.V + 3 or not counts.flags.c_contiguous:
            raise ValueError("out must be a contiguous uint64 array of V + 3 entries")
        tt = C.c_uint64()
        tb = C.c_uint64()
        fn = self._lib.tokscore_score_incremental if incremental else self._lib.tokscore_score
        self._check(fn(self._h, _p64(counts), C.byref(tt), C.byref(tb)))
        return counts, int(tt.value), int(tb.value)

    def score_async(self, d_out_ptr: int, stream_ptr: int = 0):
        self._check(self._lib.tokscore_score_async(self._h, d_out_ptr, stream_ptr))

    def async_status(self, stream_ptr: int = 0) -> bool:
        """Waits for the stream; True when the last score_async there overflowed its token stream
        (its d_out is incomplete: redo with score(), which falls back by itself)."""
        flag = C.c_int(0)
        self._check(self._lib.tokscore_async_status(self._h, stream_ptr, C.byref(flag)))
        return bool(flag.value)

    def score_toggles(self, token_index, enable) -> np.ndarray:
        """K independent moves judged at once: -> int64[K], total_tokens with entry token_index[k]
        set to enable[k] minus total_tokens of the vocabulary as it is.  Nothing is changed."""
        idx = np.ascontiguousarray(token_index, dtype=np.uint64)
        en = np.ascontiguousarray(np.broadcast_to(np.asarray(enable, dtype=np.int32), idx.shape))
        out = np.zeros(idx.shape[0], dtype=np.int64)
        if idx.shape[0]:
            self._check(self._lib.tokscore_score_toggles(self._h, _p64(idx), en.ctypes.data_as(C.POINTER(C.c_int32)),
                                                         idx.shape[0], out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    # ExtractCandidates + Trie::AddString on a closed vocabulary (GetTokens.cpp:21-27, Trie.cpp:114-128)
    def count_occurrences(self, max_cp: int):
        """-> (counts[V+3] by id, code-point positions): occurrences of every vocabulary string as
        a window of at most max_cp code points, no lower-casing."""
        counts = np.zeros(self.V + 3, dtype=np.uint64)
        pos = C.c_uint64()
        self._check(self._lib.tokscore_count_occurrences(self._h, max_cp, _p64(counts), C.byref(pos)))
        return counts, int(pos.value)

    def count_occurrences_static(self, max_cp: int, d_out_ptr: int, stream_ptr: int = 0) -> None:
        """count_occurrences without any device allocation: d_out_ptr is device memory for V+4 uint64
        (counts[V+3] | positions); enqueued on stream_ptr (0: the context's stream), not waited for.
        Scratch is a 16 MiB array inside the library, shared by every call on the GPU."""
        self._check(self._lib.tokscore_count_occurrences_static(self._h, max_cp, d_out_ptr, stream_ptr))

    # GetTokens: FileCandidates + Trie::BuildTokens (GetTokens.cpp:21-90, Trie.cpp:81-152), on the GPU
    def extract_candidates(self, max_cp: int):
        """-> (chr, uses, l_parent, r_parent, positions): the candidate set of the resident corpus in
        the reference's token order; feed the arrays to annealing.Tokens.from_arrays."""
        n = C.c_uint64()
        pos = C.c_uint64()
        self._check(self._lib.tokscore_extract_candidates(self._h, max_cp, C.byref(n), C.byref(pos)))
        chr_ = np.zeros(n.value, dtype=np.uint32)
        uses = np.zeros(n.value, dtype=np.uint64)
        lpar = np.zeros(n.value, dtype=np.uint32)
        rpar = np.zeros(n.value, dtype=np.uint32)
        u32 = C.POINTER(C.c_uint32)
        self._check(self._lib.tokscore_candidates_copy(self._h, chr_.ctypes.data_as(u32), _p64(uses), lpar.ctypes.data_as(u32),
                                                       rpar.ctypes.data_as(u32)))
        return chr_, uses, lpar, rpar, int(pos.value)

    # SolutionFile::Tokenize (SolutionFile.cpp:83-100); -1 stands for (size_t)-1
    def tokenize(self, text: bytes):
        a = np.frombuffer(text, dtype=np.uint8) if text else np.zeros(1, dtype=np.uint8)
        out = np.zeros(len(text) + 2, dtype=np.uint64)
        n = C.c_uint64()
        self._check(self._lib.tokscore_tokenize(self._h, _p8(a), len(text), _p64(out), len(out), C.byref(n)))
        return [-1 if int(x) == UNKNOWN else int(x) for x in out[:int(n.value)]]

    def info(self) -> TokscoreInfo:
        i = TokscoreInfo()
        self._check(self._lib.tokscore_get_info(self._h, C.byref(i)))
        return i

    def last_timing(self):
        ms = (C.c_float * 4)()
        self._check(self._lib.tokscore_last_timing(self._h, ms))
        return [float(x) for x in ms]

    def set_timing(self, mode: int, every: int = 1):
        """0: no HIP events; 1: all of them (the default); 2: the two around the scan kernel, on every
        `every`-th full evaluation only (tokscore_set_timing)."""
        self._check(self._lib.tokscore_set_timing(self._h, int(mode), int(every)))

    def drain_timings(self, cap: int = 256):
        """[(eval, scan, fixup, hist+reduce) ms] for every score call since the last drain."""
        ms = (C.c_float * (4 * cap))()
        n = C.c_uint32()
        self._check(self._lib.tokscore_drain_timings(self._h, ms, cap, C.byref(n)))
        return [[float(ms[4 * i + k]) for k in range(4)] for i in range(int(n.value))]

    def probe_read_bandwidth(self, nbytes: int, reps: int = 10) -> float:
        """GB/s a read-only streaming kernel reaches on this GPU over a buffer of nbytes."""
        v = C.c_double()
        self._check(self._lib.tokscore_probe_read_bandwidth(self._h, nbytes, reps, C.byref(v)))
        return float(v.value)

    def hosttrie_longest(self, text: bytes):
        a = np.frombuffer(text, dtype=np.uint8) if text else np.zeros(1, dtype=np.uint8)
        ln = C.c_uint64()
        tid = C.c_uint64()
        self._check(self._lib.tokscore_hosttrie_longest(self._h, _p8(a), len(text), C.byref(ln), C.byref(tid)))
        return int(ln.value), int(tid.value)

    def hostdfa_check(self, corpus: np.ndarray, doc_off: np.ndarray, max_slots: int = 0, lds_slots: int = 0):
        """Host self-check of the scan automaton (tokscore_hostdfa_check): -> (counts[V+3],
        total_tokens, stats[8]) from running the image over the documents on the CPU."""
        b = np.ascontiguousarray(corpus, dtype=np.uint8)
        if b.size == 0:
            b = np.zeros(1, dtype=np.uint8)
        o = np.ascontiguousarray(doc_off, dtype=np.uint64)
        counts = np.zeros(self.V + 3, dtype=np.uint64)
        tt = C.c_uint64()
        stats = np.zeros(8, dtype=np.uint64)
        self._check(self._lib.tokscore_hostdfa_check(self._h, _p8(b), _p64(o), len(o) - 1, max_slots, lds_slots, _p64(counts),
                                                     C.byref(tt), _p64(stats)))
        return counts, int(tt.value), stats
