"""ctypes access to oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference algorithm (oracle/*.c); it is
the checker for the HIP path, never the thing shipped or measured.  Nothing
under tokenizer_amd/ imports this module.
"""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_LIB = None

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True,
                           stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.ora_vocab_new.restype = C.c_void_p
        L.ora_vocab_new.argtypes = [u8p, u64p, C.c_uint64]
        L.ora_vocab_free.argtypes = [C.c_void_p]
        L.ora_vocab_max_len.restype = C.c_uint64
        L.ora_vocab_max_len.argtypes = [C.c_void_p]
        L.ora_get_id.restype = C.c_uint64
        L.ora_get_id.argtypes = [C.c_void_p, u8p, C.c_uint64]
        L.ora_tokenize.restype = C.c_uint64
        L.ora_tokenize.argtypes = [C.c_void_p, u8p, C.c_uint64, u64p]
        L.ora_score.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, u64p, u64p, u64p, C.c_int]
        L.ora_count_occurrences.argtypes = [C.c_void_p, u8p, u64p, C.c_uint64, C.c_uint32, u64p, u64p, C.c_int]
        _LIB = L
    return _LIB


def _p8(a):
    return a.ctypes.data_as(u8p)


def _p64(a):
    return a.ctypes.data_as(u64p)


class OracleVocab:
    """SolutionFile restated (ids, first-duplicate-wins, max_len_ in bytes)."""

    def __init__(self, tok_bytes: np.ndarray, tok_off: np.ndarray):
        self.V = len(tok_off) - 1
        self._b = np.ascontiguousarray(tok_bytes, dtype=np.uint8)
        if self._b.size == 0:
            self._b = np.zeros(1, dtype=np.uint8)
        self._o = np.ascontiguousarray(tok_off, dtype=np.uint64)
        self._h = lib().ora_vocab_new(_p8(self._b), _p64(self._o), self.V)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ora_vocab_free(self._h)
            self._h = None

    @property
    def max_len(self):
        return int(lib().ora_vocab_max_len(self._h))

    def get_id(self, s: bytes):
        a = np.frombuffer(s, dtype=np.uint8) if s else np.zeros(1, dtype=np.uint8)
        r = int(lib().ora_get_id(self._h, _p8(a), len(s)))
        return -1 if r == 2 ** 64 - 1 else r

    def tokenize(self, text: bytes):
        a = np.frombuffer(text, dtype=np.uint8) if text else np.zeros(1, dtype=np.uint8)
        out = np.zeros(len(text) + 2, dtype=np.uint64)
        k = int(lib().ora_tokenize(self._h, _p8(a), len(text), _p64(out)))
        return [-1 if int(x) == 2 ** 64 - 1 else int(x) for x in out[:k]]

    def score(self, corpus: np.ndarray, doc_off: np.ndarray, threads: int = 1):
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        if corpus.size == 0:
            corpus = np.zeros(1, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        counts = np.zeros(self.V + 3, dtype=np.uint64)
        tt = C.c_uint64(0)
        tb = C.c_uint64(0)
        lib().ora_score(self._h, _p8(corpus), _p64(doc_off), len(doc_off) - 1, _p64(counts), C.byref(tt),
                        C.byref(tb), threads)
        return counts, int(tt.value), int(tb.value)


    def count_occurrences(self, corpus: np.ndarray, doc_off: np.ndarray, max_cp: int, threads: int = 1):
        """-> (counts[V+3] by id, code-point positions); GetTokens.cpp:21-27 on a closed vocabulary."""
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        if corpus.size == 0:
            corpus = np.zeros(1, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        counts = np.zeros(self.V + 3, dtype=np.uint64)
        pos = C.c_uint64(0)
        lib().ora_count_occurrences(self._h, _p8(corpus), _p64(doc_off), len(doc_off) - 1, max_cp, _p64(counts),
                                    C.byref(pos), threads)
        return counts, int(pos.value)


def _read_json(path):
    # (bytes that are not UTF-8 pass through the reference's parser untouched: kept as surrogate escapes)
    if os.path.exists(path):
        with open(path, "rb") as f:
            return json.loads(f.read().decode("utf-8", "surrogateescape"))
    with gzip.open(path + ".gz", "rb") as f:
        return json.loads(f.read().decode("utf-8", "surrogateescape"))


def golden_cases():
    """The scoring cases (tokens + shard + what the compiled reference printed for them)."""
    return sorted(d for d in os.listdir(GOLDEN) if os.path.isdir(os.path.join(GOLDEN, d))
                  and (os.path.exists(os.path.join(GOLDEN, d, "tokens.json"))
                       or os.path.exists(os.path.join(GOLDEN, d, "tokens.json.gz"))))


def load_cands(name):
    """A candidate-list golden (ref_harness `cands`): dict(max_len, docs=list[bytes],
    names=list[bytes], uses=np.uint64[n]) in the reference's own candidate order."""
    j = _read_json(os.path.join(GOLDEN, name, "cands.json"))
    return {"max_len": int(j["max_len"]), "docs": [t.encode("utf-8") for t in j["docs"]],
            "names": [t[0].encode("utf-8") for t in j["tokens"]],
            "uses": np.array([t[1] for t in j["tokens"]], dtype=np.uint64)}


def load_golden(name):
    """-> dict(tokens=list[bytes], docs=list[bytes], expected=dict).

    docs are what DataFile::GetEntries() hands to Tokenize: the JSON "text"
    string as UTF-8 bytes, cut at the first NUL (DataFile.cpp:27-29 copies
    through a C string)."""
    d = os.path.join(GOLDEN, name)
    tj = _read_json(os.path.join(d, "tokens.json"))
    assert tj["version"] == "0.4.0"
    tokens = [t.encode("utf-8", "surrogateescape") for t in tj["tokens"]]
    docs = [e["text"].encode("utf-8", "surrogateescape").split(b"\0")[0] for e in _read_json(os.path.join(d, "shard.json"))]
    exp = _read_json(os.path.join(d, "expected.json"))
    V = len(tokens)
    counts = np.zeros(V + 3, dtype=np.uint64)
    for k, c in exp["counts"].items():
        counts[V + 2 if k == "-1" else int(k)] = c
    exp["counts_vec"] = counts
    return {"tokens": tokens, "docs": docs, "expected": exp}
