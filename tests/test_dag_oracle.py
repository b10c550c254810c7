"""Row a5: the annealing move primitives restated in oracle/dag_oracle.c
against the dump the COMPILED REFERENCE produced (tests/golden/dag_tiny.json, written by
`ref_harness dag`: GetTokens on a tiny corpus, then a scripted single-threaded toggle sequence
through Token::SimulateStep / Token::ApplyStep<bool>, Token.cpp:83-105)."""
import ctypes as C
import json
import math
import os

import numpy as np

import oracle_ffi as O


class Dag(C.Structure):
    _fields_ = [("n", C.c_int64), ("lpar", O.i64p), ("rpar", O.i64p), ("luses", O.u64p), ("ruses", O.u64p),
                ("enabled", O.u8p)]


class Lomax(C.Structure):
    _fields_ = [("alpha", C.c_double), ("moment1", C.c_double), ("moment2", C.c_double), ("beta", C.c_double),
                ("sigma", C.c_double)]


def _lib():
    L = O.lib()
    L.ora_dag_simulate.restype = C.c_double
    L.ora_dag_simulate.argtypes = [C.POINTER(Dag), C.c_int64]
    L.ora_dag_apply.restype = C.c_double
    L.ora_dag_apply.argtypes = [C.POINTER(Dag), C.c_int64, C.c_int]
    L.ora_lomax_get_best.restype = C.c_double
    L.ora_lomax_get_best.argtypes = [C.POINTER(Lomax), C.c_double]
    L.ora_calc_score.restype = C.c_double
    L.ora_calc_score.argtypes = [C.POINTER(Lomax), C.c_double, C.c_uint64, C.c_uint64, C.c_uint64]
    for f in ("ora_lomax_init", "ora_lomax_update_params"):
        getattr(L, f).argtypes = [C.POINTER(Lomax)]
    L.ora_lomax_set_half_life.argtypes = [C.POINTER(Lomax), C.c_double]
    L.ora_lomax_set_moments.argtypes = [C.POINTER(Lomax), C.c_double, C.c_double]
    L.ora_lomax_add_point.argtypes = [C.POINTER(Lomax), C.c_double, C.c_double]
    return L


def test_dag_primitives_match_reference_dump():
    L = _lib()
    g = json.load(open(os.path.join(O.GOLDEN, "dag_tiny.json"), encoding="utf-8"))
    toks = g["tokens"]
    n = len(toks)
    lpar = np.array([t["l"] for t in toks], dtype=np.int64)
    rpar = np.array([t["r"] for t in toks], dtype=np.int64)
    luses = np.array([t["uses"] for t in toks], dtype=np.uint64)
    ruses = luses.copy()  # Token(name, uses) seeds both branches with the same count (Token.cpp:45-48)
    enabled = np.array([t["enabled"] for t in toks], dtype=np.uint8)
    # structure of the candidate DAG (Trie.cpp:92-103): l = string minus first char, r = minus last char
    names = [t["name"] for t in toks]
    for i, t in enumerate(toks):
        assert t["size"] == len(t["name"])
        if t["size"] == 1:
            assert t["l"] == -1 and t["r"] == -1 and t["enabled"] == 1
        else:
            assert names[t["l"]] == t["name"][1:] and names[t["r"]] == t["name"][:-1]
    dag = Dag(n, lpar.ctypes.data_as(O.i64p), rpar.ctypes.data_as(O.i64p), luses.ctypes.data_as(O.u64p),
              ruses.ctypes.data_as(O.u64p), enabled.ctypes.data_as(O.u8p))
    assert len(g["steps"]) > 100
    for s in g["steps"]:
        sim = L.ora_dag_simulate(C.byref(dag), s["i"])
        app = L.ora_dag_apply(C.byref(dag), s["i"], s["enable"])
        assert sim == s["sim"], s
        assert app == s["apply"], s
    fin = np.array(g["final"], dtype=np.uint64)
    assert np.array_equal(luses, fin[:, 0])
    assert np.array_equal(ruses, fin[:, 1])
    assert np.array_equal(enabled, fin[:, 2].astype(np.uint8))


def test_lomax_and_calc_score_follow_the_reference_formulas():
    # LomaxDist.cpp:5-51 / TokenGenerator.cpp:64-69 restated; checked against the closed forms.
    L = _lib()
    d = Lomax()
    L.ora_lomax_init(C.byref(d))
    L.ora_lomax_set_moments(C.byref(d), 10.0, 500.0)
    L.ora_lomax_set_half_life(C.byref(d), 50.0)
    assert d.alpha == math.log(2) / 50.0
    L.ora_lomax_add_point(C.byref(d), 30.0, 2.0)
    m1 = 10.0 + (30.0 - 10.0) * 2.0 * d.alpha
    m2 = 500.0 + (900.0 - 500.0) * 2.0 * d.alpha
    assert d.moment1 == m1 and d.moment2 == m2
    L.ora_lomax_update_params(C.byref(d))
    temp = m2 / (m2 - 2 * m1 * m1)
    assert d.sigma == m1 * temp and d.beta == temp + 1
    p = 0.25
    t = 1.0 - 1 / d.beta
    best = d.sigma * (p ** t / t - p)
    assert L.ora_lomax_get_best(C.byref(d), p) == best
    assert L.ora_calc_score(C.byref(d), 123.0, 0, 1000, 100) == 0
    tot, pref, cnt, raw = 1000, 100, 250, 4321.0
    contrib = tot * best
    fill = cnt / pref
    assert L.ora_calc_score(C.byref(d), raw, cnt, tot, pref) == raw / contrib * fill * (2 - fill)
