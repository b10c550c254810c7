"""Host mirror of the reference's file classes (tokenizer_amd/files.py) against the
reference's own output: the `.tokens.json` the reference annealer wrote for the
wiki_annealed golden (SolutionFile.cpp:27-41 + JsonFile.cpp:35-46) and the shards the goldens
were scored on.  No GPU needed: nothing here segments text."""
import gzip
import json
import os

import numpy as np
import pytest

from tokenizer_amd import files

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _gunzip_to(tmp_path, case, name):
    dst = os.path.join(str(tmp_path), name)
    with gzip.open(os.path.join(GOLD, case, name + ".gz"), "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    return dst


def test_writer_reproduces_reference_written_file(tmp_path):
    ref = _gunzip_to(tmp_path, "wiki_annealed", "tokens.json")
    sol = files.SolutionFile(ref)
    assert sol.IsValid()
    out = os.path.join(str(tmp_path), "rewritten.tokens.json")
    again = files.SolutionFile([t.decode("utf-8") for t in sol.tokens], out)
    assert open(out, "rb").read() == open(ref, "rb").read()
    assert again.tokens == sol.tokens and again.max_len_ == sol.max_len_


def test_writer_escapes_like_rapidjson(tmp_path):
    out = os.path.join(str(tmp_path), "t.tokens.json")
    files.SolutionFile(['a"b', "c\\d", "\n", "\t", "\x01", "é/", ""], out)
    text = open(out, "rb").read().decode("utf-8")
    assert text == '{\n    "version": "0.4.0",\n    "tokens": ["a\\"b", "c\\\\d", "\\n", "\\t", "\\u0001", "é/", ""]\n}'
    back = files.SolutionFile(out)
    assert back.IsValid() and [t.decode() for t in back.tokens] == ['a"b', "c\\d", "\n", "\t", "\x01", "é/", ""]


def test_ids_first_duplicate_wins_and_max_len_in_bytes():
    sol = files.SolutionFile(os.path.join(GOLD, "dup_tokens", "tokens.json"))
    assert sol.IsValid()
    toks = json.load(open(os.path.join(GOLD, "dup_tokens", "tokens.json"), encoding="utf-8"))["tokens"]
    first = {}
    for i, t in enumerate(toks):
        first.setdefault(t, i + 2)
    assert len(first) < len(toks), "fixture is meant to hold duplicates"
    for t, i in first.items():
        assert sol.GetId(t) == i
    assert sol.GetId("definitely-not-a-token") == files.UNKNOWN
    assert sol.max_len_ == max(len(t.encode("utf-8")) for t in toks)
    assert sol.GetToken(0) == "<START>" and sol.GetToken(1) == "<END>" and sol.GetToken(files.UNKNOWN) == "<UNKNOWN>"
    assert sol.GetToken(2) == toks[0]


@pytest.mark.parametrize("doc", [
    "not json at all",
    json.dumps([1, 2, 3]),
    json.dumps({"tokens": ["a"]}),
    json.dumps({"version": "0.3.9", "tokens": ["a"]}),
    json.dumps({"version": "0.4.0"}),
    json.dumps({"version": "0.4.0", "tokens": "a"}),
    json.dumps({"version": "0.4.0", "tokens": ["a", 1]}),
])
def test_invalid_tokens_files_are_flagged_not_thrown(tmp_path, doc):
    p = os.path.join(str(tmp_path), "bad.tokens.json")
    open(p, "w").write(doc)
    assert not files.SolutionFile(p).IsValid()
    assert not files.SolutionFile(os.path.join(str(tmp_path), "missing.json")).IsValid()


@pytest.mark.parametrize("doc", [
    "{", json.dumps({}), json.dumps([]), json.dumps([1]),
    json.dumps([{"id": "0", "text": "x"}]),
    json.dumps([{"id": 0, "text": "x", "title": ""}]),
    json.dumps([{"id": "0", "text": None, "title": ""}]),
])
def test_invalid_shards_are_flagged_not_thrown(tmp_path, doc):
    p = os.path.join(str(tmp_path), "bad.json")
    open(p, "w").write(doc)
    assert not files.DataFile(p).IsValid()


def test_shard_entries_are_nul_truncated_and_packed():
    case = os.path.join(GOLD, "nul_doc")
    shard = files.DataFile(os.path.join(case, "shard.json"))
    assert shard.IsValid()
    exp = json.load(open(os.path.join(case, "expected.json")))
    raw = json.load(open(os.path.join(case, "shard.json"), encoding="utf-8"))
    assert any("\x00" in e["text"] for e in raw), "fixture is meant to hold a NUL"
    corpus, off = shard.packed()
    assert len(off) == exp["docs"] + 1 and int(off[-1]) == exp["total_bytes"] == corpus.size
    for e, r in zip(shard.GetEntries(), raw):
        assert e.text == r["text"].split("\x00")[0].encode("utf-8")
    assert off.dtype == np.uint64 and corpus.dtype == np.uint8
