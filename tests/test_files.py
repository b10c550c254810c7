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


def test_metadata_file_scan_and_format(tmp_path):
    # the bytes below are what the compiled reference wrote for this directory (ref_harness `cands`
    # on a.json, sub/b.json, an invalid bad.json and a note.txt): MetadataFile.cpp:33-69
    from tokenizer_amd.files import MetadataFile
    data = tmp_path / "data"
    (data / "sub").mkdir(parents=True)
    shard = json.dumps([{"id": "1", "title": "t", "text": "hello"}])
    (data / "a.json").write_text(shard)
    (data / "sub" / "b.json").write_text(shard)
    (data / "bad.json").write_text("{}")
    (data / "note.txt").write_text("x")
    m = MetadataFile(str(data / ".metadata.json"))
    assert not m.IsValid()  # built, not read (MetadataFile.cpp:73-74)
    assert [e.path for e in m.GetFiles()] == ["a.json", "sub/b.json"]
    assert [e.path for e in m.GetFiles(1)] == ["a.json"]
    assert m.GetRootPath() == os.path.realpath(str(data))
    want = ('{\n    "version": "0.4.0",\n    "files": [{\n            "path": "a.json"\n        }, {\n'
            '            "path": "sub/b.json"\n        }]\n}')
    assert (data / ".metadata.json").read_text() == want
    again = MetadataFile(str(data / ".metadata.json"))
    assert again.IsValid() and [e.path for e in again.GetFiles()] == ["a.json", "sub/b.json"]
    (data / "c.json").write_text(shard)
    assert len(MetadataFile(str(data / ".metadata.json")).GetFiles()) == 2       # the index is trusted as it is
    assert len(MetadataFile(str(data / ".metadata.json"), True).GetFiles()) == 3  # rebuild
    empty = tmp_path / "empty"
    empty.mkdir()
    MetadataFile(str(empty / ".metadata.json"))
    assert (empty / ".metadata.json").read_text() == '{\n    "version": "0.4.0",\n    "files": []\n}'
    (empty / ".metadata.json").write_text('{"version": "0.3.0", "files": []}')
    assert not MetadataFile(str(empty / ".metadata.json")).IsValid()  # wrong version: rebuilt
