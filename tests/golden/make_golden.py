#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the COMPILED REFERENCE.

Run in the build container only (needs oracle/_ref/ref_harness, which links the
unmodified sources under /root/reference -- see oracle/Makefile):

    make -C oracle && python tests/golden/make_golden.py

Every case directory holds data only:
    tokens.json        a .tokens.json ({"version","tokens"}; SolutionFile.cpp:14-41)
    shard.json[.gz]    a corpus shard (array of {id,title,text}; DataFile.cpp:3-17)
    expected.json[.gz] what ref_harness `score` printed for that pair: counts per id
                       ("-1" = UNKNOWN), total_tokens, total_bytes, tokens per document
                       and, for the small cases, the id sequence Tokenize returned.
dag_*.json are dumps of ref_harness `dag` (Token.cpp:17-105 primitives).

The reference has no tests or fixtures of its own; these
files are the pin for oracle/ and, through it, for the HIP path.
"""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tokenizer_amd import synth  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
VERSION = "0.4.0"  # /root/reference/src/config.h:3


def write_tokens(path, tokens):
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"version": VERSION, "tokens": tokens}, f, ensure_ascii=False)


def write_shard(path, texts):
    entries = [{"id": str(i), "text": t, "title": "doc %d" % i} for i, t in enumerate(texts)]
    with open(path, "w", encoding="utf-8") as f:
        json.dump(entries, f, ensure_ascii=False)


def run_case(name, tokens, texts, ids=True, gz=False, tokens_path=None):
    d = os.path.join(HERE, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    tp = os.path.join(d, "tokens.json")
    if tokens_path:
        shutil.copy(tokens_path, tp)
    else:
        write_tokens(tp, tokens)
    sp = os.path.join(d, "shard.json")
    write_shard(sp, texts)
    ep = os.path.join(d, "expected.json")
    cmd = [HARNESS, "score", tp, sp, ep] + (["--ids"] if ids else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    exp = json.load(open(ep))
    exp.pop("tokenize_seconds", None)
    exp.pop("threads", None)
    with open(ep, "w") as f:
        json.dump(exp, f, separators=(",", ":"))
    if gz:
        for p in (sp, ep):
            with open(p, "rb") as fi, gzip.GzipFile(p + ".gz", "wb", mtime=0) as fo:
                fo.write(fi.read())
            os.remove(p)
    print("%-18s docs=%d bytes=%d tokens=%d" % (name, exp["docs"], exp["total_bytes"], exp["total_tokens"]))


def docs_of(corpus, off):
    b = corpus.tobytes()
    return [b[int(off[i]):int(off[i + 1])].decode("utf-8") for i in range(len(off) - 1)]


def gen_cands():
    # (13) f1: the candidate list (names + uses = occurrence counts) GetTokens builds from a
    # small mixed shard: ASCII words, CJK, 2- and 4-byte characters, a short and an empty-ish doc.
    texts = docs_of(*synth.wiki_like(2_500, seed=11, n_words=300, median_doc=400))
    texts += docs_of(*synth.cjk_heavy(1_200, seed=12, doc_lo=64, doc_hi=600))
    texts += ["Ünïcödé 𝄞𝄞 clef", "a", "aaaaaaaaaaaaaaaaaaaaaaaa"]
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "data")
        os.makedirs(data)
        write_shard(os.path.join(data, "a.json"), texts)
        out = os.path.join(tmp, "cands.json")
        subprocess.run([HARNESS, "cands", data, "6", out], check=True, cwd=tmp, stdout=subprocess.DEVNULL)
        j = json.load(open(out))
    d = os.path.join(HERE, "cands_mixed")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    with gzip.GzipFile(os.path.join(d, "cands.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps({"max_len": j["max_len"], "docs": texts, "tokens": j["tokens"]},
                           separators=(",", ":"), ensure_ascii=False).encode("utf-8"))
    print("cands_mixed candidates=%d docs=%d" % (len(j["tokens"]), len(texts)))


def gen_invalid_utf8():
    # rapidjson's default flags do not validate UTF-8 (JsonFile.cpp:21-27): bytes that are no UTF-8
    # pass through tokens and article texts as they are.  Written byte by byte (json.dump could not).
    d = os.path.join(HERE, "invalid_utf8")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    tp, sp, ep = (os.path.join(d, n) for n in ("tokens.json", "shard.json", "expected.json"))
    with open(tp, "wb") as f:
        f.write(b'{"version": "0.4.0", "tokens": ["a", "b", "c", "f", "caf", "\xe9", "f\xe9", " ", "\xc3\xa9", "\xff\xfe"]}')
    with open(sp, "wb") as f:
        f.write(b'[{"id": "0", "text": "caf\xe9 caf\xc3\xa9", "title": "latin-1 and utf-8"},'
                b' {"id": "1", "text": "ab\xff\xfe\xffc \x80", "title": "t\xe9"},'
                b' {"id": "2", "text": "\xc3", "title": "cut sequence"}]')
    subprocess.run([HARNESS, "score", tp, sp, ep, "--ids"], check=True, stdout=subprocess.DEVNULL)
    exp = json.load(open(ep))
    exp.pop("tokenize_seconds", None)
    exp.pop("threads", None)
    with open(ep, "w") as f:
        json.dump(exp, f, separators=(",", ":"))
    print("%-18s docs=%d bytes=%d tokens=%d" % ("invalid_utf8", exp["docs"], exp["total_bytes"], exp["total_tokens"]))


TEXT_CASES = ["invalid_utf8", "ascii_caps", "cjk_unknown", "dup_tokens", "nul_doc", "short_docs", "upper_token", "prefix_greedy"]


def gen_text():
    # SolutionFile::Detokenize / Prettify (SolutionFile.cpp:102-118) of every document's ids, for
    # the small cases as they stand (ref_harness `text`); hex because a token may hold any byte.
    for name in TEXT_CASES:
        d = os.path.join(HERE, name)
        out = os.path.join(d, "text.json")
        subprocess.run([HARNESS, "text", os.path.join(d, "tokens.json"), os.path.join(d, "shard.json"), out], check=True)
        j = json.load(open(out))
        with open(out, "w") as f:
            json.dump(j, f, separators=(",", ":"))
        print("%-18s text of %d documents" % (name, len(j["detokenize_hex"])))


def gen_anneal():
    # (14) f2 / f4: GetTokens' own `.candidates-all-5.bin` (WriteTokens) for a small mixed shard, and
    # a single-threaded, seeded run of TokenGenerator::WorkerTask on it: the state after every
    # task and GetSolution at the end (ref_harness `trace`).
    texts = docs_of(*synth.wiki_like(4_000, seed=21, n_words=200, median_doc=500))
    texts += docs_of(*synth.cjk_heavy(1_500, seed=22, doc_lo=64, doc_hi=500))
    texts += ["Ünïcödé 𝄞𝄞 clef 𝄞", "b", "abababababababab"]
    pref, batch, seed, ntasks, max_len = 150, 16, 12345, 700, 5
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "data")
        os.makedirs(data)
        write_shard(os.path.join(data, "a.json"), texts)
        out = os.path.join(tmp, "trace.json")
        subprocess.run([HARNESS, "trace", data, str(max_len), str(pref), str(batch), str(seed), str(ntasks), out],
                       check=True, cwd=tmp, stdout=subprocess.DEVNULL)
        j = json.load(open(out))
        blob = open(os.path.join(data, ".candidates-all-%d.bin" % max_len), "rb").read()
    d = os.path.join(HERE, "anneal_mixed")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    j["docs"] = texts
    with gzip.GzipFile(os.path.join(d, "trace.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(j, separators=(",", ":"), ensure_ascii=False).encode("utf-8"))
    with gzip.GzipFile(os.path.join(d, "candidates.bin.gz"), "wb", mtime=0) as f:
        f.write(blob)
    print("anneal_mixed tokens=%d tot_cand=%d tasks=%d solution=%d bin=%d bytes"
          % (j["n_tokens"], j["tot_cand"], len(j["states"]), len(j["solution"]), len(blob)))


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/ref_harness first (make -C oracle)")

    # (1) R1: ASCII with capitals -- input is lower-cased, vocabulary is not.
    run_case("ascii_caps",
             ["the", " the", "he", "th", " ", "t", "h", "e", "q", "u", "i", "c", "k", "b", "r", "o", "w", "n",
              "f", "x", ".", "quick", " brown", "fox"],
             ["The Quick BROWN fox. THE the tHe", "Zebra THE quick", "the"])

    # (2) R2/R4: multibyte text, some characters absent from the vocabulary
    #     (each missing byte costs one UNKNOWN).
    run_case("cjk_unknown",
             ["日本", "日", "本", "語", "日本語", "の", " ", "a", "é", "ét", "t", "中"],
             ["日本語の本 日本 été 中国語", "国", "a日b本c", "été"])

    # (3) R3: a token that is a prefix of another; greedy longest-first != optimal.
    run_case("prefix_greedy",
             ["a", "b", "c", "ab", "abc", "bcd", "d", "abcd", "cd", "abcde_never", "bc"],
             ["abcd", "abcbcd", "ababcabcdabc", "dcba", "abce", "bcdabc"])

    # (4) R6: duplicate strings in a hand-written tokens file keep the FIRST id.
    run_case("dup_tokens",
             ["a", "b", "ab", "a", "ab", "c", "b", "abc", "abc"],
             ["abcabc", "aabbcc", "cab"])

    # (5) R3/R8: empty article, article shorter than max_len_, article ending mid-window.
    run_case("short_docs",
             ["hello world", "hello", "hell", "he", "h", "e", "l", "o", " ", "w", "r", "d", "wor", "world!"],
             ["", "h", "he", "hello worl", "hello world", "hello world!", "", "worl", "xyz", ""])

    # (6) R9: GetEntries() copies text through a C string => truncated at the first NUL.
    run_case("nul_doc",
             ["a", "b", "c", "abc", "\u0001"],
             ["abc\u0000abcabc", "\u0000abc", "ab\u0001c", "abcabc"])

    # (7) R1: a token containing A-Z can never match; its lower-case twin does.
    run_case("upper_token",
             ["The", "the", "T", "t", "h", "e", "HE", "he", " ", "A", "a"],
             ["The the THE HE he A a", "tHE"])

    # (8) non-merging / periodic chains: adversarial for any speculative parallel
    #     segmentation (chains started one byte apart never meet inside a run),
    #     one document far longer than any GPU work chunk.
    run_case("periodic",
             ["ab", "ba", "a", "b", "aa", "aaa", "abab", "x", "=", "==", "xyz", "-", "--", "---", "\n"],
             ["ab" * 150000 + "x" + "ba" * 1000,
              "a" * 70001,
              "=" * 4099 + "\nxyz\n" + "-" * 777 + "ab" * 37 + "b",
              "b" + "ab" * 50000,
              "aab" * 30000],
             ids=False, gz=True)

    # (9) ~300 KB wiki-like shard scored with the vocabulary produced by the
    #     REFERENCE's own pipeline (GetTokens -> TokenGenerator -> GetSolution).
    corpus, off = synth.wiki_like(300_000, seed=1234, n_words=20_000, median_doc=1500)
    texts = docs_of(corpus, off)
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "data")
        os.makedirs(data)
        half = len(texts) // 2
        write_shard(os.path.join(data, "a.json"), texts[:half])
        write_shard(os.path.join(data, "b.json"), texts[half:])
        tok = os.path.join(tmp, "tokens.json")
        subprocess.run([HARNESS, "anneal", data, "10", "30000", "30", "12", tok], check=True, cwd=tmp,
                       stdout=subprocess.DEVNULL)
        run_case("wiki_annealed", None, texts, ids=False, gz=True, tokens_path=tok)
    d = os.path.join(HERE, "wiki_annealed")
    with open(os.path.join(d, "tokens.json"), "rb") as fi, \
            gzip.GzipFile(os.path.join(d, "tokens.json.gz"), "wb", mtime=0) as fo:
        fo.write(fi.read())
    os.remove(os.path.join(d, "tokens.json"))

    # (10) random valid UTF-8 over a 4k alphabet, vocabulary sampled from the corpus.
    corpus, off = synth.random_utf8(96_000, seed=355, doc_lo=256, doc_hi=40_000)
    vocab = synth.build_vocab(corpus, 6000, max_cp=10)
    run_case("random_utf8", [t.decode("utf-8") for t in vocab], docs_of(corpus, off), ids=False, gz=True)

    # (11) CJK-heavy multibyte text.
    corpus, off = synth.cjk_heavy(96_000, seed=5, doc_lo=64, doc_hi=20_000)
    vocab = synth.build_vocab(corpus[:60_000], 5000, max_cp=10)
    run_case("cjk_heavy", [t.decode("utf-8") for t in vocab], docs_of(corpus, off), ids=False, gz=True)

    # (12) a5 DAG primitives on a tiny corpus.
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "data")
        os.makedirs(data)
        write_shard(os.path.join(data, "a.json"),
                    ["abracadabra abracadabra", "the cat sat on the mat", "日本語の本 日本"])
        out = os.path.join(HERE, "dag_tiny.json")
        subprocess.run([HARNESS, "dag", data, "5", "400", "7", out], check=True, cwd=tmp,
                       stdout=subprocess.DEVNULL)
        j = json.load(open(out))
        with open(out, "w") as f:
            json.dump(j, f, separators=(",", ":"), ensure_ascii=False)
        print("dag_tiny tokens=%d steps=%d" % (len(j["tokens"]), len(j["steps"])))

    gen_cands()
    gen_anneal()
    gen_invalid_utf8()
    gen_text()


if __name__ == "__main__":
    if sys.argv[1:] == ["cands"]:  # only the f1 case (the annealed cases are not reproducible bit for bit)
        gen_cands()
    elif sys.argv[1:] == ["anneal"]:
        gen_anneal()
    elif sys.argv[1:] == ["text"]:
        gen_text()
    elif sys.argv[1:] == ["invalid_utf8"]:
        gen_invalid_utf8()
    else:
        main()
