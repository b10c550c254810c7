"""Seeded synthetic corpora and vocabularies (SURVEY.md section 8d stand-ins).

Real Wikipedia-2020-11 shards are not available offline, so the benchmark and
the parity tests use deterministic generators with the same shape:

* wiki_like   -- Zipfian pseudo-English articles (BASELINE configs 1-3)
* random_utf8 -- uniform code points over a 4k-symbol 1/2/3/4-byte alphabet
                 (config 4, deep-trie / gather stress)
* cjk_heavy   -- CJK-dominated multibyte text (config 5)
* build_vocab -- all single characters + top substrings by count*(len-1)

A corpus is (bytes: np.uint8[B], doc_offsets: np.uint64[D+1]); document d is
bytes[doc_offsets[d]:doc_offsets[d+1]] -- the packed form of
DataFile::GetEntries() (/root/reference/src/files/DataFile.cpp:24-31).
"""
from __future__ import annotations

import numpy as np

_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_P = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4,
                      2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
_LETTER_P = _LETTER_P / _LETTER_P.sum()
_ACCENTS = "éèüöäñçåøß"


def _lexicon(rng: np.random.Generator, n_words: int):
    """Lexicon as (flat bytes, starts, lengths); ~3% of words carry a 2-byte Latin letter."""
    lens = np.clip(rng.geometric(0.22, n_words) + 1, 1, 12)
    lens[:64] = np.clip(lens[:64], 1, 4)  # the hottest words are short, as in English
    total = int(lens.sum())
    flat = _LETTERS[rng.choice(len(_LETTERS), size=total, p=_LETTER_P)].copy()
    starts = np.zeros(n_words, dtype=np.int64)
    np.cumsum(lens[:-1], out=starts[1:])
    words = [flat[s:s + l].tobytes() for s, l in zip(starts.tolist(), lens.tolist())]
    acc = rng.random(n_words) < 0.03
    acc[:200] = False
    for i in np.nonzero(acc)[0].tolist():
        w = words[i].decode("ascii")
        k = int(rng.integers(0, len(w)))
        a = _ACCENTS[int(rng.integers(0, len(_ACCENTS)))]
        words[i] = (w[:k] + a + w[k + 1:]).encode("utf-8")
    lens = np.array([len(w) for w in words], dtype=np.int64)
    flat = np.frombuffer(b"".join(words), dtype=np.uint8)
    starts = np.zeros(n_words, dtype=np.int64)
    np.cumsum(lens[:-1], out=starts[1:])
    return flat, starts, lens


_SEPS = [b" ", b", ", b". ", b"\n", b" (", b") ", b"; ", b": ", b" - ", b"\n\n", b"'s ", b" \""]
_SEP_P = np.array([78, 6, 6, 2.5, 1, 1, 0.8, 0.8, 0.7, 1.2, 1.2, 0.8])
_SEP_P = _SEP_P / _SEP_P.sum()


def wiki_like(total_bytes: int, seed: int = 20201120, n_words: int = 200_000,
              median_doc: int = 2048, sigma: float = 1.2):
    """Zipf(s=1) words from an a-z lexicon, 5% capitalised, punctuation/newlines,
    ~0.5% non-ASCII; article lengths log-normal (median 2 KB, clip 64 B..2 MB)."""
    rng = np.random.default_rng(seed)
    lex, lstart, llen = _lexicon(rng, n_words)
    p = 1.0 / np.arange(1, n_words + 1)
    cdf = np.cumsum(p / p.sum())
    sep_flat = np.frombuffer(b"".join(_SEPS), dtype=np.uint8)
    sep_len = np.array([len(s) for s in _SEPS], dtype=np.int64)
    sep_start = np.zeros(len(_SEPS), dtype=np.int64)
    np.cumsum(sep_len[:-1], out=sep_start[1:])
    sep_cdf = np.cumsum(_SEP_P)

    pieces = []
    word_starts = []  # absolute byte offset of every word occurrence (document cut points)
    produced = 0
    block = 4_000_000
    while produced < total_bytes:
        n = int(min(block, (total_bytes - produced) / 5.5 + 1024))
        w = np.minimum(np.searchsorted(cdf, rng.random(n)), n_words - 1)
        s = np.minimum(np.searchsorted(sep_cdf, rng.random(n)), len(_SEPS) - 1)
        wl = llen[w]
        sl = sep_len[s]
        tl = wl + sl
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(tl, out=off[1:])
        nb = int(off[-1])
        occ = np.repeat(np.arange(n), tl)
        within = np.arange(nb) - off[occ]
        in_word = within < wl[occ]
        out = np.empty(nb, dtype=np.uint8)
        src_w = lstart[w][occ] + within
        src_s = sep_start[s][occ] + (within - wl[occ])
        out[in_word] = lex[src_w[in_word]]
        out[~in_word] = sep_flat[src_s[~in_word]]
        cap = rng.random(n) < 0.05
        first = off[:-1][cap]
        fb = out[first]
        ok = (fb >= 97) & (fb <= 122)
        out[first[ok]] = fb[ok] - 32
        pieces.append(out)
        word_starts.append(off[:-1] + produced)
        produced += nb
    data = np.concatenate(pieces)
    ws = np.concatenate(word_starts)
    # cut at the last word start <= total_bytes so no UTF-8 sequence is split
    end = int(ws[np.searchsorted(ws, total_bytes, side="right") - 1]) if produced > total_bytes else produced
    if end == 0:
        end = produced
    data = data[:end]
    ws = ws[ws < end]
    n_docs_guess = max(4, int(end / median_doc * 2) + 16)
    dl = np.clip(np.exp(rng.normal(np.log(median_doc), sigma, n_docs_guess)), 64, 2 << 20)
    cuts = np.cumsum(dl)
    cuts = cuts[cuts < end]
    idx = np.unique(ws[np.minimum(np.searchsorted(ws, cuts), len(ws) - 1)])
    idx = idx[(idx > 0) & (idx < end)]
    offsets = np.concatenate(([0], idx, [end])).astype(np.uint64)
    return np.ascontiguousarray(data), offsets


def _encode_codepoints(cps: np.ndarray):
    """UTF-8 encode an int array of code points -> (bytes u8[], start offset per code point)."""
    cps = cps.astype(np.int64)
    nb = np.where(cps < 0x80, 1, np.where(cps < 0x800, 2, np.where(cps < 0x10000, 3, 4)))
    off = np.zeros(len(cps) + 1, dtype=np.int64)
    np.cumsum(nb, out=off[1:])
    out = np.zeros(int(off[-1]), dtype=np.uint8)
    s = off[:-1]
    m1 = nb == 1
    out[s[m1]] = cps[m1]
    m2 = nb == 2
    out[s[m2]] = 0xC0 | (cps[m2] >> 6)
    out[s[m2] + 1] = 0x80 | (cps[m2] & 0x3F)
    m3 = nb == 3
    out[s[m3]] = 0xE0 | (cps[m3] >> 12)
    out[s[m3] + 1] = 0x80 | ((cps[m3] >> 6) & 0x3F)
    out[s[m3] + 2] = 0x80 | (cps[m3] & 0x3F)
    m4 = nb == 4
    out[s[m4]] = 0xF0 | (cps[m4] >> 18)
    out[s[m4] + 1] = 0x80 | ((cps[m4] >> 12) & 0x3F)
    out[s[m4] + 2] = 0x80 | ((cps[m4] >> 6) & 0x3F)
    out[s[m4] + 3] = 0x80 | (cps[m4] & 0x3F)
    return out, off


def _cut_docs(rng, cp_off: np.ndarray, lo: int, hi: int):
    """Document offsets (bytes) on code-point boundaries, lengths log-uniform in [lo, hi]."""
    end = int(cp_off[-1])
    n_guess = max(4, int(end / lo) + 4)
    dl = np.exp(rng.uniform(np.log(lo), np.log(hi), n_guess))
    cuts = np.cumsum(dl)
    cuts = cuts[cuts < end]
    idx = np.unique(cp_off[np.minimum(np.searchsorted(cp_off, cuts), len(cp_off) - 1)])
    idx = idx[(idx > 0) & (idx < end)]
    return np.concatenate(([0], idx, [end])).astype(np.uint64)


def random_utf8(total_bytes: int, seed: int = 355, doc_lo: int = 4096, doc_hi: int = 4 << 20):
    """Valid UTF-8, code points uniform over a 4k alphabet mixing 1/2/3/4-byte encodings."""
    rng = np.random.default_rng(seed)
    a1 = np.arange(0x20, 0x7F)
    a1 = np.resize(a1, 1024)  # 95 printable ASCII symbols repeated: 25% of draws are 1-byte
    a2 = rng.choice(np.arange(0x80, 0x800), 1024, replace=False)
    a3 = rng.choice(np.concatenate((np.arange(0x800, 0xD800), np.arange(0xE000, 0x10000))), 1024, replace=False)
    a4 = rng.choice(np.arange(0x10000, 0x30000), 1024, replace=False)
    alpha = np.concatenate((a1, a2, a3, a4))
    n = int(total_bytes / 2.5)
    cps = alpha[rng.integers(0, len(alpha), n)]
    data, off = _encode_codepoints(cps)
    k = int(np.searchsorted(off, total_bytes, side="right") - 1)
    data = data[:off[k]]
    off = off[:k + 1]
    return np.ascontiguousarray(data), _cut_docs(rng, off, doc_lo, doc_hi)


def cjk_heavy(total_bytes: int, seed: int = 5, doc_lo: int = 512, doc_hi: int = 65536):
    """70% Zipf over 6k CJK ideographs, 20% kana, 10% ASCII."""
    rng = np.random.default_rng(seed)
    n = int(total_bytes / 2.75)
    han = 0x4E00 + rng.permutation(0x9FFF - 0x4E00)[:6000]
    p = 1.0 / np.arange(1, 6001)
    cdf = np.cumsum(p / p.sum())
    kind = rng.random(n)
    cps = np.empty(n, dtype=np.int64)
    m = kind < 0.7
    cps[m] = han[np.minimum(np.searchsorted(cdf, rng.random(int(m.sum()))), 5999)]
    m2 = (kind >= 0.7) & (kind < 0.9)
    cps[m2] = rng.integers(0x3041, 0x30FF, int(m2.sum()))
    m3 = kind >= 0.9
    asc = np.frombuffer(b" etaoinshrdlu.,\n0123456789ETAOIN", dtype=np.uint8)
    cps[m3] = asc[rng.integers(0, len(asc), int(m3.sum()))]
    data, off = _encode_codepoints(cps)
    k = int(np.searchsorted(off, total_bytes, side="right") - 1)
    data = data[:off[k]]
    off = off[:k + 1]
    return np.ascontiguousarray(data), _cut_docs(rng, off, doc_lo, doc_hi)


def build_vocab(sample: np.ndarray, vocab_size: int, max_cp: int = 10, lowercase: bool = True):
    """All single characters of `sample` + top substrings (2..max_cp code points) by
    count*(len-1), most valuable first then the single characters -- the order
    TokenGenerator::GetSolution emits (/root/reference/src/tokenizer/TokenGenerator.cpp:190-217).
    The sample is ASCII-lower-cased first because SolutionFile::Tokenize lower-cases its
    input but never the vocabulary (SolutionFile.cpp:87-88), so upper-case tokens are dead."""
    text = sample.tobytes().decode("utf-8", errors="ignore")
    if lowercase:
        text = text.translate({i: i + 32 for i in range(65, 91)})
    cps = np.frombuffer(text.encode("utf-32-le"), dtype=np.uint32).astype(np.uint64)
    n = len(cps)
    roots = np.unique(cps)
    want = max(0, vocab_size - len(roots))
    cand_score = []
    cand_pos = []
    cand_len = []
    h = cps.copy()
    mult = np.uint64(0x9E3779B97F4A7C15)
    for L in range(2, max_cp + 1):
        if n < L:
            break
        with np.errstate(over="ignore"):
            h = (h[:-1] * mult) ^ (cps[L - 1:] + np.uint64(L) * np.uint64(0xD6E8FEB86659FD93))
            h ^= h >> np.uint64(29)
        u, first, cnt = np.unique(h, return_index=True, return_counts=True)
        keep = cnt >= 2
        cand_score.append(cnt[keep].astype(np.int64) * (L - 1))
        cand_pos.append(first[keep])
        cand_len.append(np.full(int(keep.sum()), L, dtype=np.int64))
    tokens = []
    if cand_score and want > 0:
        score = np.concatenate(cand_score)
        pos = np.concatenate(cand_pos)
        ln = np.concatenate(cand_len)
        order = np.lexsort((pos, -score))[:want]
        seen = set()
        for i in order.tolist():
            s = text[int(pos[i]):int(pos[i]) + int(ln[i])]
            if s not in seen:  # distinct strings can share a rolling hash only by collision
                seen.add(s)
                tokens.append(s.encode("utf-8"))
    tokens.extend(chr(int(c)).encode("utf-8") for c in roots.tolist())
    return tokens


def pack_vocab(tokens):
    """list[bytes] -> (bytes u8[], offsets u64[V+1]) for the C-ABI."""
    lens = np.array([len(t) for t in tokens], dtype=np.uint64)
    off = np.zeros(len(tokens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = np.frombuffer(b"".join(tokens), dtype=np.uint8) if int(off[-1]) else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(flat), off


def pack_docs(texts):
    """list[bytes] -> (bytes u8[], doc_offsets u64[D+1])."""
    return pack_vocab(texts)


def sample_vocab(corpus: np.ndarray, vocab_size: int, seed: int = 355, cp_lo: int = 2, cp_hi: int = 10):
    """`vocab_size` distinct tokens of cp_lo..cp_hi code points cut
    from the (ASCII-lower-cased) corpus itself at random code-point positions, so that deep trie
    paths exist; then every single character of the corpus."""
    rng = np.random.default_rng(seed)
    b = np.ascontiguousarray(corpus, dtype=np.uint8).copy()
    up = (b >= 65) & (b <= 90)
    b[up] += 32
    starts = np.flatnonzero((b & 0xC0) != 0x80)  # code-point starts
    n = len(starts)
    ends = np.append(starts, len(b))
    raw = b.tobytes()
    tokens, seen = [], set()
    while len(tokens) < vocab_size and n > cp_hi:
        want = vocab_size - len(tokens)
        pos = rng.integers(0, n - cp_hi, size=want + want // 8 + 16)
        ln = rng.integers(cp_lo, cp_hi + 1, size=len(pos))
        for p, l in zip(pos.tolist(), ln.tolist()):
            t = raw[int(ends[p]):int(ends[p + l])]
            if b"\x00" in t or t in seen:
                continue
            seen.add(t)
            tokens.append(t)
            if len(tokens) == vocab_size:
                break
    singles = sorted({raw[int(ends[i]):int(ends[i + 1])] for i in rng.integers(0, n, size=min(n, 400_000)).tolist()} - seen) if n else []
    return tokens + singles
