// candidates.cpp -- see candidates.h.  Host code, no GPU involved.
#include "candidates.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <thread>

namespace tokanneal {

namespace {

const char kVersion[] = "0.4.0"; // src/config.h:3; written with its terminating NUL (Token.cpp:166-167)

// utf8::internal::sequence_length (utf8/core.h:134-147); 0 = not a lead byte
inline int SeqLen(uint8_t lead) {
	if (lead < 0x80) return 1;
	if ((lead >> 5) == 0x6) return 2;
	if ((lead >> 4) == 0xe) return 3;
	if ((lead >> 3) == 0x1e) return 4;
	return 0;
}

// utf8::unchecked::next (utf8/unchecked.h:104-131) with the read clipped at `end` (the reference
// reads past the end of a truncated sequence: undefined there, zeros here)
inline uint32_t NextCp(const uint8_t *&p, const uint8_t *end) {
	auto at = [&](int k) -> uint32_t { return p + k < end ? p[k] : 0u; };
	uint32_t cp = p[0];
	switch (SeqLen(p[0])) {
	case 2:
		cp = ((cp << 6) & 0x7ff) + (at(1) & 0x3f);
		p += 1;
		break;
	case 3:
		cp = ((cp << 12) & 0xffff) + ((at(1) << 6) & 0xfff) + (at(2) & 0x3f);
		p += 2;
		break;
	case 4:
		cp = ((cp << 18) & 0x1fffff) + ((at(1) << 12) & 0x3ffff) + ((at(2) << 6) & 0xfff) + (at(3) & 0x3f);
		p += 3;
		break;
	default:
		break;
	}
	p += 1;
	if (p > end) p = end;
	return cp;
}

// Counting trie keyed by (parent node, code point) in one open-addressing table; node 0 is the
// root.  Replaces the reference's pointer trie with sorted child vectors (Trie.h:20-37): the
// order of children is restored once, after counting.
struct CountTrie {
	std::vector<uint32_t> parent{0}, chr{0};
	std::vector<uint64_t> freq{0};
	std::vector<uint32_t> table; // node id, 0 = empty
	uint64_t mask = 0;

	CountTrie() { Rehash(1u << 16); }
	static uint64_t Hash(uint32_t p, uint32_t c) {
		uint64_t x = ((uint64_t)p << 32 | c) * 0x9E3779B97F4A7C15ull;
		return x ^ (x >> 29);
	}
	void Rehash(uint64_t cap) {
		table.assign(cap, 0);
		mask = cap - 1;
		for (uint32_t n = 1; n < parent.size(); n++) {
			uint64_t h = Hash(parent[n], chr[n]) & mask;
			while (table[h]) h = (h + 1) & mask;
			table[h] = n;
		}
	}
	uint32_t Find(uint32_t p, uint32_t c) const {
		uint64_t h = Hash(p, c) & mask;
		for (uint32_t n; (n = table[h]) != 0; h = (h + 1) & mask)
			if (parent[n] == p && chr[n] == c) return n;
		return 0;
	}
	uint32_t FindOrAdd(uint32_t p, uint32_t c) {
		uint64_t h = Hash(p, c) & mask;
		for (uint32_t n; (n = table[h]) != 0; h = (h + 1) & mask)
			if (parent[n] == p && chr[n] == c) return n;
		const uint32_t n = (uint32_t)parent.size();
		parent.push_back(p), chr.push_back(c), freq.push_back(0);
		table[h] = n;
		if ((uint64_t)n * 2 > mask) Rehash((mask + 1) * 2);
		return n;
	}
};

} // namespace

void AppendUtf8(uint32_t cp, std::string *s) {
	if (cp < 0x80) {
		s->push_back((char)cp);
	} else if (cp < 0x800) {
		s->push_back((char)((cp >> 6) | 0xc0));
		s->push_back((char)((cp & 0x3f) | 0x80));
	} else if (cp < 0x10000) {
		s->push_back((char)((cp >> 12) | 0xe0));
		s->push_back((char)(((cp >> 6) & 0x3f) | 0x80));
		s->push_back((char)((cp & 0x3f) | 0x80));
	} else {
		s->push_back((char)((cp >> 18) | 0xf0));
		s->push_back((char)(((cp >> 12) & 0x3f) | 0x80));
		s->push_back((char)(((cp >> 6) & 0x3f) | 0x80));
		s->push_back((char)((cp & 0x3f) | 0x80));
	}
}

size_t CandidateSet::TokenSize(size_t i) const {
	size_t len = 1;
	for (uint32_t p = rpar[i]; p != kNone; p = rpar[p]) len++;
	return len;
}

std::string CandidateSet::GetName(size_t i) const {
	std::string s;
	AppendUtf8(chr[i], &s);
	for (uint32_t p = lpar[i]; p != kNone; p = lpar[p]) AppendUtf8(chr[p], &s);
	return s;
}

bool CandidateSet::Validate(std::string *why) const {
	const size_t n = size();
	if (uses.size() != n || lpar.size() != n || rpar.size() != n) {
		if (why) *why = "arrays of different length";
		return false;
	}
	// a token is a root on both sides or on neither (Trie.cpp:145-146): the annealer indexes a
	// non-root's two parents without looking
	for (size_t i = 0; i < n; i++) {
		if ((lpar[i] == kNone) != (rpar[i] == kNone)) {
			if (why) *why = "a token with a parent on one side only";
			return false;
		}
	}
	// each chain (l and r) has to end: walk every token once, a walk that meets itself is a cycle
	for (int side = 0; side < 2; side++) {
		const std::vector<uint32_t> &par = side ? rpar : lpar;
		std::vector<uint8_t> state(n, 0); // 0 new, 1 on the current walk, 2 known to end
		std::vector<uint32_t> path;
		for (size_t i = 0; i < n; i++) {
			path.clear();
			uint32_t v = (uint32_t)i;
			while (v != kNone && state[v] == 0) {
				state[v] = 1;
				path.push_back(v);
				v = par[v];
				if (v != kNone && v >= n) {
					if (why) *why = "parent index out of range";
					return false;
				}
			}
			if (v != kNone && state[v] == 1) {
				if (why) *why = "parent chain does not end";
				return false;
			}
			for (uint32_t u : path) state[u] = 2;
		}
	}
	return true;
}

void BuildCandidates(const uint8_t *bytes, const uint64_t *off, uint64_t n_docs, uint32_t max_cp, CandidateSet *out) {
	out->clear();
	CountTrie t;
	std::vector<uint32_t> cps;
	for (uint64_t d = 0; d < n_docs; d++) {
		// ExtractCandidates (GetTokens.cpp:21-27): utf8to32, then one AddString per position
		cps.clear();
		const uint8_t *p = bytes + off[d], *end = bytes + off[d + 1];
		while (p < end) cps.push_back(NextCp(p, end));
		const size_t n = cps.size();
		for (size_t i = 0; i < n; i++) {
			const size_t len = std::min<size_t>(n - i, max_cp);
			uint32_t node = 0;
			t.freq[0]++; // Trie.cpp:117
			for (size_t k = 0; k < len; k++) {
				node = t.FindOrAdd(node, cps[i + k]);
				t.freq[node]++; // Trie.cpp:122
			}
		}
	}
	const uint32_t nodes = (uint32_t)t.parent.size();
	if (nodes <= 1) return;
	// children of every node, ascending code point (the order FindChild/CreateChild keep, Trie.cpp:16-37)
	std::vector<uint32_t> first(nodes + 1, 0), kids(nodes - 1);
	for (uint32_t n = 1; n < nodes; n++) first[t.parent[n] + 1]++;
	for (uint32_t n = 0; n < nodes; n++) first[n + 1] += first[n];
	{
		std::vector<uint32_t> fill(first.begin(), first.end() - 1);
		for (uint32_t n = 1; n < nodes; n++) kids[fill[t.parent[n]]++] = n;
	}
	for (uint32_t n = 0; n < nodes; n++)
		std::sort(kids.begin() + first[n], kids.begin() + first[n + 1],
		          [&](uint32_t a, uint32_t b) { return t.chr[a] < t.chr[b]; });
	// BuildToken (Trie.cpp:81-91): pre-order, one Token per node, chr = first code point of the string
	const size_t n_tok = nodes - 1;
	out->chr.resize(n_tok), out->uses.resize(n_tok), out->lpar.resize(n_tok), out->rpar.resize(n_tok);
	std::vector<uint32_t> index(nodes, kNone), suffix(nodes, 0), stack;
	for (uint32_t k = first[1]; k-- > first[0];) stack.push_back(kids[k]);
	uint32_t next = 0;
	while (!stack.empty()) {
		const uint32_t n = stack.back();
		stack.pop_back();
		const uint32_t par = t.parent[n];
		index[n] = next;
		out->chr[next] = par == 0 ? t.chr[n] : out->chr[index[par]];
		out->uses[next] = t.freq[n];
		out->rpar[next] = par == 0 ? kNone : index[par]; // Trie.cpp:98
		// CompParents walks the suffix down in step (Trie.cpp:94-96): suffix(n) = suffix(parent) + chr
		suffix[n] = par == 0 ? 0 : t.Find(suffix[par], t.chr[n]);
		next++;
		for (uint32_t k = first[n + 1]; k-- > first[n];) stack.push_back(kids[k]);
	}
	for (uint32_t n = 1; n < nodes; n++) out->lpar[index[n]] = suffix[n] == 0 ? kNone : index[suffix[n]]; // Trie.cpp:99
}

bool WriteCandidates(const CandidateSet &c, const std::string &path, std::string *why) {
	const size_t n = c.size();
	for (size_t i = 0; i < n; i++)
		if (c.uses[i] == 0) {
			if (why) *why = "token " + std::to_string(i) + " has no uses: the format cannot hold it";
			return false;
		}
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) {
		if (why) *why = "cannot open " + path;
		return false;
	}
	std::string buf;
	buf.append(kVersion, sizeof kVersion); // with the NUL
	const uint64_t cnt = n;
	buf.append((const char *)&cnt, sizeof cnt);
	bool ok = true;
	auto flush = [&](bool force) {
		if (ok && (force || buf.size() > (1u << 20))) {
			ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
			buf.clear();
		}
	};
	for (size_t i = 0; i < n; i++) {
		AppendUtf8(c.chr[i], &buf);
		for (uint64_t v = c.uses[i]; v;) { // Token.cpp:177-183
			uint8_t b = v & 0x7F;
			v >>= 7;
			if (v) b |= 0x80;
			buf.push_back((char)b);
		}
		flush(false);
	}
	for (size_t i = 0; i < n; i++) {
		const uint32_t lr[2] = {c.lpar[i], c.rpar[i]};
		buf.append((const char *)lr, sizeof lr);
		flush(false);
	}
	flush(true);
	ok = (fclose(f) == 0) && ok;
	if (!ok && why) *why = "write to " + path + " failed";
	return ok;
}

ReadCode ReadCandidates(const std::string &path, CandidateSet *out) {
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return READ_BAD_STREAM;
	std::vector<uint8_t> data;
	{
		uint8_t tmp[1 << 16];
		size_t got;
		while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) data.insert(data.end(), tmp, tmp + got);
	}
	fclose(f);
	const uint8_t *p = data.data(), *end = p + data.size();
	if ((size_t)(end - p) < sizeof kVersion || memcmp(p, kVersion, sizeof kVersion) != 0) return READ_INVALID_VERSION;
	p += sizeof kVersion;
	out->clear(); // from here on the output is overwritten (Token.h:26-27)
	uint64_t n = 0;
	if ((size_t)(end - p) < sizeof n) return READ_BAD_FILE;
	memcpy(&n, p, sizeof n);
	p += sizeof n;
	if (n == 0) return READ_OK; // Token.cpp:121 (trailing bytes are not looked at)
	if (n >= kNone || n > (uint64_t)(end - p)) return READ_BAD_FILE; // a token takes 2 bytes at least
	out->chr.reserve(n), out->uses.reserve(n);
	for (uint64_t i = 0; i < n; i++) {
		if (p >= end) return READ_BAD_FILE;
		const int len = std::max(1, SeqLen(*p));
		if (end - p < len) return READ_BAD_FILE;
		out->chr.push_back(NextCp(p, end));
		uint64_t uses = 0;
		int sft = 0;
		uint8_t byte;
		do {
			if (p >= end || sft > 63) return READ_BAD_FILE;
			byte = *p++;
			uses |= (uint64_t)(byte & 0x7F) << sft;
			sft += 7;
		} while (byte & 0x80);
		out->uses.push_back(uses);
	}
	if ((uint64_t)(end - p) < n * 8) return READ_BAD_FILE;
	out->lpar.resize(n), out->rpar.resize(n);
	for (uint64_t i = 0; i < n; i++) {
		memcpy(&out->lpar[i], p, 4);
		memcpy(&out->rpar[i], p + 4, 4);
		p += 8;
	}
	if (p != end) return READ_NOT_DONE; // Token.cpp:159
	if (!out->Validate(nullptr)) return READ_BAD_FILE;
	return READ_OK;
}

// ---- union of two candidate sets --------------------------------------------------------
// What GetTokens builds when FileCandidates (GetTokens.cpp:31-90) feeds several files into
// one Trie: the windows of all documents, each string once, its uses added up, numbered in
// pre-order with children by code point (Trie.cpp:81-91,137-142).  Here the two tries are
// walked side by side in that order; both inputs have to be in it already.
namespace {

struct PreorderView {
	const CandidateSet &c;
	std::vector<uint32_t> last; // LAST code point of each string: chr at the end of its l chain
	std::vector<uint32_t> span; // tokens in the subtree of each token, itself included
	explicit PreorderView(const CandidateSet &set) : c(set) {}

	bool Prepare(std::string *why) {
		const size_t n = c.size();
		last.assign(n, 0);
		span.assign(n, 1);
		std::vector<uint8_t> known(n, 0);
		std::vector<uint32_t> path;
		for (size_t i = 0; i < n; i++) {
			if (c.rpar[i] != kNone && c.rpar[i] >= i) {
				if (why) *why = "not in trie pre-order: a token comes before its prefix";
				return false;
			}
			path.clear();
			uint32_t v = (uint32_t)i;
			while (!known[v] && c.lpar[v] != kNone) {
				path.push_back(v);
				v = c.lpar[v];
			}
			if (!known[v]) last[v] = c.chr[v], known[v] = 1;
			for (uint32_t u : path) last[u] = last[v], known[u] = 1;
		}
		for (size_t i = n; i-- > 0;)
			if (c.rpar[i] != kNone) span[c.rpar[i]] += span[i];
		// pre-order: the subtree of a token is the span right behind it
		for (size_t i = 0; i < n; i++) {
			const uint32_t p = c.rpar[i];
			if (p != kNone && i >= (size_t)p + span[p]) {
				if (why) *why = "not in trie pre-order: a token lies outside its prefix's subtree";
				return false;
			}
		}
		return true;
	}
};

} // namespace

bool MergeCandidates(const CandidateSet &a, const CandidateSet &b, CandidateSet *out, std::string *why) {
	out->clear();
	// (parent chains that end, indices in range: CandidateSet::Validate, which every way of
	// making a set has already run)
	PreorderView A(a), B(b);
	{
		// the two views are independent and bound by memory latency: one thread each
		bool ok_b = false;
		std::string why_b;
		std::thread tb([&] { ok_b = B.Prepare(&why_b); });
		const bool ok_a = A.Prepare(why);
		tb.join();
		if (!ok_a) return false;
		if (!ok_b) {
			if (why) *why = why_b;
			return false;
		}
	}
	const size_t na = a.size(), nb = b.size();
	std::vector<uint32_t> map_a(na, kNone), map_b(nb, kNone);
	// sibling ranges still to be merged, [cur, end) on either side; children of one merged
	// token are pushed when it is numbered, so the numbering is the pre-order of the union
	struct Frame {
		uint32_t a_cur, a_end, b_cur, b_end;
	};
	std::vector<Frame> stack;
	stack.push_back({0, (uint32_t)na, 0, (uint32_t)nb});
	uint64_t next = 0;
	auto order_error = [&] {
		if (why) *why = "not in trie pre-order: children out of code-point order";
		out->clear();
		return false;
	};
	while (!stack.empty()) {
		Frame &f = stack.back();
		const bool has_a = f.a_cur < f.a_end, has_b = f.b_cur < f.b_end;
		if (!has_a && !has_b) {
			stack.pop_back();
			continue;
		}
		const uint32_t ca = has_a ? A.last[f.a_cur] : 0, cb = has_b ? B.last[f.b_cur] : 0;
		if (has_a && has_b && ca == cb) {
			const uint32_t x = f.a_cur, y = f.b_cur;
			map_a[x] = map_b[y] = (uint32_t)next++;
			f.a_cur += A.span[x];
			f.b_cur += B.span[y];
			if ((f.a_cur < f.a_end && A.last[f.a_cur] <= ca) || (f.b_cur < f.b_end && B.last[f.b_cur] <= cb)) return order_error();
			stack.push_back({x + 1, x + A.span[x], y + 1, y + B.span[y]}); // (invalidates f)
		} else if (has_a && (!has_b || ca < cb)) {
			// a subtree only one side has keeps its own pre-order
			const uint32_t x = f.a_cur;
			for (uint32_t i = x; i < x + A.span[x]; i++) map_a[i] = (uint32_t)next++;
			f.a_cur += A.span[x];
			if (f.a_cur < f.a_end && A.last[f.a_cur] <= ca) return order_error();
		} else {
			const uint32_t y = f.b_cur;
			for (uint32_t i = y; i < y + B.span[y]; i++) map_b[i] = (uint32_t)next++;
			f.b_cur += B.span[y];
			if (f.b_cur < f.b_end && B.last[f.b_cur] <= cb) return order_error();
		}
		if (next >= kNone) {
			if (why) *why = "more than 2^32 - 2 tokens";
			out->clear();
			return false;
		}
	}
	for (uint32_t m : map_a)
		if (m == kNone) return order_error();
	for (uint32_t m : map_b)
		if (m == kNone) return order_error();
	const size_t n = (size_t)next;
	out->chr.assign(n, 0);
	out->uses.assign(n, 0);
	out->lpar.assign(n, kNone);
	out->rpar.assign(n, kNone);
	// both sets hold every suffix and prefix of their strings, so a token's parents come
	// from the set it came from
	// (the tokens of one set land on different merged tokens, so its slices can go in parallel;
	// the two sets share merged tokens and go one after the other)
	auto place = [&](const CandidateSet &c, const std::vector<uint32_t> &map) {
		const size_t total = c.size();
		const size_t workers = total < (1u << 20) ? 1 : std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency()));
		auto slice = [&](size_t lo, size_t hi) {
			for (size_t i = lo; i < hi; i++) {
				const uint32_t m = map[i];
				out->chr[m] = c.chr[i];
				out->uses[m] += c.uses[i];
				if (c.lpar[i] != kNone) out->lpar[m] = map[c.lpar[i]];
				if (c.rpar[i] != kNone) out->rpar[m] = map[c.rpar[i]];
			}
		};
		std::vector<std::thread> pool;
		for (size_t w = 1; w < workers; w++) pool.emplace_back(slice, total * w / workers, total * (w + 1) / workers);
		slice(0, total / workers);
		for (std::thread &t : pool) t.join();
	};
	place(a, map_a);
	place(b, map_b);
	return true;
}

} // namespace tokanneal
