// Sanitizer driver for the host side of libtokscore.so (trie.cpp: build, hot-first packing,
// toggles, string rows): random vocabularies with duplicates and prefixes, every answer of the
// packed image checked against a plain longest-match over the token list.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../tokenizer_amd/csrc/trie.h"

using namespace tokscore;

#define CHECK(x)                                                         \
	do {                                                                 \
		if (!(x)) {                                                      \
			fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #x); \
			return 1;                                                    \
		}                                                                \
	} while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return (uint32_t)(rng_state >> 11);
}

int main() {
	for (int round = 0; round < 6; round++) {
		const uint32_t V = 50 + rnd() % 3000, alpha = 2 + rnd() % 40, maxl = 1 + rnd() % 12;
		std::vector<std::string> tok(V);
		std::vector<uint8_t> bytes;
		std::vector<uint64_t> off{0};
		for (uint32_t i = 0; i < V; i++) {
			const uint32_t n = rnd() % (maxl + 1); // some empty tokens
			for (uint32_t k = 0; k < n; k++) tok[i].push_back((char)(0x61 + rnd() % alpha + (rnd() % 7 == 0 ? 0x80 : 0)));
			if (i > 3 && rnd() % 9 == 0) tok[i] = tok[rnd() % i]; // duplicates (SolutionFile.cpp:56: the first id wins)
			bytes.insert(bytes.end(), tok[i].begin(), tok[i].end());
			off.push_back(bytes.size());
		}
		std::vector<bool> on(V, true);
		VocabTrie trie;
		std::string err;
		CHECK(trie.Build(bytes.data(), off.data(), V, &err));
		std::vector<uint8_t> text(20000);
		for (auto &b : text) b = (uint8_t)(0x61 + rnd() % alpha + (rnd() % 7 == 0 ? 0x80 : 0));
		if (round & 1) trie.AddHotnessSample(text.data(), text.size());
		PackedTrie packed;
		CHECK(trie.Pack(&packed, &err));
		std::vector<int32_t> slot_row, tok_row;
		const uint32_t rows = trie.StringRows(packed.slots.size(), &slot_row, &tok_row);
		CHECK(rows > 0 && tok_row.size() == V && slot_row.size() == packed.slots.size());
		const std::vector<uint8_t> &text_in = text;
		auto check_all = [&]() -> bool {
			for (size_t p = 0; p < text.size(); p += 7) {
				uint64_t len = 0, id = 0;
				VocabTrie::Longest(packed, text.data() + p, text.size() - p, &len, &id);
				uint64_t want_len = 0, want_id = 0;
				for (uint32_t i = 0; i < V; i++) {
					const std::string &t = tok[i];
					if (!on[i] || t.empty() || t.size() > text.size() - p || t.size() < want_len) continue;
					if (memcmp(t.data(), text.data() + p, t.size()) != 0) continue;
					if (t.size() > want_len) want_len = t.size(), want_id = i + 2; // equal length: the earlier index stays
				}
				if (len != want_len || (len && id != want_id)) {
					fprintf(stderr, "round mismatch at %zu: len %llu/%llu id %llu/%llu\n", p, (unsigned long long)len,
					        (unsigned long long)want_len, (unsigned long long)id, (unsigned long long)want_id);
					return false;
				}
			}
			return true;
		};
		CHECK(check_all());
		// The stream scan's image (trie.h: StreamImage) walked the way scan_stream_kernel walks it --
		// documents lower-cased with a 0x00 after each, no limit compare, no "nothing matched" state --
		// must count what the plain greedy loop over Longest() counts (SolutionFile.cpp:89-97).
		StreamImage img;
		CHECK(trie.PackStream(&img));
		auto check_stream = [&]() -> bool {
			// (documents hold no NUL, DataFile.cpp:27-29; the vocabulary may: such tokens can never match)
			std::vector<uint8_t> text = text_in;
			for (auto &b : text)
				if (b == 0) b = 1;
			std::vector<uint64_t> doc{0};
			for (size_t p = 0; p < text.size();) {
				p += (rnd() % 5 == 0) ? 0 : 1 + rnd() % 900; // some empty documents
				doc.push_back(p < text.size() ? p : text.size());
			}
			std::vector<uint64_t> want(V + 3, 0), got(V + 3, 0);
			uint64_t want_total = 0, got_total = 0;
			std::vector<uint8_t> prep;
			for (size_t d = 0; d + 1 < doc.size(); d++) {
				for (uint64_t p = doc[d]; p < doc[d + 1];) {
					uint64_t len = 0, id = 0;
					VocabTrie::Longest(packed, text.data() + p, doc[d + 1] - p, &len, &id);
					want[len ? id : V + 2]++;
					want_total++;
					p += len ? len : 1;
				}
				for (uint64_t p = doc[d]; p < doc[d + 1]; p++) {
					uint8_t b = text[p];
					prep.push_back(img.code[(b >= 'A' && b <= 'Z') ? (uint8_t)(b + 32) : b]);
				}
				prep.push_back(0);
			}
			prep.resize(prep.size() + 64, 0);
			const uint32_t end = (uint32_t)(prep.size() - 64);
			uint32_t cp = 0, lend = 0, lslot = 0, nb = img.root_base;
			while (cp < end) {
				const uint32_t c = prep[cp];
				const uint32_t t = nb + c;
				if (t >= img.lt.size()) return false;
				const uint32_t x = img.lt[t] ^ c;
				const uint32_t cp1 = cp + 1;
				if (x == 0x80u) lend = cp1, lslot = t;
				nb = (x & 0x7Fu) == 0 ? img.base[t] : 0u;
				if (nb) {
					cp = cp1;
					continue;
				}
				cp = lend;
				nb = img.root_base;
				const uint32_t id = img.slot_id[lslot];
				if (id == StreamImage::kSkipSlot) continue; // document separator
				if (id >= V + 3) return false;
				got[id]++;
				got_total++;
			}
			if (cp != end || got_total != want_total) {
				fprintf(stderr, "stream walk: total %llu/%llu\n", (unsigned long long)got_total, (unsigned long long)want_total);
				return false;
			}
			return got == want;
		};
		CHECK(check_stream());
		for (int k = 0; k < 200; k++) { // toggles, checked in batches
			const uint32_t i = rnd() % V;
			on[i] = !on[i];
			uint64_t entry = 0;
			const int64_t slot = trie.Toggle(i, on[i], &packed, &entry);
			CHECK(slot < (int64_t)packed.slots.size());
			const int64_t sslot = trie.PatchStream(i, &img);
			CHECK(sslot < (int64_t)img.lt.size());
			if (k % 50 == 49) CHECK(check_all() && check_stream());
		}
	}
	printf("ok\n");
	return 0;
}
