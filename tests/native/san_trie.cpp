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
		for (int k = 0; k < 200; k++) { // toggles, checked in batches
			const uint32_t i = rnd() % V;
			on[i] = !on[i];
			uint64_t entry = 0;
			const int64_t slot = trie.Toggle(i, on[i], &packed, &entry);
			CHECK(slot < (int64_t)packed.slots.size());
			if (k % 50 == 49) CHECK(check_all());
		}
	}
	printf("ok\n");
	return 0;
}
