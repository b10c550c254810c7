// Sanitizer driver for the host libraries (candidates / annealer / shard reader): built by
// tests/test_native_sanitizers.py with -fsanitize=address,undefined (and once with thread) and
// run on inputs it makes itself.  Exit code 0 = every check held and no sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tokfiles.h"
#include "../../tokenizer_amd/csrc/annealer.h"
#include "../../tokenizer_amd/csrc/candidates.h"

using namespace tokanneal;

#define CHECK(x)                                                         \
	do {                                                                 \
		if (!(x)) {                                                      \
			fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #x); \
			return 1;                                                    \
		}                                                                \
	} while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() {
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return (uint32_t)(rng_state >> 11);
}

int main(int argc, char **argv) {
	const std::string tmp = argc > 1 ? argv[1] : "/tmp";
	const int threads = argc > 2 ? atoi(argv[2]) : 1;
	// a corpus of short documents over a small alphabet with some 2-, 3- and 4-byte characters
	static const char *alpha[] = {"a", "b", "c", "d", "e", " ", "t", "h", "\xc3\xa9", "\xe6\x97\xa5", "\xf0\x9d\x84\x9e"};
	std::vector<uint8_t> bytes;
	std::vector<uint64_t> off{0};
	for (int d = 0; d < 300; d++) {
		const int n = (int)(rnd() % 200);
		for (int k = 0; k < n; k++) {
			const char *s = alpha[rnd() % 11];
			bytes.insert(bytes.end(), s, s + strlen(s));
		}
		off.push_back(bytes.size());
	}
	CandidateSet c;
	BuildCandidates(bytes.data(), off.data(), off.size() - 1, 6, &c);
	CHECK(c.size() > 1000);
	std::string why;
	CHECK(c.Validate(&why));
	for (size_t i = 0; i < c.size(); i += 97) {
		const std::string name = c.GetName(i);
		CHECK(!name.empty() && c.TokenSize(i) >= 1 && c.TokenSize(i) <= 6);
	}
	// the set of two halves of the corpus, merged, is the set of the whole
	{
		const size_t half = (off.size() - 1) / 2;
		std::vector<uint64_t> off_b(off.begin() + half, off.end());
		CandidateSet a, b, m;
		BuildCandidates(bytes.data(), off.data(), half, 6, &a);
		BuildCandidates(bytes.data(), off_b.data(), off_b.size() - 1, 6, &b);
		CHECK(MergeCandidates(a, b, &m, &why));
		CHECK(m.chr == c.chr && m.uses == c.uses && m.lpar == c.lpar && m.rpar == c.rpar);
		std::swap(b.chr.front(), b.chr.back()); // no longer in code-point order: refused, not walked off the end
		CandidateSet x;
		MergeCandidates(a, b, &x, &why);
		CHECK(MergeCandidates(b, CandidateSet(), &x, &why) || !why.empty());
	}
	// file round trip
	const std::string path = tmp + "/san_candidates.bin";
	CHECK(WriteCandidates(c, path, &why));
	CandidateSet back;
	CHECK(ReadCandidates(path, &back) == READ_OK);
	CHECK(back.chr == c.chr && back.uses == c.uses && back.lpar == c.lpar && back.rpar == c.rpar);
	// truncated copies must be refused, never read out of bounds
	{
		FILE *f = fopen(path.c_str(), "rb");
		CHECK(f);
		std::vector<uint8_t> blob(1 << 22);
		blob.resize(fread(blob.data(), 1, blob.size(), f));
		fclose(f);
		const std::string cut = tmp + "/san_cut.bin";
		for (size_t n : {size_t(0), size_t(3), size_t(6), size_t(13), size_t(15), blob.size() / 3, blob.size() - 1}) {
			f = fopen(cut.c_str(), "wb");
			CHECK(f);
			fwrite(blob.data(), 1, n, f);
			fclose(f);
			CandidateSet x;
			CHECK(ReadCandidates(cut, &x) != READ_OK);
		}
	}
	// a file whose token has a parent on one side only must be refused: the annealer indexes a
	// non-root's parents without looking (SimulateStep / ApplyStep would go out of bounds)
	{
		CandidateSet bad = back;
		size_t v = 0;
		while (v < bad.size() && bad.rpar[v] == kNone) v++;
		CHECK(v < bad.size());
		bad.lpar[v] = kNone;
		CHECK(!bad.Validate(&why));
		const std::string bpath = tmp + "/san_bad.bin";
		CHECK(WriteCandidates(bad, bpath, &why));
		CandidateSet x;
		CHECK(ReadCandidates(bpath, &x) != READ_OK);
	}
	// annealing: a few passes, then every move taken back
	const std::vector<uint64_t> uses0 = c.uses;
	const std::vector<uint32_t> rpar0 = c.rpar;
	TokenGenerator gen(std::move(c), 200, 16);
	gen.Generate(2, (unsigned)threads, 7);
	const GeneratorState st = gen.State();
	CHECK(st.gen_cnt > 0 && st.enabled_cnt > 0);
	const std::vector<std::string> sol = gen.GetSolution();
	CHECK(sol.size() >= st.enabled_cnt);
	if (threads == 1) {
		for (size_t i = 0; i < gen.size(); i++)
			if (rpar0[i] != kNone && gen.enabled(i)) gen.ApplyStep(i, false);
		for (size_t i = 0; i < gen.size(); i++) CHECK(gen.l_uses(i) == uses0[i] && gen.r_uses(i) == uses0[i]);
	}
	// shard reader: a valid shard with escapes, then every prefix of it (all invalid, none may crash)
	const std::string shard = "[{\"id\":\"1\",\"title\":\"t\\u00e9\",\"text\":\"a\\n\\\"b\\\\ \\ud834\\udd1e x\\u0000y\"},"
	                          " {\"text\":\"second\",\"id\":\"2\",\"n\":[1,2.5e3,{\"k\":null}],\"title\":\"\"}]";
	const std::string spath = tmp + "/san_shard.json";
	for (size_t n = shard.size();; n--) {
		FILE *f = fopen(spath.c_str(), "wb");
		CHECK(f);
		fwrite(shard.data(), 1, n, f);
		fclose(f);
		tokfiles_shard *s = nullptr;
		int valid = -1;
		CHECK(tokfiles_shard_read(&s, spath.c_str(), &valid) == TOKFILES_OK);
		CHECK(valid == (n == shard.size() ? 1 : 0));
		if (valid) {
			CHECK(tokfiles_shard_docs(s) == 2);
			std::vector<uint8_t> b(tokfiles_shard_bytes(s, TOKFILES_TEXT) + 1);
			std::vector<uint64_t> o(3);
			CHECK(tokfiles_shard_copy(s, TOKFILES_TEXT, b.data(), o.data()) == TOKFILES_OK);
			CHECK(o[2] == tokfiles_shard_bytes(s, TOKFILES_TEXT) && o[1] == strlen("a\n\"b\\ \xf0\x9d\x84\x9e x"));
		} else {
			CHECK(tokfiles_shard_docs(s) == 0);
		}
		tokfiles_shard_free(s);
		if (n == 0) break;
	}
	printf("ok\n");
	return 0;
}
