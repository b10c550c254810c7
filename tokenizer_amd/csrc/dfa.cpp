// dfa.cpp -- see dfa.h.  Host code only.
#include "dfa.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <thread>

#include "trie.h"

namespace tokscore {

namespace {
inline uint8_t ascii_lower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; } // SolutionFile.cpp:87-88

struct PhaseTimer { // (TOKSCORE_TRACE_HOST: where the builder's time goes)
	bool on = getenv("TOKSCORE_TRACE_HOST") != nullptr;
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	void mark(const char *what) {
		if (!on) return;
		const auto n = std::chrono::steady_clock::now();
		fprintf(stderr, "BuildDfa: %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
		t = n;
	}
};

struct Entry {
	uint32_t target; // state
	uint32_t arr;    // event emitted on arrival (virtual entries), kEvNone for a real edge
	uint8_t code;
	uint8_t virt;
};
} // namespace

double DfaImage::hot_fraction() const {
	if (sample_steps == 0) return 1.0;
	return 1.0 - (double)sample_cold / (double)sample_steps;
}

void DfaImage::event_ids(uint32_t ev, std::vector<uint32_t> *out) const {
	if (ev == kEvNone) return;
	if (!(ev & kEvList)) {
		out->push_back(ev);
		return;
	}
	const uint32_t o = ev & ~kEvList;
	for (uint32_t i = 0; i < pool[o]; i++) out->push_back(pool[o + 1 + i]);
}

bool VocabTrie::BuildDfa(DfaImage *out, const DfaOptions &opt, std::string *why) const {
	out->valid = false;
	auto fail = [&](const char *msg) {
		if (why) *why = msg;
		return false;
	};
	const uint32_t N = (uint32_t)first_child_.size();
	if (N == 0) return fail("no vocabulary");
	PhaseTimer pt;
	// ---- codes: 0 = separator, 1 .. k = the bytes that label an edge, k + 1 = every other byte ----
	bool is_label[256] = {};
	for (const Edge &e : edges_) is_label[e.byte] = true;
	is_label[0] = false; // documents hold no NUL (DataFile.cpp:27-29): such edges can never be followed
	uint32_t k = 0;
	for (uint32_t b = 1; b < 256; b++) k += is_label[b] ? 1 : 0;
	const uint32_t ncodes = k + 2;
	const bool fm = opt.fail_mode;
	const uint32_t failc = ncodes;              // (failure form) the pseudo-code of a row's failure entry
	const uint32_t ncr = ncodes + (fm ? 1 : 0); // codes a row can hold an entry for
	if (ncr > DfaImage::kMaxCodes) return fail("too many distinct edge bytes (the label is a byte, all ones = empty)");
	const uint32_t other = k + 1;
	{
		uint32_t next = 1;
		out->code[0] = 0;
		for (uint32_t b = 1; b < 256; b++) out->code[b] = (uint8_t)(is_label[b] ? next++ : other);
	}
	const uint8_t *code = out->code;
	// ---- states: the trie nodes of depth >= 1 that can be reached, and a stand-in for every code
	// without a root child (the separator, "other", bytes that only occur inside tokens) ----
	const uint32_t ROOT = 0;
	const uint32_t S = N + ncodes; // state ids: [1, N) nodes, N + c stand-in of code c
	if (S >= (1u << 24)) return fail("too many trie nodes (entry keys hold a state in 24 bits)");
	std::vector<uint32_t> par(N, 0);
	std::vector<uint8_t> pbyte(N, 0), reach(N, 0);
	reach[0] = 1;
	for (uint32_t n = 0; n < N; n++) {
		const Edge *e = edges_.data() + first_child_[n];
		for (uint32_t i = 0; i < n_children_[n]; i++) {
			par[e[i].child] = n;
			pbyte[e[i].child] = e[i].byte;
			if (reach[n] && e[i].byte != 0) reach[e[i].child] = 1;
		}
	}
	std::vector<uint32_t> d1(ncodes);
	for (uint32_t c = 0; c < ncodes; c++) d1[c] = N + c;
	for (uint32_t b = 1; b < 256; b++)
		if (is_label[b] && root_child_[b]) d1[code[b]] = root_child_[b];
	const uint32_t unknown = (uint32_t)(V_ + 2);
	auto id_end = [&](uint32_t s) -> uint32_t { // 0 = no token ends here
		if (s >= N) return s == N ? 0u : unknown; // the separator counts for nothing
		const uint32_t id = ActiveId(s);
		return id ? id : (depth_[s] == 1 ? unknown : 0u);
	};
	auto child = [&](uint32_t s, uint8_t b) -> uint32_t { return (s < N && b != 0) ? Child(s, b) : 0u; };
	// ---- F, f: what falling off at a state decides ----
	std::vector<uint32_t> f(S, ROOT), le(N, 0);
	std::vector<uint32_t> F_off(S + 1, 0), F_dat;
	F_dat.reserve(S + S / 4);
	std::vector<uint8_t> str;
	auto F_append = [&](uint32_t s, std::vector<uint32_t> *to) {
		for (uint32_t i = F_off[s]; i < F_off[s + 1]; i++) to->push_back(F_dat[i]);
	};
	std::vector<uint32_t> pops;
	// nodes are numbered breadth-first: everything a node's walk below meets is shallower and done.
	// F lists are appended in state order, so the stand-ins (state ids >= N) are done first in a
	// scratch copy and appended after the nodes.
	std::vector<std::vector<uint32_t>> F_stand(ncodes);
	for (uint32_t c = 0; c < ncodes; c++)
		if (id_end(N + c)) F_stand[c].push_back(id_end(N + c));
	auto F_of = [&](uint32_t s, std::vector<uint32_t> *to) {
		if (s >= N) to->insert(to->end(), F_stand[s - N].begin(), F_stand[s - N].end());
		else F_append(s, to);
	};
	auto delta = [&](uint32_t st, uint8_t b, std::vector<uint32_t> *emit) -> uint32_t {
		for (;;) {
			if (st == ROOT) return d1[code[b]];
			const uint32_t ch = child(st, b);
			if (ch) return ch;
			F_of(st, emit);
			st = f[st];
		}
	};
	for (uint32_t n = 1; n < N; n++) {
		F_off[n] = (uint32_t)F_dat.size();
		if (!reach[n]) continue;
		const uint32_t d = depth_[n];
		le[n] = id_end(n) ? n : le[par[n]];
		if (d == 1) {
			F_dat.push_back(id_end(n));
			continue;
		}
		str.resize(d);
		for (uint32_t x = n, i = d; x != 0; x = par[x]) str[--i] = pbyte[x];
		pops.clear();
		pops.push_back(id_end(le[n]));
		uint32_t st = ROOT;
		for (uint32_t i = depth_[le[n]]; i < d; i++) st = delta(st, str[i], &pops);
		f[n] = st;
		F_dat.insert(F_dat.end(), pops.begin(), pops.end());
	}
	F_off[N] = (uint32_t)F_dat.size();
	for (uint32_t c = 0; c < ncodes; c++) {
		F_dat.insert(F_dat.end(), F_stand[c].begin(), F_stand[c].end());
		F_off[N + c + 1] = (uint32_t)F_dat.size();
	}
	pt.mark("failure function");
	// ---- events ----
	out->pool.clear();
	uint32_t max_events = 1;
	auto encode = [&](const std::vector<uint32_t> &ids) -> uint32_t {
		if (ids.empty()) return DfaImage::kEvNone;
		if (ids.size() == 1) return ids[0];
		max_events = std::max<uint32_t>(max_events, (uint32_t)ids.size());
		const uint32_t o = (uint32_t)out->pool.size();
		out->pool.push_back((uint32_t)ids.size());
		out->pool.insert(out->pool.end(), ids.begin(), ids.end());
		return DfaImage::kEvList | o;
	};
	// M(s) = F(s) ++ M(f(s)): everything emitted when the whole chain is followed to the root
	std::vector<uint32_t> M(S, DfaImage::kEvNone);
	{
		std::vector<uint32_t> ids;
		auto chain_ids = [&](uint32_t s) {
			ids.clear();
			if (fm) F_of(s, &ids); // (failure form: a miss emits F(s) alone, f is followed state by state)
			else
				for (uint32_t x = s; x != ROOT; x = f[x]) F_of(x, &ids);
		};
		for (uint32_t s = 1; s < S; s++) {
			if (s < N && !reach[s]) continue;
			chain_ids(s);
			M[s] = encode(ids);
		}
	}
	pt.mark("event lists");
	// ---- rows: real children, then the virtual entries of f's chain ----
	std::vector<uint64_t> row_off(S + 1, 0);
	std::vector<Entry> ent;
	{
		std::vector<uint32_t> acc;
		uint8_t seen[256];
		uint64_t budget = opt.max_slots + (uint64_t)ncodes * ncodes;
		for (uint32_t s = 1; s < S; s++) {
			row_off[s] = ent.size();
			if (s >= N || !reach[s]) continue;
			memset(seen, 0, sizeof seen);
			const Edge *e = edges_.data() + first_child_[s];
			for (uint32_t i = 0; i < n_children_[s]; i++) {
				if (e[i].byte == 0) continue;
				seen[e[i].byte] = 1;
				ent.push_back(Entry{e[i].child, DfaImage::kEvNone, code[e[i].byte], 0});
			}
			if (fm) {
				if (f[s] != ROOT) ent.push_back(Entry{f[s], DfaImage::kEvNone, (uint8_t)failc, 0});
			} else if (f[s] != ROOT) {
				acc.clear();
				F_of(s, &acc);
				uint32_t arr_ev = 0;
				bool arr_done = false;
				for (uint32_t v = f[s]; v != ROOT; v = f[v]) {
					if (v < N) {
						const Edge *ve = edges_.data() + first_child_[v];
						for (uint32_t i = 0; i < n_children_[v]; i++) {
							const uint8_t b = ve[i].byte;
							if (b == 0 || seen[b]) continue;
							seen[b] = 1;
							if (!arr_done) arr_ev = encode(acc), arr_done = true;
							ent.push_back(Entry{ve[i].child, arr_ev, code[b], 1});
						}
					}
					F_of(v, &acc);
					arr_done = false;
				}
			}
			if (ent.size() > budget) return fail("the automaton needs more slots than the limit");
			if (!fm && opt.max_virtual_per_node && ent.size() > (uint64_t)N * (1 + opt.max_virtual_per_node))
				return fail("more virtual entries per trie node than the caller takes");
			std::sort(ent.begin() + (std::ptrdiff_t)row_off[s], ent.end(), [](const Entry &a, const Entry &b) { return a.code < b.code; });
		}
		row_off[S] = ent.size();
	}
	auto row_find = [&](uint32_t s, uint8_t c) -> int64_t {
		uint64_t lo = row_off[s], hi = row_off[s + 1];
		while (lo < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			if (ent[mid].code < c) lo = mid + 1;
			else hi = mid;
		}
		return (lo < row_off[s + 1] && ent[lo].code == c) ? (int64_t)lo : -1;
	};
	pt.mark("rows");
	// ---- hotness: the sample run through the automaton (document boundaries ignored: layout only) ----
	std::vector<uint64_t> w_row(S, 0);
	std::vector<uint64_t> w_ent(ent.size(), 0);
	std::vector<uint32_t> w_d1(ncodes, 0);
	const bool reuse = opt.reuse && opt.reuse != out && opt.reuse->vocab_tag == opt.vocab_tag && opt.reuse->sample_steps != 0 &&
	                   opt.reuse->sample_w_ent.size() == ent.size() && opt.reuse->sample_w_row.size() == S;
	if (reuse) {
		for (size_t j = 0; j < ent.size(); j++) w_ent[j] = opt.reuse->sample_w_ent[j];
		w_row = opt.reuse->sample_w_row;
		out->sample_steps = opt.reuse->sample_steps, out->sample_virtual = opt.reuse->sample_virtual, out->sample_miss = opt.reuse->sample_miss;
	} else out->sample_steps = out->sample_virtual = out->sample_miss = 0;
	if (!reuse && opt.sample && opt.sample_bytes) {
		// (a table entry-of(state, code) while it stays small: the search in the row cost 40 ns per sample byte)
		std::vector<uint32_t> dense;
		if ((uint64_t)S * ncr <= (32u << 20)) {
			dense.assign((size_t)S * ncr, 0);
			for (uint32_t s = 1; s < S; s++)
				for (uint64_t j = row_off[s]; j < row_off[s + 1]; j++) dense[(size_t)s * ncr + ent[j].code] = (uint32_t)j + 1;
		}
		auto find = [&](uint32_t st, uint32_t c) -> int64_t { return dense.empty() ? row_find(st, (uint8_t)c) : (int64_t)dense[(size_t)st * ncr + c] - 1; };
		pt.mark("sample: table");
		// (the sample in up to eight pieces on threads of their own, each starting in the separator's state and counting
		// into arrays of its own: layout only, so a piece's first few steps from the wrong state do not matter, and
		// 2 MB through a 32 MB table on one thread was 23 of an image's 44 ms)
		struct Counts {
			std::vector<uint32_t> row, en, d1c; // (a piece is a fraction of 2 MB: 32 bits hold its counts)
			uint64_t steps = 0, virt = 0, miss = 0;
		};
		auto run = [&](uint64_t begin, uint64_t end, Counts &k) {
			uint32_t st = d1[0];
			for (uint64_t i = begin; i < end; i++) {
				const uint8_t c = code[ascii_lower(opt.sample[i])];
				k.steps++;
				if (st == d1[0]) { // the root row (after a separator): every code is there
					k.d1c[c]++;
					st = d1[c];
					continue;
				}
				k.row[st]++;
				int64_t j = find(st, c);
				if (fm) {
					// (failure form: a miss is followed by the look at the row's failure entry, which counts as a lookup of
					// the row and a use of that entry; the byte stays until some state takes it or the root is reached)
					while (j < 0 && f[st] != ROOT) {
						k.miss++;
						const int64_t jf = find(st, failc);
						k.row[st]++;
						k.en[jf]++;
						st = ent[jf].target;
						k.row[st]++;
						k.steps++;
						j = find(st, c);
					}
				}
				if (j >= 0) {
					k.en[j]++;
					k.virt += ent[j].virt;
					st = ent[j].target;
				} else {
					k.miss++;
					k.d1c[c]++;
					st = d1[c];
				}
			}
		};
		unsigned pieces = opt.sample_bytes >= (1u << 20) ? 8u : opt.sample_bytes >= (1u << 18) ? 4u : 1u;
		if (const char *e = getenv("TOKSCORE_SAMPLE_THREADS")) pieces = (unsigned)std::max(1, std::min(16, atoi(e)));
		std::vector<Counts> cnt(pieces);
		for (auto &k : cnt) k.row.assign(S, 0), k.en.assign(ent.size(), 0), k.d1c.assign(ncodes, 0);
		{
			std::vector<std::thread> th;
			for (unsigned t = 1; t < pieces; t++)
				th.emplace_back(run, opt.sample_bytes * t / pieces, opt.sample_bytes * (t + 1) / pieces, std::ref(cnt[t]));
			run(0, opt.sample_bytes / pieces, cnt[0]);
			for (auto &x : th) x.join();
		}
		for (const Counts &k : cnt) {
			for (size_t s2 = 0; s2 < S; s2++) w_row[s2] += k.row[s2];
			for (size_t j = 0; j < ent.size(); j++) w_ent[j] += k.en[j];
			for (uint32_t c = 0; c < ncodes; c++) w_d1[c] += k.d1c[c];
			out->sample_steps += k.steps, out->sample_virtual += k.virt, out->sample_miss += k.miss;
		}
	}
	const bool sampled = out->sample_steps != 0;
	// what an earlier image counted on the whole corpus: uses of entries LDS did not hold.  The sample's
	// counts are brought to the corpus's scale first, so that the two can be compared.
	out->vocab_tag = opt.vocab_tag;
	if (!reuse) {
		out->sample_w_row = w_row;
		out->sample_w_ent.resize(ent.size());
		for (size_t j = 0; j < ent.size(); j++) out->sample_w_ent[j] = (uint32_t)std::min<uint64_t>(w_ent[j], 0xFFFFFFFFull);
	} else if (out->sample_w_ent.empty()) {
		out->sample_w_ent = opt.reuse->sample_w_ent;
		out->sample_w_row = opt.reuse->sample_w_row;
	}
	pt.mark("sample");
	const bool have_prior = opt.prior_keys && opt.prior_counts && opt.prior_n != 0;
	if (have_prior) {
		if (sampled && opt.corpus_bytes > opt.sample_bytes) {
			const uint64_t scale = opt.corpus_bytes / opt.sample_bytes;
			for (auto &w : w_row) w *= scale;
			for (auto &w : w_ent) w *= scale;
		}
		size_t k = 0; // (entries are in key order: rows by state, a row's entries by code)
		for (uint32_t s = 1; s < S && k < opt.prior_n; s++)
			for (uint64_t j = row_off[s]; j < row_off[s + 1] && k < opt.prior_n; j++) {
				const uint32_t key = s << 8 | ent[j].code;
				while (k < opt.prior_n && opt.prior_keys[k] < key) k++;
				if (k == opt.prior_n || opt.prior_keys[k] != key) continue;
				const uint32_t n = opt.prior_counts[k];
				w_ent[j] += n;
				w_row[s] += n;
				w_row[ent[j].target] += n;
			}
	}
	const bool weighted = sampled || have_prior;
	// ---- which entries and rows are worth LDS ----
	// The global array holds the whole automaton: every row complete, beyond hot_cap.  The LDS array
	// is an overlay for the rows in use, the rows with the most lookups per LDS entry first, for as
	// long as entries fit: a row is there COMPLETE, or PARTIAL -- only the entries in use; a miss in a
	// partial row proves nothing, so it is looked up again in the row's complete copy -- whichever is
	// worth the slots: the rows with the most misses per unused entry are completed while slots are left.
	const uint32_t hot_cap = std::min<uint32_t>(std::max<uint32_t>(opt.lds_slots, 2 * ncr + 64), DfaImage::kMaxHot);
	std::vector<uint32_t> nhot(S, 0), ncold(S, 0);
	std::vector<uint64_t> w_miss(S, 0); // lookups of a row that miss
	std::vector<uint8_t> is_d1(S, 0), klass(S, 0); // klass: 0 global array only, 1 complete in LDS, 2 partial in LDS
	std::vector<uint8_t> sel(ent.size(), 0);       // the entry is one of those a partial row keeps in LDS
	for (uint32_t c = 0; c < ncodes; c++) is_d1[d1[c]] = 1;
	if (!weighted) { // nothing to go by: breadth-first, everything counts as used
		for (uint32_t s = 1; s < N; s++) w_row[s] = 1 + (max_len_ + 1 - std::min<uint64_t>(depth_[s], max_len_ + 1));
		std::fill(w_ent.begin(), w_ent.end(), 1u);
	}
	std::vector<uint32_t> owner(ent.size(), 0);
	for (uint32_t s = 1; s < S; s++) {
		uint64_t used_sum = 0;
		for (uint64_t j = row_off[s]; j < row_off[s + 1]; j++) owner[j] = s, used_sum += w_ent[j];
		w_miss[s] = weighted && w_row[s] > used_sum ? w_row[s] - used_sum : 0;
	}
	uint64_t d1_slots = ncodes;
	for (uint32_t c = 1; c < ncodes; c++) d1_slots += row_off[d1[c] + 1] - row_off[d1[c]];
	{
		// What a slot of LDS saves is second looks: an entry kept there saves its uses, the slots that complete
		// a row save the row's misses and the uses of the entries they add (a miss in a partial row is looked up
		// again, a miss in a complete one is a miss).  Everything that saves at least `th` lookups per slot goes
		// in, for the smallest th that fits (first-fit leaves holes: some slack); entries fill what is left.
		std::vector<uint64_t> cand;
		for (uint64_t j = 0; j < ent.size(); j++) {
			if (is_d1[owner[j]]) sel[j] = 1;
			else if (w_ent[j]) cand.push_back(j);
		}
		std::stable_sort(cand.begin(), cand.end(), [&](uint64_t a, uint64_t b) { return w_ent[a] > w_ent[b]; });
		const double room = hot_cap > d1_slots + 2 * ncodes ? (double)(hot_cap - d1_slots - 2 * ncodes) : 0.0;
		auto choose = [&](double th, bool apply) -> double {
			double slots = 0;
			for (uint32_t s = 1; s < S; s++) {
				if (is_d1[s] || row_off[s + 1] == row_off[s]) continue;
				uint32_t n_in = 0, n_out = 0;
				uint64_t w_un = 0;
				for (uint64_t j = row_off[s]; j < row_off[s + 1]; j++) {
					if (w_ent[j] && (double)w_ent[j] >= th) n_in++;
					else n_out++, w_un += w_ent[j];
				}
				const bool complete = n_out != 0 && w_miss[s] + w_un != 0 && (double)(w_miss[s] + w_un) / n_out >= th;
				slots += complete ? n_in + n_out : n_in;
				if (apply)
					for (uint64_t j = row_off[s]; j < row_off[s + 1]; j++) sel[j] = complete || (w_ent[j] && (double)w_ent[j] >= th);
			}
			return slots;
		};
		double lo = 0.25, hi = 1;
		for (uint64_t j : cand) hi = std::max(hi, (double)w_ent[j] + 1);
		for (uint32_t s = 1; s < S; s++) hi = std::max(hi, (double)w_miss[s] + 1);
		if (choose(lo, false) > 0.985 * room) {
			for (int it = 0; it < 48 && hi - lo > 1e-3 * lo; it++) {
				const double mid = std::sqrt(lo * hi);
				if (choose(mid, false) > 0.985 * room) lo = mid;
				else hi = mid;
			}
			lo = hi;
		}
		double taken = choose(lo, true);
		auto count_row = [&](uint32_t s) {
			nhot[s] = ncold[s] = 0;
			for (uint64_t j = row_off[s]; j < row_off[s + 1]; j++) {
				if (sel[j]) nhot[s]++;
				else ncold[s]++;
			}
			klass[s] = nhot[s] ? (ncold[s] ? 2 : 1) : 0;
		};
		for (uint64_t j : cand) {
			if (taken + 1 > 0.995 * room) break;
			if (!sel[j]) sel[j] = 1, taken++;
		}
		for (uint32_t s = 1; s < S; s++)
			if (!is_d1[s] && row_off[s + 1] != row_off[s]) count_row(s);
	}
	std::vector<uint32_t> order; // rows with slots in LDS, in the order they are placed: the most lookups per slot first
	for (uint32_t s = 1; s < S; s++)
		if (!is_d1[s] && klass[s] != 0) order.push_back(s);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
		return (double)w_row[a] / std::max<uint32_t>(1, nhot[a]) > (double)w_row[b] / std::max<uint32_t>(1, nhot[b]);
	});
	pt.mark("choice");
	// ---- placement ----
	struct Occ {
		std::vector<uint8_t> used, base_used;
		std::vector<uint32_t> next_free;
		std::vector<uint64_t> bits; // `used`, one bit per slot: a row's entries are tested against 128 slots at a time
		void ensure(uint64_t n) {
			if (used.size() < n) {
				const uint64_t old = used.size(), m = std::max<uint64_t>(n, std::max<uint64_t>(4096, old * 2));
				used.resize(m, 0);
				bits.resize(m / 64 + 8, 0);
				base_used.resize(m, 0);
				next_free.resize(m);
				for (uint64_t i = old; i < m; i++) next_free[i] = (uint32_t)i;
			}
		}
		uint64_t find_free(uint64_t s) {
			ensure(s + 1);
			uint64_t r = s;
			while (next_free[r] != r) {
				r = next_free[r];
				ensure(r + 1);
			}
			while (next_free[s] != r) {
				const uint64_t t = next_free[s];
				next_free[s] = (uint32_t)r;
				s = t;
			}
			return r;
		}
		bool is_used(uint64_t t) {
			ensure(t + 1);
			return used[t] != 0;
		}
		// does a row whose entries' codes are the bits of m[0 .. 3] fit at base b?
		bool fits(uint64_t b, const uint64_t m[4]) {
			ensure(b + 260);
			const uint64_t i = b >> 6, sh = b & 63;
			uint64_t hit = 0;
			for (int q = 0; q < 4; q++)
				if (m[q]) hit |= (sh ? (bits[i + q] >> sh) | (bits[i + q + 1] << (64 - sh)) : bits[i + q]) & m[q];
			return hit == 0;
		}
		void take(uint64_t t) {
			ensure(t + 2);
			bits[t >> 6] |= 1ull << (t & 63);
			used[t] = 1;
			next_free[t] = (uint32_t)(t + 1);
		}
		bool base_taken(uint64_t b) {
			ensure(b + 1);
			return base_used[b] != 0;
		}
		void take_base(uint64_t b) {
			ensure(b + 1);
			base_used[b] = 1;
		}
		// a base for a state that stores nothing: any value of its own, at or after lo (0: none below hi)
		uint64_t fresh_base(uint64_t lo, uint64_t hi) {
			uint64_t b = std::max<uint64_t>(lo, base_cursor);
			while (base_taken(b)) b++;
			if (hi && b >= hi) return 0;
			base_cursor = b + 1;
			take_base(b);
			max_base = std::max(max_base, b);
			return b;
		}
		uint64_t max_used = 0, max_base = 0, base_cursor = 1;
	};
	Occ H, G; // the LDS array [0, hot_cap) and the global array [hot_cap, ...): bases are unique within each
	std::vector<uint32_t> base_h(S, 0), base_g(S, 0);
	// the root row: dense at base 0, slot c = the depth-1 state of code c
	for (uint32_t c = 0; c < ncodes; c++) H.take(c);
	H.take_base(0);
	// depth-1 rows at c * stride: the smallest stride whose occupied slots are all different and clear of the root row
	uint32_t stride = 0;
	{
		std::vector<uint8_t> tmp;
		for (uint32_t s = 1; s <= ncodes && !stride; s++) {
			tmp.assign((size_t)ncodes * s + ncodes + 1, 0);
			bool ok = true;
			for (uint32_t c = 1; c < ncodes && ok; c++)
				for (uint64_t j = row_off[d1[c]]; j < row_off[d1[c] + 1] && ok; j++) {
					const uint32_t t = c * s + ent[j].code;
					ok = t >= ncodes && !tmp[t];
					tmp[t] = 1;
				}
			if (ok) stride = s;
		}
		if (!stride) stride = ncodes; // (always fits: rows then do not overlap at all)
	}
	if ((uint64_t)(ncodes - 1) * stride + ncodes > hot_cap) return fail("the depth-1 rows do not fit the LDS array");
	std::vector<uint8_t> has_id(S, 0); // the state has an LDS base of its own
	std::vector<uint8_t> in_lds(S, 0); // its row has slots in LDS: 1 every entry, 2 the entries in use
	for (uint32_t c = 0; c < ncodes; c++) {
		base_h[d1[c]] = c * stride;
		klass[d1[c]] = 1;
		has_id[d1[c]] = 1;
		in_lds[d1[c]] = 1;
		H.take_base(c * stride);
		if (c == 0) continue;
		for (uint64_t j = row_off[d1[c]]; j < row_off[d1[c] + 1]; j++) H.take(c * stride + ent[j].code);
	}
	// A row into occupancy `occ` with bases in [lo, hi) (hi = 0: no upper limit); only_used: just the
	// entries in use.  False when nothing fits below hi.
	const uint64_t slot_limit = std::min<uint64_t>(opt.max_slots, DfaImage::kMaxSlots);
	auto place = [&](Occ &occ, uint32_t s, bool only_used, uint64_t lo, uint64_t hi, uint32_t *base_out) -> bool {
		const Entry *e = ent.data() + row_off[s];
		const uint32_t nc = (uint32_t)(row_off[s + 1] - row_off[s]);
		auto in = [&](uint32_t i) { return !only_used || sel[row_off[s] + i] != 0; };
		uint32_t lead = 0;
		while (lead < nc && !in(lead)) lead++;
		if (lead == nc) { // nothing to store (a partial row none of whose entries is in use): any base of its own
			const uint64_t b = occ.fresh_base(lo, hi);
			if (!b) return false;
			*base_out = (uint32_t)b;
			return true;
		}
		const uint32_t c0 = e[lead].code;
		uint64_t mk[4] = {0, 0, 0, 0}; // the codes of the entries to store
		for (uint32_t i = lead; i < nc; i++)
			if (in(i)) mk[e[i].code >> 6] |= 1ull << (e[i].code & 63);
		uint64_t t = occ.find_free(std::max<uint64_t>(lo + c0, (uint64_t)c0 + 1));
		uint64_t b = 0;
		for (uint32_t tries = 0;; tries++) {
			b = t - c0;
			if (hi && b >= hi) return false;
			if (!occ.base_taken(b) && occ.fits(b, mk)) break;
			if (tries >= 24 && !hi) { // (the global array need not be tight: soon beyond everything placed so far -- 512 tries cost 200 ms)
				for (b = std::max<uint64_t>(occ.max_base + 1, lo);; b++) {
					bool fits = !occ.base_taken(b);
					for (uint32_t i = lead; fits && i < nc; i++) fits = !in(i) || !occ.is_used(b + e[i].code);
					if (fits) break;
				}
				break;
			}
			t = occ.find_free(t + 1);
		}
		occ.take_base(b);
		occ.max_base = std::max(occ.max_base, b);
		*base_out = (uint32_t)b;
		for (uint32_t i = lead; i < nc; i++)
			if (in(i)) {
				occ.take(b + e[i].code);
				occ.max_used = std::max<uint64_t>(occ.max_used, b + e[i].code);
			}
		return true;
	};
	const uint64_t hot_hi = hot_cap - ncr; // an LDS row's base: every slot it can probe is below hot_cap
	uint32_t n_sel = 0, n_demoted = 0;
	for (uint32_t s : order) {
		if (klass[s] == 0) continue;
		n_sel++;
		if (klass[s] == 1 && !place(H, s, false, 1, hot_hi, &base_h[s])) klass[s] = ncold[s] ? 2 : 0;
		if (klass[s] == 2 && !place(H, s, true, 1, hot_hi, &base_h[s])) klass[s] = 0;
		n_demoted += klass[s] == 0;
		has_id[s] = klass[s] != 0;
		in_lds[s] = klass[s];
	}
	// the base the states without one of their own share: a value no state has, so no slot "belongs" to it
	uint32_t cold_base = 1;
	while (H.base_taken(cold_base)) cold_base++;
	H.take_base(cold_base);
	// every other state: a base of its own while values last -- the states in use first
	uint32_t n_cold = 0;
	{
		// (the states the depth-1 rows lead to come first: the state after a miss is computed, c * stride, with
		// no word to say "partial", so those rows must be complete in LDS, which takes a base for every target)
		std::vector<uint8_t> must(S, 0);
		for (uint32_t c = 1; c < ncodes; c++)
			for (uint64_t j = row_off[d1[c]]; j < row_off[d1[c] + 1]; j++) must[ent[j].target] = 1;
		std::vector<uint32_t> rest;
		for (uint32_t s = 1; s < S; s++)
			if (!has_id[s] && (s >= N || reach[s])) {
				// (failure form: a state is known by the ENTRY it was entered through, see the image below -- no base of its
				// own is needed, and the states without slots in LDS all look their bytes up at cold_base, where nothing is)
				if (fm) base_h[s] = cold_base, n_cold++;
				else rest.push_back(s);
			}
		std::stable_sort(rest.begin(), rest.end(), [&](uint32_t a, uint32_t b) { return must[a] != must[b] ? must[a] > must[b] : w_row[a] > w_row[b]; });
		for (uint32_t s : rest) {
			const uint64_t b = H.fresh_base(1, hot_hi);
			if (!b) {
				if (must[s]) return fail("the depth-1 rows' targets do not fit the LDS array");
				base_h[s] = cold_base;
				n_cold++;
				continue;
			}
			base_h[s] = (uint32_t)b;
			has_id[s] = 1;
			if (row_off[s + 1] != row_off[s]) klass[s] = 2; // a partial row that stores nothing
			else klass[s] = 1;                            // no row at all: every lookup misses, rightly
		}
	}
	pt.mark("placement (LDS)");
	// the whole automaton in the global array
	G.max_base = hot_cap - 1;
	G.base_cursor = hot_cap;
	for (uint32_t s = 1; s < S; s++) {
		if (row_off[s + 1] == row_off[s]) continue;
		place(G, s, false, hot_cap, 0, &base_g[s]);
		if (G.max_used + ncr + 1 > slot_limit) return fail("the automaton needs more slots than the limit");
	}
	for (uint32_t s = 1; s < S; s++) // states without a row and without an LDS base: told apart by a global base that stores nothing
		if (!fm && !has_id[s] && (s >= N || reach[s]) && row_off[s + 1] == row_off[s]) base_g[s] = (uint32_t)G.fresh_base(hot_cap, 0);
	const uint64_t nslots = std::max<uint64_t>(std::max(G.max_used + 1, G.max_base + ncr + 1), hot_cap); // base + any code is a readable slot
	if (nslots > slot_limit) return fail("the automaton needs more slots than the limit");
	// An entry can be in LDS when its target has a base of its own (the others are only ever entered from
	// the global array, which carries their global base); a row that leaves an entry out is a partial one.
	auto stored = [&](uint32_t s, uint64_t j) { return (in_lds[s] == 1 || (in_lds[s] == 2 && sel[j] != 0)) && (fm || has_id[ent[j].target]); };
	std::vector<uint8_t> partial_row(S, 0);
	for (uint32_t s = 1; s < S; s++) {
		if (row_off[s + 1] == row_off[s]) {
			partial_row[s] = !fm && !has_id[s]; // (failure form: no row, every lookup misses, rightly, at whatever base)
			continue;
		}
		bool all = in_lds[s] == 1;
		for (uint64_t j = row_off[s]; all && j < row_off[s + 1]; j++) all = stored(s, j);
		partial_row[s] = !all;
	}
	pt.mark("placement (global)");
	// ---- the image ----
	out->ncodes = ncodes;
	out->fail_code = fm ? failc : 0u;
	out->stride = stride;
	out->nslots = (uint32_t)nslots;
	out->hot_slots = hot_cap;
	out->cold_base = cold_base;
	out->hot.assign(hot_cap, DfaImage::kEmptyWord);
	out->slot.assign(2 * nslots, 0);
	for (uint64_t i = 0; i < nslots; i++) out->slot[2 * i] = DfaImage::kEmptyWord;
	out->gof.assign(fm ? nslots : hot_cap, 0);
	out->ev_miss.assign(nslots, DfaImage::kEvNone);
	out->ev_arr.assign(nslots, DfaImage::kEvNone);
	out->ent_key.assign(nslots - hot_cap, DfaImage::kEvNone);
	out->real_slots = out->virtual_slots = out->hot_used = 0;
	// the word of a state: where its next byte is looked up
	auto state_word = [&](uint32_t s) -> uint32_t {
		return base_h[s] | (partial_row[s] ? DfaImage::kPartial : 0u) | (fm && f[s] == ROOT ? DfaImage::kRootFail : 0u);
	};
	// Failure form: the scan knows a state by the entry it arrived through (slot index T, either array; after a miss that
	// consumed the byte: the root row's slot of the byte) -- every copy of an entry carries what is needed about its
	// target: ev_miss[T] = F(target), gof[T] = the target's global base.  There are as many keys as slots, so the
	// number of states is not limited by the values a 16-bit base can take, and any entry can be in LDS.
	auto key_entry = [&](uint32_t t, uint32_t target) {
		out->ev_miss[t] = M[target];
		out->gof[t] = row_off[target + 1] != row_off[target] ? base_g[target] : 0u;
	};
	for (uint32_t s = 1; s < S && !fm; s++) { // the states' keys
		if (s < N && !reach[s]) continue;
		if (has_id[s]) {
			out->ev_miss[base_h[s]] = M[s];
			out->gof[base_h[s]] = partial_row[s] ? base_g[s] : 0u;
		} else out->ev_miss[base_g[s]] = M[s];
	}
	for (uint32_t c = 0; c < ncodes; c++) {
		out->hot[c] = c << DfaImage::kLabelShift | state_word(d1[c]);
		if (fm) key_entry(c, d1[c]);
		out->real_slots++;
		out->hot_used++;
	}
	uint64_t cold_steps = 0; // sample lookups that need the global array
	uint32_t n_partial = 0;
	for (uint32_t s = 1; s < S; s++) {
		if (row_off[s + 1] == row_off[s]) continue;
		if (!is_d1[s] && partial_row[s]) {
			n_partial++;
			uint64_t in_lds = 0;
			for (uint64_t j = row_off[s]; j < row_off[s + 1]; j++)
				if (stored(s, j)) in_lds += w_ent[j];
			cold_steps += weighted && w_row[s] > in_lds ? w_row[s] - in_lds : 0;
		}
		for (uint64_t j = row_off[s]; j < row_off[s + 1]; j++) {
			const Entry &e = ent[j];
			const uint32_t word = (uint32_t)e.code << DfaImage::kLabelShift | (e.virt ? DfaImage::kArr : 0u) | state_word(e.target);
			if (stored(s, j)) {
				const uint32_t t = base_h[s] + e.code;
				out->hot[t] = word;
				out->ev_arr[t] = e.arr;
				if (fm) key_entry(t, e.target);
				out->hot_used++;
			}
			{
				const uint32_t t = base_g[s] + e.code;
				out->slot[2 * (uint64_t)t] = word;
				out->slot[2 * (uint64_t)t + 1] = base_g[e.target];
				out->ev_arr[t] = e.arr;
				if (fm) key_entry(t, e.target);
				out->ent_key[t - hot_cap] = s << 8 | e.code;
			}
			if (e.virt) out->virtual_slots++;
			else out->real_slots++;
		}
	}
	pt.mark("image");
	out->sample_cold = sampled && !have_prior ? cold_steps : 0;
	out->max_events = max_events;
	out->valid = true;
	if (getenv("TOKSCORE_TRACE_HOST"))
		fprintf(stderr, "BuildDfa: %u slots (%u real, %u virtual entries), LDS array %u of %u used (%zu rows in use, %u chosen, %u found no place, %u partial, "
		        "%u states without a base of their own), %u codes, stride %u; sample: %.4f virtual, %.4f miss; %.6f of the weighted lookups beyond LDS%s\n",
		        out->nslots, out->real_slots, out->virtual_slots, out->hot_used, out->hot_slots, order.size(), n_sel, n_demoted, n_partial, n_cold, ncodes, stride,
		        (double)out->sample_virtual / std::max<uint64_t>(1, out->sample_steps),
		        (double)out->sample_miss / std::max<uint64_t>(1, out->sample_steps),
		        (double)cold_steps / (double)std::max<uint64_t>(1, std::accumulate(w_row.begin(), w_row.end(), (uint64_t)0)),
		        have_prior ? " (with an earlier image's counts)" : "");
	return true;
}

void DfaRun(const DfaImage &img, const uint8_t *bytes, const uint64_t *off, uint64_t D, uint64_t V, uint64_t *counts,
            uint64_t *total_tokens, uint64_t *global_reads, std::vector<uint32_t> *global_hits) {
	uint64_t n_global = 0;
	if (global_hits) global_hits->assign(img.nslots - img.hot_slots, 0);
	for (uint64_t i = 0; i < V + 3; i++) counts[i] = 0;
	*total_tokens = 0;
	std::vector<uint64_t> n_miss(img.ev_miss.size(), 0), n_arr(img.ev_arr.size(), 0);
	uint32_t S = img.start_word(), G = 0;
	uint32_t T = 0; // (failure form) the entry the state was entered through: the root row's slot of the separator at first
	const bool fm = img.fail_code != 0;
	uint64_t kind[4] = {0, 0, 0, 0}, steps = 0; // (trace) second looks: partial row miss / hit, no LDS base miss / hit
	auto label = [&](uint32_t x) { return (x >> DfaImage::kLabelShift) & DfaImage::kLabelMask; };
	for (uint64_t d = 0; d < D; d++) {
		for (uint64_t p = off[d]; p <= off[d + 1]; p++) {
			const uint32_t c = p < off[d + 1] ? img.code[ascii_lower(bytes[p])] : 0u;
			for (;;) { // (the failure form looks a byte up again in f's state; the flattened form takes every byte at once)
				steps++;
				const uint32_t b = S & DfaImage::kBaseMask;
				// one lookup in the state's row: LDS first, the global array where LDS does not hold the row completely
				auto look = [&](uint32_t code, uint32_t *id_out, uint32_t *x_out, uint32_t *g_out) -> bool {
					uint32_t id = b + code; // (always inside the LDS array)
					uint32_t x = img.hot[id];
					bool hit = label(x) == code;
					if (!hit && (S & DfaImage::kPartial)) {
						n_global++;
						id = (fm ? img.gof[T] : b == img.cold_base ? G : img.gof[b]) + code;
						x = img.slot[2 * (uint64_t)id];
						hit = label(x) == code;
						if (hit) *g_out = img.slot[2 * (uint64_t)id + 1];
						if (hit && global_hits) (*global_hits)[id - img.hot_slots]++;
						kind[(b == img.cold_base ? 2 : 0) + (hit ? 1 : 0)]++;
					}
					*id_out = id, *x_out = x;
					return hit;
				};
				uint32_t id, x, g_next = G;
				if (look(c, &id, &x, &g_next)) {
					if (x & DfaImage::kArr) n_arr[id]++;
					S = x;
					G = g_next;
					T = id;
					break;
				}
				n_miss[fm ? T : b == img.cold_base ? G : b]++;
				if (!fm || (S & DfaImage::kRootFail)) {
					S = c * img.stride | (fm ? DfaImage::kRootFail : 0u);
					T = c;
					break;
				}
				if (!look(img.fail_code, &id, &x, &g_next)) { // (cannot happen: every state whose f is not the root has the entry)
					S = c * img.stride | DfaImage::kRootFail;
					T = c;
					break;
				}
				S = x;
				G = g_next;
				T = id;
			}
		}
	}
	// (the run ends in the separator's state: nothing is pending)
	std::vector<uint32_t> ids;
	for (size_t t = 0; t < n_miss.size(); t++) {
		for (int k = 0; k < 2; k++) {
			const uint64_t n = k ? n_arr[t] : n_miss[t];
			if (!n) continue;
			ids.clear();
			img.event_ids(k ? img.ev_arr[t] : img.ev_miss[t], &ids);
			for (uint32_t id : ids) {
				counts[id] += n;
				*total_tokens += n;
			}
		}
	}
	counts[0] = counts[1] = D; // START / END once per document (SolutionFile.cpp:85,98)
	if (global_reads) *global_reads = n_global;
	if (getenv("TOKSCORE_TRACE_HOST"))
		fprintf(stderr, "DfaRun: %llu steps; second looks: partial row %llu miss %llu hit, no base of its own %llu miss %llu hit\n", (unsigned long long)steps,
		        (unsigned long long)kind[0], (unsigned long long)kind[1], (unsigned long long)kind[2], (unsigned long long)kind[3]);
}

// ---- the search automaton of a batch of strings (dfa.h) ----------------------------------------------
bool BuildSearchImage(const uint8_t *codes, const uint32_t *off, uint32_t n, uint32_t ncodes, uint32_t max_slots, SearchImage *out) {
	// (runs once per batch of moves, on the host, in front of the launches: flat arrays, no allocation per state)
	*out = SearchImage{};
	if (ncodes < 2 || ncodes > DfaImage::kMaxCodes || max_slots > DfaImage::kMaxHot || n >= 0x10000u) return false;
	const uint32_t cap = off[n] + 1; // states: the root and at most one per byte of the strings
	const uint32_t NC = ncodes;
	std::vector<uint32_t> delta((size_t)cap * NC, 0); // first the children (0 = none: the root is nobody's child), then the full transition function
	std::vector<uint32_t> parent(cap, 0), fail(cap, 0), depth(cap, 0), out_link(cap, 0), first_end(cap, 0xFFFFFFFFu), next_end(n, 0xFFFFFFFFu);
	std::vector<uint32_t> first_child(cap, 0), next_sibling(cap, 0); // a state's children, a list through next_sibling
	std::vector<uint8_t> code_of(cap, 0);
	uint32_t S = 1, max_len = 0;
	for (uint32_t k = 0; k < n; k++) {
		const uint32_t len = off[k + 1] - off[k];
		if (len < 2 || len > 0x3FFFu) continue;
		bool ok = true;
		for (uint32_t i = off[k]; i < off[k + 1]; i++)
			if (codes[i] == 0 || codes[i] >= NC) ok = false;
		if (!ok) continue; // (cannot occur in a prepared corpus)
		uint32_t v = 0;
		for (uint32_t i = off[k]; i < off[k + 1]; i++) {
			const uint8_t c = codes[i];
			uint32_t &to = delta[(size_t)v * NC + c];
			if (!to) {
				to = S;
				parent[S] = v, code_of[S] = c, depth[S] = depth[v] + 1;
				next_sibling[S] = first_child[v], first_child[v] = S;
				S++;
			}
			v = to;
		}
		next_end[k] = first_end[v]; // strings that end at v: a list through next_end
		first_end[v] = k;
		max_len = std::max(max_len, len);
	}
	// breadth first: fail links; a state's transitions are its fail state's with its own children on top (a row copied,
	// the children written over it), and so is the set of codes that lead deeper than depth 1 -- its row's entries -- as a
	// bit mask over the codes (four words: codes are bytes)
	struct Mask {
		uint64_t w[4];
	};
	std::vector<Mask> row(S, Mask{{0, 0, 0, 0}});
	std::vector<uint32_t> order;
	order.reserve(S);
	for (uint32_t u = first_child[0]; u; u = next_sibling[u]) order.push_back(u);
	for (size_t h = 0; h < order.size(); h++) {
		const uint32_t v = order[h];
		const uint32_t f = depth[v] == 1 ? 0u : delta[(size_t)fail[parent[v]] * NC + code_of[v]];
		fail[v] = f;
		out_link[v] = first_end[v] != 0xFFFFFFFFu ? v : out_link[f];
		uint32_t *dv = &delta[(size_t)v * NC];
		// (v's own children sit in dv already: kept aside while the fail state's row is copied in)
		memcpy(dv, &delta[(size_t)f * NC], NC * sizeof(uint32_t));
		row[v] = row[f];
		for (uint32_t u = first_child[v]; u; u = next_sibling[u]) {
			dv[code_of[u]] = u;
			row[v].w[code_of[u] >> 6] |= 1ull << (code_of[u] & 63);
			order.push_back(u);
		}
	}
	// rows: the depth-1 state of code c at c * stride (empty where no string starts with c; row 0, the separator's, is
	// the start state) -- the smallest stride at which their entries do not meet; the others next fit, every state
	// a base of its own (a lookup from base b at code c can only take an entry of the row that starts at b).  Slots and
	// bases as bit sets: a place is tried with four shifted ANDs, whatever the row holds.
	std::vector<uint32_t> base(S, 0);
	const size_t words = ((size_t)max_slots + NC) / 64 + 6;
	std::vector<uint64_t> occ(words, 0), taken(words, 0); // slots in use, bases in use
	auto window = [&](const std::vector<uint64_t> &bits, size_t at) { // 64 bits from bit `at` on
		const size_t i = at >> 6, sh = at & 63;
		return sh ? (bits[i] >> sh) | (bits[i + 1] << (64 - sh)) : bits[i];
	};
	auto fits = [&](uint32_t b, const Mask &m) {
		for (int q = 0; q < 4; q++)
			if (m.w[q] && (window(occ, (size_t)b + 64 * q) & m.w[q])) return false;
		return true;
	};
	auto place = [&](uint32_t b, const Mask &m) {
		for (int q = 0; q < 4; q++) {
			if (!m.w[q]) continue;
			const size_t at = (size_t)b + 64 * q, i = at >> 6, sh = at & 63;
			occ[i] |= m.w[q] << sh;
			if (sh) occ[i + 1] |= m.w[q] >> (64 - sh);
		}
		taken[b >> 6] |= 1ull << (b & 63);
	};
	uint32_t stride = 0;
	for (uint32_t st = 1; st <= NC && !stride; st++) {
		if ((size_t)(NC - 1) * st + NC > max_slots) return false;
		bool ok = true;
		for (uint32_t v = first_child[0]; v && ok; v = next_sibling[v]) {
			if (!fits(code_of[v] * st, row[v])) ok = false;
			else place(code_of[v] * st, row[v]);
		}
		if (ok) stride = st;
		std::fill(occ.begin(), occ.begin() + ((size_t)NC * st + NC) / 64 + 2, 0);
		std::fill(taken.begin(), taken.begin() + ((size_t)NC * st + NC) / 64 + 2, 0);
	}
	if (!stride) return false; // (never: at stride = ncodes the rows are disjoint)
	for (uint32_t c = 0; c < NC; c++) {
		const uint32_t v = c ? delta[c] : 0u; // (the root's row still holds its children only)
		place(c * stride, v ? row[v] : Mask{{0, 0, 0, 0}}); // (a base of its own also where no string starts with c)
		if (v) base[v] = c * stride;
	}
	uint32_t top = (NC - 1) * stride + NC;
	{
		// a row with entries is tried from the place of the last one on, never from the array's begin again (a few more
		// slots, a tenth of the time); a row without entries only needs a base nobody has
		uint32_t from = 0, next = 0;
		for (uint32_t v : order) {
			if (depth[v] == 1) continue; // (placed above)
			const bool has = (row[v].w[0] | row[v].w[1] | row[v].w[2] | row[v].w[3]) != 0;
			uint32_t b = has ? next : from;
			for (;; b++) {
				if (b + NC > max_slots) return false;
				if ((taken[b >> 6] >> (b & 63)) & 1ull) continue;
				if (!has || fits(b, row[v])) break;
			}
			base[v] = b;
			place(b, row[v]);
			while ((taken[from >> 6] >> (from & 63)) & 1ull) from++;
			if (has) next = b + 1;
			top = std::max(top, b + NC);
		}
	}
	if (top > max_slots) return false;
	out->hot.assign(top, DfaImage::kEmptyWord);
	out->out_first.assign(top, SearchImage::kNone);
	std::vector<uint32_t> first_of(S, SearchImage::kNone); // a state's records, shared by the entries that lead to it
	auto records_of = [&](uint32_t u) {
		if (first_of[u] != SearchImage::kNone) return first_of[u];
		first_of[u] = (uint32_t)out->out_list.size();
		for (uint32_t w = out_link[u]; w != 0; w = out_link[fail[w]])
			for (uint32_t k = first_end[w]; k != 0xFFFFFFFFu; k = next_end[k]) out->out_list.push_back(k | depth[w] << 16);
		out->out_list.back() |= 0x80000000u;
		return first_of[u];
	};
	uint32_t entries = 0;
	for (uint32_t v = 1; v < S; v++) {
		const uint32_t *dv = &delta[(size_t)v * NC];
		for (int q = 0; q < 4; q++) {
			for (uint64_t m = row[v].w[q]; m; m &= m - 1) {
				const uint32_t c = 64 * q + (uint32_t)__builtin_ctzll(m), u = dv[c], t = base[v] + c;
				uint32_t w = c << 16 | base[u];
				if (out_link[u] != 0) {
					w |= DfaImage::kArr;
					const uint32_t o = records_of(u), r = out->out_list[o];
					if (r & 0x80000000u) { // one string ends here: its record in place of the list's index, its length in the word
						out->out_first[t] = r | SearchImage::kInline;
						const uint32_t len = (r >> 16) & 0x3FFFu;
						if (len <= 63) w |= len << SearchImage::kLenShift;
					} else out->out_first[t] = o;
				}
				out->hot[t] = w;
				entries++;
			}
		}
	}
	out->entries = entries;
	out->hot_slots = top, out->stride = stride, out->max_len = max_len, out->states = S;
	return true;
}

void SearchRun(const SearchImage &img, const uint8_t *text, uint64_t n, std::vector<std::pair<uint32_t, uint64_t>> *found) {
	uint32_t S = 0; // (the start state: base 0)
	for (uint64_t i = 0; i < n; i++) {
		const uint32_t c = text[i], t = (S & DfaImage::kBaseMask) + c;
		const uint32_t x = t < img.hot_slots ? img.hot[t] : DfaImage::kEmptyWord;
		if (((x >> DfaImage::kLabelShift) & DfaImage::kLabelMask) == c) {
			S = x;
			if (x & DfaImage::kArr) {
				const uint32_t f = img.out_first[t];
				if (f & SearchImage::kInline) {
					found->emplace_back(f & 0xFFFFu, i + 1 - ((f >> 16) & 0x3FFFu));
				} else {
					for (uint32_t o = f;; o++) {
						const uint32_t r = img.out_list[o];
						found->emplace_back(r & 0xFFFFu, i + 1 - ((r >> 16) & 0x3FFFu));
						if (r & 0x80000000u) break;
					}
				}
			}
		} else S = c * img.stride;
	}
}

} // namespace tokscore
