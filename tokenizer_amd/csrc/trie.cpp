// trie.cpp -- see trie.h.
#include "trie.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <thread>

namespace tokscore {

namespace {
inline uint8_t ascii_lower(uint8_t c) {
	// std::tolower in the "C" locale (SolutionFile.cpp:87-88): A-Z only.
	return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c;
}
} // namespace

bool VocabTrie::Build(const uint8_t *bytes, const uint64_t *off, uint64_t V, std::string *err) {
	V_ = V;
	max_len_ = 0;
	first_child_.clear();
	n_children_.clear();
	edges_.clear();
	depth_.clear();
	head_tok_.clear();
	weight_.clear();
	next_dup_.assign(V, -1);
	tok_node_.assign(V, 0);
	enabled_.assign(V, 1);
	node_slot_.clear();
	sampled_ = false;

	for (uint64_t i = 0; i < V; i++) {
		if (off[i + 1] < off[i]) {
			if (err) *err = "vocabulary offsets are not monotone";
			return false;
		}
		// max_len_ covers every entry, duplicates included (SolutionFile.cpp:55,68)
		max_len_ = std::max<uint64_t>(max_len_, off[i + 1] - off[i]);
	}
	if (max_len_ == 0) {
		if (err) *err = "vocabulary has no non-empty token (the reference would never advance)";
		return false;
	}

	// Sort token indices by (string, index): equal strings end up adjacent in
	// ascending index order, so the first of a run is the id the reference keeps
	// (unordered_map::insert fails on duplicates, SolutionFile.cpp:56).
	std::vector<uint32_t> order(V);
	std::iota(order.begin(), order.end(), 0u);
	auto len_of = [&](uint32_t i) { return off[i + 1] - off[i]; };
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
		const uint64_t la = len_of(a), lb = len_of(b);
		const int c = std::memcmp(bytes + off[a], bytes + off[b], (size_t)std::min(la, lb));
		if (c != 0) return c < 0;
		if (la != lb) return la < lb;
		return a < b;
	});

	// Breadth-first construction over sorted ranges: the children of one node are
	// created back to back, so every node's edges are contiguous and byte-sorted.
	struct Range {
		uint64_t lo, hi;
	};
	std::vector<Range> ranges;
	auto new_node = [&](uint64_t lo, uint64_t hi, uint32_t depth) {
		first_child_.push_back(0);
		n_children_.push_back(0);
		depth_.push_back(depth);
		head_tok_.push_back(-1);
		ranges.push_back({lo, hi});
		return (uint32_t)(first_child_.size() - 1);
	};
	new_node(0, V, 0);
	for (uint32_t node = 0; node < first_child_.size(); node++) {
		uint64_t lo = ranges[node].lo;
		const uint64_t hi = ranges[node].hi;
		const uint32_t d = depth_[node];
		// tokens that end exactly here (length == depth) come first in the range
		int64_t prev = -1;
		while (lo < hi && len_of(order[lo]) == d) {
			const uint32_t t = order[lo];
			tok_node_[t] = node;
			if (d > 0) { // the empty string can never match (Tokenize tries len >= 1 only)
				if (prev < 0) head_tok_[node] = t;
				else next_dup_[prev] = t;
				prev = t;
			}
			lo++;
		}
		first_child_[node] = (uint32_t)edges_.size();
		while (lo < hi) {
			const uint8_t c = bytes[off[order[lo]] + d];
			uint64_t e = lo + 1;
			while (e < hi && bytes[off[order[e]] + d] == c) e++;
			edges_.push_back({c, 0});
			n_children_[node]++;
			lo = e;
		}
		// create the children after the edge list is complete (keeps edges contiguous)
		lo = ranges[node].lo;
		while (lo < hi && len_of(order[lo]) == d) lo++;
		for (uint32_t k = 0; k < n_children_[node]; k++) {
			const uint8_t c = edges_[first_child_[node] + k].byte;
			uint64_t e = lo + 1;
			while (e < hi && bytes[off[order[e]] + d] == c) e++;
			edges_[first_child_[node] + k].child = new_node(lo, e, d + 1);
			lo = e;
		}
	}
	weight_.assign(first_child_.size(), 0);
	root_child_.assign(256, 0);
	for (uint32_t k = 0; k < n_children_[0]; k++) root_child_[edges_[first_child_[0] + k].byte] = edges_[first_child_[0] + k].child;
	return true;
}

uint32_t VocabTrie::Child(uint32_t node, uint8_t c) const {
	if (node == 0) return root_child_[c];
	const Edge *e = edges_.data() + first_child_[node];
	uint32_t lo = 0, hi = n_children_[node];
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (e[mid].byte < c) lo = mid + 1;
		else hi = mid;
	}
	return (lo < n_children_[node] && e[lo].byte == c) ? e[lo].child : 0;
}

uint32_t VocabTrie::ActiveId(uint32_t node) const {
	for (int64_t t = head_tok_[node]; t >= 0; t = next_dup_[t])
		if (enabled_[t]) return (uint32_t)t + 2; // token i -> id i+2 (SolutionFile.cpp:49-57)
	return 0;
}

void VocabTrie::AddHotnessSample(const uint8_t *sample, uint64_t n) {
	// (layout only: the sample in up to eight pieces on threads of their own, each greedy segmentation starting at its
	// piece's first byte and counting into an array of its own -- 2 MB on one thread was 35 ms of the first toggle)
	auto run = [&](uint64_t begin, uint64_t end, std::vector<uint32_t> *w) {
		uint64_t pos = begin;
		while (pos < end) {
			uint32_t node = 0;
			uint64_t len = 0;
			for (uint64_t d = 0; pos + d < n; d++) {
				if (n_children_[node] == 0) break;
				(*w)[node]++;
				const uint32_t ch = Child(node, ascii_lower(sample[pos + d]));
				if (!ch) break;
				node = ch;
				if (ActiveId(node)) len = d + 1;
			}
			pos += len ? len : 1;
		}
	};
	const unsigned pieces = n >= (1u << 20) ? 8u : n >= (1u << 18) ? 4u : 1u;
	std::vector<std::vector<uint32_t>> w(pieces, std::vector<uint32_t>(weight_.size(), 0));
	{
		std::vector<std::thread> th;
		for (unsigned t = 1; t < pieces; t++) th.emplace_back(run, n * t / pieces, n * (t + 1) / pieces, &w[t]);
		run(0, n / pieces, &w[0]);
		for (auto &x : th) x.join();
	}
	for (const auto &v : w)
		for (size_t i = 0; i < weight_.size(); i++) weight_[i] += v[i];
	sampled_ = true;
}

bool VocabTrie::Pack(PackedTrie *out, std::string *err) {
	const uint32_t N = (uint32_t)first_child_.size();
	// Parents ordered hot-first; unsampled tries fall back to breadth-first order
	// (node numbering is already breadth-first), i.e. "top levels first".
	std::vector<uint32_t> parents;
	parents.reserve(N);
	for (uint32_t n = 1; n < N; n++)
		if (n_children_[n]) parents.push_back(n);
	if (sampled_) {
		std::stable_sort(parents.begin(), parents.end(), [&](uint32_t a, uint32_t b) { return weight_[a] > weight_[b]; });
	}
	parents.insert(parents.begin(), 0u); // the root always goes first

	std::vector<uint8_t> used(1024, 0);      // slot occupied
	std::vector<uint8_t> base_used(1024, 0); // value already serves as some node's base
	// next_free[s] = a slot >= s that is free, or leads to one (slots only ever go from free to
	// used, so the links are compressed as they are followed): the search below visits free
	// slots only, however dense the array has become
	std::vector<uint32_t> next_free(1024);
	for (uint32_t i = 0; i < 1024; i++) next_free[i] = i;
	auto ensure = [&](uint64_t n) {
		if (used.size() < n) {
			const uint64_t old = used.size();
			const uint64_t m = std::max<uint64_t>(n, old * 2);
			used.resize(m, 0);
			base_used.resize(m, 0);
			next_free.resize(m);
			for (uint64_t i = old; i < m; i++) next_free[i] = (uint32_t)i;
		}
	};
	auto find_free = [&](uint64_t s) -> uint64_t {
		ensure(s + 1);
		uint64_t r = s;
		while (next_free[r] != r) {
			r = next_free[r];
			ensure(r + 1);
		}
		while (next_free[s] != r) { // compress
			const uint64_t t = next_free[s];
			next_free[s] = (uint32_t)r;
			s = t;
		}
		return r;
	};
	// A deep vocabulary (long tokens over a large alphabet) is mostly single-child nodes, and a
	// hole whose 256 possible bases are all taken can never be filled: after this many free
	// slots tried, the node takes the first fitting base above every base handed out so far.
	constexpr uint32_t kMaxTries = 256;
	node_slot_.assign(N, 0);
	std::vector<uint32_t> node_base(N, 0);
	used[0] = 1; // root
	next_free[0] = 1;
	uint64_t max_used = 0, max_base = 0;
	for (uint32_t node : parents) {
		const Edge *e = edges_.data() + first_child_[node];
		const uint32_t nc = n_children_[node];
		const uint32_t c0 = e[0].byte;
		// smallest unused base b >= 1 with every slot b + c free: candidates are the bases that
		// put the first child on a free slot
		uint64_t s = find_free(std::max<uint64_t>(1, (uint64_t)c0 + 1));
		uint64_t b = 0;
		for (uint32_t tries = 0;; tries++) {
			b = s - c0;
			ensure(b + 257);
			bool ok = !base_used[b];
			for (uint32_t k = 0; ok && k < nc; k++) ok = !used[b + e[k].byte];
			if (ok) break;
			if (tries >= kMaxTries) {
				// no base above max_base is taken: the first one whose slots are free
				for (b = max_base + 1;; b++) {
					ensure(b + 257);
					bool fits = true;
					for (uint32_t k = 0; fits && k < nc; k++) fits = !used[b + e[k].byte];
					if (fits) break;
				}
				break;
			}
			s = find_free(s + 1);
		}
		base_used[b] = 1;
		max_base = std::max(max_base, b);
		node_base[node] = (uint32_t)b;
		for (uint32_t k = 0; k < nc; k++) {
			const uint64_t slot = b + e[k].byte;
			used[slot] = 1;
			ensure(slot + 2);
			next_free[slot] = (uint32_t)(slot + 1);
			node_slot_[e[k].child] = (uint32_t)slot;
			max_used = std::max<uint64_t>(max_used, slot);
		}
	}
	const uint64_t nslots = max_used + 1 + 256; // base + 255 is always a readable slot
	if (nslots > PackedTrie::kMaxSlots) {
		if (err) *err = "vocabulary too large for the packed trie (more than 2^23 slots)";
		return false;
	}
	out->nodes = N;
	out->max_len = max_len_;
	out->unknown_id = (uint32_t)(V_ + 2);
	out->slots.assign(nslots, 0); // empty: valid = 0, base = 0, id = 0
	for (uint32_t node = 0; node < N; node++) {
		uint64_t entry = (uint64_t)node_base[node] << PackedTrie::kBaseShift; // root: valid = 0
		entry |= (uint64_t)ActiveId(node) << 32;
		out->slots[node_slot_[node]] = entry; // slot 0 for the root; unique per node otherwise
	}
	// label + valid: walk the edges
	for (uint32_t node = 0; node < N; node++) {
		const Edge *e = edges_.data() + first_child_[node];
		for (uint32_t k = 0; k < n_children_[node]; k++)
			out->slots[node_slot_[e[k].child]] |= (uint64_t)(PackedTrie::kValid | e[k].byte);
	}
	return true;
}

// lt / slot_id of the slot of `node` (a child of the root when depth 1), reached by `code`
void VocabTrie::StreamSlotValues(uint32_t node, uint32_t code, uint8_t *lt, uint32_t *slot_id) const {
	const uint32_t id = ActiveId(node);
	if (depth_[node] == 1) { // root row: always a token end; no token of its own = UNKNOWN
		*lt = (uint8_t)(code | 0x80u);
		*slot_id = id ? id : (uint32_t)(V_ + 2);
	} else {
		*lt = (uint8_t)(code | (id ? 0x80u : 0u));
		*slot_id = id ? id : StreamImage::kSkipSlot;
	}
}

bool VocabTrie::PackStream(StreamImage *out) {
	out->valid = false;
	const uint32_t N = (uint32_t)first_child_.size();
	// codes: 0 = separator (byte 0); then the bytes that label an edge and can start a UTF-8 sequence,
	// one code for every other such byte; then (from first_cont_code on) the continuation bytes
	// 0x80-0xBF that label an edge and one code for the other continuation bytes -- so that
	// "this byte starts a code point" is `code < first_cont_code` (the counting kernel needs it)
	bool is_label[256] = {};
	for (const Edge &e : edges_) is_label[e.byte] = true;
	is_label[0] = false; // edges labelled NUL are left out
	uint32_t k = 0;
	for (uint32_t b = 1; b < 256; b++)
		if (is_label[b]) k++;
	if (k + 3 > StreamImage::kCodes) return false;
	uint32_t other = 0; // the largest code: every continuation byte without an edge
	{
		auto is_cont = [](uint32_t b) { return (b & 0xC0u) == 0x80u; };
		uint32_t next = 1;
		out->code[0] = 0;
		for (uint32_t b = 1; b < 256; b++)
			if (is_label[b] && !is_cont(b)) out->code[b] = (uint8_t)next++;
		const uint32_t other_lead = next++;
		out->first_cont_code = next;
		for (uint32_t b = 0x80; b < 0xC0; b++)
			if (is_label[b]) out->code[b] = (uint8_t)next++;
		other = next;
		for (uint32_t b = 1; b < 256; b++)
			if (!is_label[b]) out->code[b] = (uint8_t)(is_cont(b) ? other : other_lead);
	}
	// Rows widest first (packs tightest).  The image is used only when ALL of it fits in LDS, so
	// unlike Pack there is nothing to gain from a hot-first order.  Nodes below an edge labelled
	// NUL are unreachable and get no row (node numbers are breadth-first: parents come first).
	std::vector<uint8_t> reach(N, 0);
	reach[0] = 1;
	for (uint32_t n = 0; n < N; n++) {
		if (!reach[n]) continue;
		const Edge *e = edges_.data() + first_child_[n];
		for (uint32_t i = 0; i < n_children_[n]; i++)
			if (e[i].byte != 0) reach[e[i].child] = 1;
	}
	std::vector<uint32_t> parents;
	for (uint32_t n = 1; n < N; n++)
		if (n_children_[n] && reach[n]) parents.push_back(n);
	std::stable_sort(parents.begin(), parents.end(), [&](uint32_t a, uint32_t b) { return n_children_[a] > n_children_[b]; });
	// first-fit placement over a free-slot chain (slots only ever go from free to used)
	std::vector<uint8_t> used(4096, 0), base_used(4096, 0);
	std::vector<uint32_t> next_free(4096);
	for (uint32_t i = 0; i < 4096; i++) next_free[i] = i;
	auto ensure = [&](uint64_t n) {
		if (used.size() < n) {
			const uint64_t old = used.size(), m = std::max<uint64_t>(n, old * 2);
			used.resize(m, 0);
			base_used.resize(m, 0);
			next_free.resize(m);
			for (uint64_t i = old; i < m; i++) next_free[i] = (uint32_t)i;
		}
	};
	auto find_free = [&](uint64_t s) -> uint64_t {
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
	};
	uint64_t max_used = 0, max_base = 0;
	auto place = [&](const uint32_t *codes, uint32_t nc) -> uint64_t {
		const uint32_t c0 = codes[0];
		uint64_t s = find_free(std::max<uint64_t>(1, (uint64_t)c0 + 1));
		uint64_t b = 0;
		for (uint32_t tries = 0;; tries++) {
			b = s - c0;
			ensure(b + StreamImage::kCodes + 1);
			bool ok = !base_used[b];
			for (uint32_t i = 0; ok && i < nc; i++) ok = !used[b + codes[i]];
			if (ok) break;
			if (tries >= 4096) { // (deep vocabularies: see Pack)
				for (b = max_base + 1;; b++) {
					ensure(b + StreamImage::kCodes + 1);
					bool fits = true;
					for (uint32_t i = 0; fits && i < nc; i++) fits = !used[b + codes[i]];
					if (fits) break;
				}
				break;
			}
			s = find_free(s + 1);
		}
		base_used[b] = 1;
		max_base = std::max(max_base, b);
		for (uint32_t i = 0; i < nc; i++) {
			const uint64_t slot = b + codes[i];
			used[slot] = 1;
			ensure(slot + 2);
			next_free[slot] = (uint32_t)(slot + 1);
			max_used = std::max(max_used, slot);
		}
		return b;
	};
	std::vector<uint32_t> node_base(N, 0);
	node_sslot_.assign(N, 0);
	{
		// The root row is dense and sits at base 0: "no children" and "back at the root" are then
		// the same value, and a walker that has finished a token is at the root without being reset.
		// Slot 0 is the separator's (no node of the trie has it: node_sslot_ == 0 means "not in the image").
		ensure(other + 2);
		for (uint32_t c = 0; c <= other; c++) {
			used[c] = 1;
			next_free[c] = c + 1;
		}
		base_used[0] = 1;
		max_used = other;
		node_base[0] = 0;
		for (uint32_t i = 0; i < n_children_[0]; i++) {
			const Edge &e = edges_[first_child_[0] + i];
			if (e.byte != 0) node_sslot_[e.child] = node_base[0] + out->code[e.byte];
		}
	}
	std::vector<uint32_t> codes;
	for (uint32_t node : parents) {
		const Edge *e = edges_.data() + first_child_[node];
		codes.clear();
		for (uint32_t i = 0; i < n_children_[node]; i++)
			if (e[i].byte != 0) codes.push_back(out->code[e[i].byte]);
		if (codes.empty()) continue;
		std::sort(codes.begin(), codes.end()); // (place() starts from the smallest)
		const uint32_t b = (uint32_t)place(codes.data(), (uint32_t)codes.size());
		node_base[node] = b;
		for (uint32_t i = 0; i < n_children_[node]; i++)
			if (e[i].byte != 0) node_sslot_[e[i].child] = b + out->code[e[i].byte];
		if (max_used + StreamImage::kCodes > StreamImage::kMaxSlots) return false;
	}
	const uint64_t nslots = std::max(max_used + 1, max_base + StreamImage::kCodes); // base + any code is a readable slot
	if (nslots > StreamImage::kMaxSlots) return false;
	out->lt.assign(nslots, 0);
	out->base.assign(nslots, 0);
	out->slot_id.assign(nslots, StreamImage::kSkipSlot);
	out->root_base = node_base[0];
	for (uint32_t c = 0; c <= other; c++) { // root slots without a child of their own
		const uint32_t t = node_base[0] + c;
		out->lt[t] = (uint8_t)(c | 0x80u);
		out->slot_id[t] = c == 0 ? StreamImage::kSkipSlot : (uint32_t)(V_ + 2);
	}
	for (uint32_t node = 1; node < N; node++) {
		const uint32_t t = node_sslot_[node];
		if (!t) continue;
		StreamSlotValues(node, out->lt[t] & 0x7Fu, &out->lt[t], &out->slot_id[t]); // (the code is filled in below)
		out->base[t] = (uint16_t)node_base[node];
	}
	// codes of the real nodes: walk the edges
	for (uint32_t node = 0; node < N; node++) {
		if (node != 0 && node_sslot_[node] == 0) continue;
		const Edge *e = edges_.data() + first_child_[node];
		for (uint32_t i = 0; i < n_children_[node]; i++) {
			if (e[i].byte == 0) continue;
			const uint32_t t = node_sslot_[e[i].child];
			out->lt[t] = (uint8_t)((out->lt[t] & 0x80u) | out->code[e[i].byte]);
		}
	}
	out->valid = true;
	return true;
}

int64_t VocabTrie::PatchStream(uint64_t token, StreamImage *img) const {
	if (token >= V_ || !img->valid || node_sslot_.empty()) return -1;
	const uint32_t node = tok_node_[token];
	if (node == 0) return -1;
	const uint32_t t = node_sslot_[node];
	if (!t) return -1;
	uint8_t lt;
	uint32_t id;
	StreamSlotValues(node, img->lt[t] & 0x7Fu, &lt, &id);
	if (lt == img->lt[t] && id == img->slot_id[t]) return -1;
	img->lt[t] = lt;
	img->slot_id[t] = id;
	return t;
}

int64_t VocabTrie::PeekStream(uint64_t token, bool enable, const StreamImage &img, bool *flips) {
	*flips = false;
	if (token >= V_ || !img.valid || node_sslot_.empty()) return -1;
	const uint32_t node = tok_node_[token];
	if (node == 0) return -1;
	const uint32_t t = node_sslot_[node];
	if (!t) return -1;
	const uint8_t was = enabled_[token];
	enabled_[token] = enable ? 1 : 0;
	uint8_t lt;
	uint32_t id;
	StreamSlotValues(node, img.lt[t] & 0x7Fu, &lt, &id);
	enabled_[token] = was;
	*flips = ((lt ^ img.lt[t]) & 0x80u) != 0;
	return t;
}

void VocabTrie::AccumulateEnds(const uint64_t *slot_ends, uint32_t max_cp, uint64_t *counts) const {
	(void)max_cp; // (the walks stop there by themselves: deeper nodes have no ends)
	for (uint64_t i = 0; i < V_ + 3; i++) counts[i] = 0;
	const uint32_t N = (uint32_t)first_child_.size();
	if (node_sslot_.size() != N) return;
	// UTF-8 state of every node's string: bytes still owed to the current code point, -1 = not UTF-8
	std::vector<int8_t> owed(N, 0);
	std::vector<uint64_t> visits(N, 0);
	for (uint32_t n = 0; n < N; n++) { // breadth-first numbering: a parent comes before its children
		const Edge *e = edges_.data() + first_child_[n];
		for (uint32_t i = 0; i < n_children_[n]; i++) {
			const uint32_t b = e[i].byte;
			int8_t st = owed[n];
			if (st < 0) st = -1;
			else if (st > 0) st = (b & 0xC0u) == 0x80u ? (int8_t)(st - 1) : (int8_t)-1;
			else st = b < 0x80u ? 0 : b >= 0xC2u && b < 0xE0u ? 1 : b >= 0xE0u && b < 0xF0u ? 2 : b >= 0xF0u && b < 0xF5u ? 3 : -1;
			owed[e[i].child] = st;
		}
	}
	for (uint32_t n = N; n-- > 0;) { // children before parents
		if (n != 0 && node_sslot_[n]) visits[n] += slot_ends[node_sslot_[n]];
		const Edge *e = edges_.data() + first_child_[n];
		for (uint32_t i = 0; i < n_children_[n]; i++) visits[n] += visits[e[i].child];
		if (n == 0 || owed[n] != 0) continue;
		const uint32_t id = ActiveId(n);
		if (id) counts[id] += visits[n];
	}
}

uint32_t VocabTrie::StringRows(uint64_t nslots, std::vector<int32_t> *slot_row, std::vector<int32_t> *tok_row) const {
	slot_row->assign(nslots, -1);
	tok_row->assign(V_, -1);
	uint32_t rows = 0;
	if (node_slot_.empty()) return 0;
	for (uint64_t i = 0; i < V_; i++) {
		const uint32_t node = tok_node_[i];
		if (node == 0) continue; // empty string
		int32_t &r = (*slot_row)[node_slot_[node]];
		if (r < 0) r = (int32_t)rows++;
		(*tok_row)[i] = r;
	}
	return rows;
}

int64_t VocabTrie::Toggle(uint64_t token, bool enable, PackedTrie *packed, uint64_t *new_entry) {
	if (token >= V_) return -1;
	enabled_[token] = enable ? 1 : 0;
	const uint32_t node = tok_node_[token];
	if (node == 0 || node_slot_.empty()) return -1; // empty string, or not packed yet
	const uint32_t slot = node_slot_[node];
	const uint64_t e = (packed->slots[slot] & 0xFFFFFFFFull) | ((uint64_t)ActiveId(node) << 32);
	if (e == packed->slots[slot]) return -1;
	packed->slots[slot] = e;
	*new_entry = e;
	return slot;
}

void VocabTrie::Longest(const PackedTrie &p, const uint8_t *text, uint64_t n, uint64_t *len, uint64_t *id) {
	*len = 0;
	*id = 0;
	uint32_t base = PackedTrie::base(p.slots[0]);
	for (uint64_t d = 0; d < n && base; d++) {
		const uint32_t c = ascii_lower(text[d]);
		const uint64_t e = p.slots[base + c];
		if (!PackedTrie::matches(e, c)) break;
		base = PackedTrie::base(e);
		if (PackedTrie::id(e)) {
			*len = d + 1;
			*id = PackedTrie::id(e);
		}
	}
}

} // namespace tokscore
