// dfa.h -- the vocabulary as a "one transition per corpus byte" automaton (round 3).
//
// SolutionFile::Tokenize (/root/reference/src/files/SolutionFile.cpp:89-97) probes
// len = min(max_len_, remaining) .. 1 at every token start: a trie walk that goes as deep as the
// bytes allow, falls back to the longest token end it passed and starts again behind it.  The
// bytes between that token end and the point where the walk fell off are looked at twice (1.15
// lookups per byte on the bench vocabulary, 2.7 - 3.4 on the vocabulary the reference's own annealer
// writes, tests/golden/wiki_annealed, whose prefixes are mostly not tokens).  This image removes the
// second look: a state is a trie node s ("the bytes since the last decided token end spell s"), and
// for every state the builder works out, once, what falling off at s means --
//   F(s)  the tokens that are decided at that moment: the longest token end on s's path, then the
//         greedy tokens of what is left for as long as they are decided too, and
//   f(s)  the node spelled by the undecided rest (always a trie path),
// so that the transition on byte c is child(s, c) if it exists and otherwise the transition of f(s),
// with F(s) emitted -- the construction of LinMaxMatch (Song et al., "Fast WordPiece Tokenization",
// EMNLP 2021) applied to the reference's rule set: bytes, not code points; every byte is a token of
// its own or UNKNOWN (SolutionFile.cpp:92-94), so the chain of f always ends at the root.  The scan
// kernel (kernels_dfa.hip) wants exactly ONE slot read per byte, so the chain is flattened:
//   * a miss in s's row means "all of f's chain emitted, next state = the depth-1 state of c", and
//     the depth-1 states' rows sit at base c * stride, so that state needs no read at all;
//   * a byte c for which some state on f's chain does have a child gets a VIRTUAL slot in s's own
//     row (flagged kArr) that leads to that child's row and stands for the tokens emitted on the way.
// Token counts come from events, not from ids: the kernel logs the key of the state it left by a miss
// (the state's LDS base, or its global base for the states that share cold_base: ev_miss[key] = every
// token of the chain) and the index of the virtual entry it arrived on (ev_arr[entry]); the histogram
// over keys is turned into counts[id] with these tables at the end of an evaluation.
// What LDS holds is chosen entry by entry (a sample of the corpus run through the automaton, and after
// a plan's first evaluation the scan's own counters, say which entries are used): see DfaImage below.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace tokscore {

struct DfaImage {
	// slot word: kArr : 1 | partial : 1 | 0 : 6 | label (code) : 8 | base : 16.  The state a word stands
	// for looks its next byte up at slot base + code OF THE LDS ARRAY; `partial` says that LDS does not
	// hold that state's row completely, so a miss there proves nothing.  Empty slots carry the all-ones
	// label, which no code equals.  The label is a byte of its own (the scan compares it with the corpus
	// byte in one instruction).  The scan's loop gives the state a lane takes after a miss the top byte of the
	// state left, shifted left by one: the sign bit then says "that state was partial" (kArr drops out, and so
	// does the mark itself one step later), so that one sign test finds both the lanes whose miss LDS may not
	// decide (sign and miss) and the arrivals on kArr entries (sign and hit), and nothing else.
	static constexpr uint32_t kLabelShift = 16;
	static constexpr uint32_t kLabelMask = 0xFFu;
	static constexpr uint32_t kArr = 0x80000000u;
	static constexpr uint32_t kPartial = 0x40000000u;
	// (failure form only, see fail_code) the state's failure target is the root: a miss there consumes the byte
	// and leads to the byte's depth-1 state, as every miss does in the flattened form
	static constexpr uint32_t kRootFail = 0x20000000u;
	static constexpr uint32_t kStateMask = 0xFFFFu | kPartial | kRootFail; // what of a word says which state it leads to
	static constexpr uint32_t kBaseMask = 0xFFFFu;
	static constexpr uint32_t kEmptyWord = 0xFF0000u;
	static constexpr uint32_t kMaxCodes = 255;      // codes 0 .. ncodes - 1 in the label's byte, all ones = empty
	static constexpr uint32_t kMaxHot = 65536;      // (the base's 16 bits)
	static constexpr uint64_t kMaxSlots = 1u << 23;
	// events: kEvNone = nothing to count, id < 2^31 = one token id, kEvList | o = pool[o] ids at pool[o + 1 ..]
	static constexpr uint32_t kEvNone = 0xFFFFFFFFu;
	static constexpr uint32_t kEvList = 0x80000000u;

	// Two arrays over one index space.  `slot` [hot_slots, nslots) holds the whole automaton, read from
	// global memory (L2): every state's row, complete, at the state's GLOBAL base.  `hot` [0, hot_slots)
	// is what the scan keeps in LDS: the root row, the depth-1 rows, and an overlay for the rows that are
	// used -- a row is there COMPLETE, or PARTIAL: only the entries in use.
	//   A state is known by its LDS base, a value no other state has (a state without a row, or whose row
	// LDS does not hold at all, still gets a base of its own while values last: a base needs no slots,
	// only that no other row starts there, so every lookup at base + code misses).  States beyond that
	// share `cold_base` and are told apart by their global base, which the scan carries beside the state
	// word while it is in one of them; they are only ever entered from the global array, whose entries
	// hold the target's global base in a second word.  A word's `partial` bit is set for partial rows and
	// for cold_base: after a miss in LDS the lookup is repeated at global base + code, the global base
	// being gof[LDS base], or the carried one under cold_base.
	//   Events are counted per key: leaving a state by a miss under the state's key (its LDS base, or its
	// global base under cold_base), arriving on a virtual entry under the entry's index (either array).
	std::vector<uint32_t> hot;
	std::vector<uint32_t> slot;    // [2 * nslots]: slot[2 i] the word of entry i, slot[2 i + 1] its target's global base (i >= hot_slots)
	std::vector<uint32_t> gof;     // [hot_slots] by LDS base: the state's global base (0: its row is complete in LDS, or it has none)
	std::vector<uint32_t> ev_miss; // [nslots] by state key: what leaving the state by a miss emits
	std::vector<uint32_t> ev_arr;  // [nslots] by entry index: what arriving on a virtual entry emits
	std::vector<uint32_t> ent_key; // [nslots - hot_slots] host only: global entry -> trie node of its row << 8 | code (kEvNone: no entry)
	uint32_t hot_slots = 0;        // = DfaOptions::lds_slots
	uint32_t cold_base = 0;
	uint32_t hot_used = 0;         // entries in `hot`
	uint64_t sample_cold = 0;      // sample lookups that needed `slot`
	std::vector<uint32_t> pool;
	uint8_t code[256] = {};
	uint32_t ncodes = 0;  // code 0 = document separator, ncodes - 1 = every byte that labels no edge
	// The FAILURE FORM (DfaOptions::fail_mode; 0 here = the flattened form above).  Rows hold the trie's own edges only,
	// and f is followed explicitly: a miss in s emits F(s) alone (ev_miss[key] = F(s)) and leads to f(s) WITHOUT
	// consuming the byte -- f(s)'s word is an entry of s's row under the pseudo-code fail_code = ncodes, which no
	// corpus byte has -- unless f(s) is the root (kRootFail in every word that leads to s), where the byte is
	// consumed and the next state is its depth-1 state.  No virtual entries, no arrivals: 1 + (trie nodes) * 2 slots
	// at most whatever the vocabulary, and at most two lookups per byte on average (a failure pops what hits pushed)
	// instead of one -- the form for vocabularies whose prefixes are mostly not tokens, whose flattened rows run
	// into millions of virtual entries (kernels_dfa.hip: dfa_fscan_kernel).
	uint32_t fail_code = 0;
	uint32_t stride = 0;  // the row of the depth-1 state of code c starts at slot c * stride (row 0 = the root's)
	uint32_t nslots = 0;
	uint32_t real_slots = 0, virtual_slots = 0;
	uint32_t max_events = 0; // longest event list
	// what the hotness sample saw (layout only, never results)
	uint64_t sample_steps = 0, sample_virtual = 0, sample_miss = 0;
	// ... kept for the next image of the same vocabulary state (DfaOptions::reuse): uses per builder entry, lookups per state
	std::vector<uint32_t> sample_w_ent;
	std::vector<uint64_t> sample_w_row;
	uint64_t vocab_tag = 0;
	bool valid = false;

	// the state after a document separator: slot 0 (the root row is the separator's row)
	uint32_t start_word() const { return hot.empty() ? 0u : hot[0] & kStateMask; }
	// share of the sample's lookups that `hot` alone answered
	double hot_fraction() const;
	// appends an event's ids to out (with repetitions)
	void event_ids(uint32_t ev, std::vector<uint32_t> *out) const;
};

struct DfaOptions {
	uint32_t lds_slots = 40000;    // slots of the LDS array
	uint64_t max_slots = 6u << 20; // give up above this many slots (virtual slots of deep vocabularies run into tens of millions)
	const uint8_t *sample = nullptr; // raw corpus bytes for the choice of what LDS holds (lower-cased here)
	uint64_t sample_bytes = 0;
	uint64_t corpus_bytes = 0;     // what the sample stands for (scales its counts against `prior`)
	// uses of entries that an earlier image counted on the whole corpus (the scan's counters of the global
	// array, through DfaImage::ent_key): keys ascending, or null.  Keys name trie nodes, which toggles do
	// not renumber, so the counts stay useful for the images built after toggles.
	const uint32_t *prior_keys = nullptr, *prior_counts = nullptr;
	size_t prior_n = 0;
	// the image built last, when the vocabulary has not changed since (same vocab_tag, set by the caller): its sample
	// counts are taken over instead of running the sample again (40 ns per sample byte on the host)
	const DfaImage *reuse = nullptr;
	uint64_t vocab_tag = 0;
	bool fail_mode = false;        // build the failure form (DfaImage::fail_code)
	// flattened form only, 0 = no limit: give up as soon as the rows hold more than this many virtual entries per trie
	// node (the caller would turn such an image away anyway: what the reference's annealer writes has 23)
	uint32_t max_virtual_per_node = 0;
};

// ---- a batch of strings searched in one pass (tokscore_score_toggles) ------------------------------
// The same LDS array and the same step (kernels_dfa.hip: dfa_search_kernel runs the scan's loop on it), another
// automaton: Aho-Corasick over the batch's strings in the corpus' code space.  State = the longest suffix of the
// text read that is a prefix of some string; the transition on c goes to the deepest such state for the text
// plus c.  Flattened as the scoring image is: the states of depth 1 have their rows at c * stride, a miss means
// "the depth-1 state of c" (its row is empty when no string starts with c: the root's behaviour), and every
// transition to a state of depth >= 2 is an entry of its own in the row of the state it leaves -- a few thousand
// for 64 strings, all of them in LDS, no partial rows, no global array.  An entry whose target has a string of the
// batch as a suffix carries kArr: the scan then logs (entry, position of the byte), and out_first / out_list say
// which strings end there.  Strings of one code cannot be found this way and are left out (found[k] = 0): a
// one-byte string is never toggled as a token end (the root row is all token ends).
struct SearchImage {
	std::vector<uint32_t> hot;       // [hot_slots] DfaImage's word layout; the start state is word 0 (base 0, an empty row)
	std::vector<uint32_t> out_first; // [hot_slots] by entry: kNone where nothing ends; kInline | the record itself where one string ends; else the index of the first record in out_list
	std::vector<uint32_t> out_list;  // string k | bytes << 16 (14 bits) | 0x80000000 on the last of an entry's records
	uint32_t hot_slots = 0, stride = 0, max_len = 0, states = 0, entries = 0;
	static constexpr uint32_t kNone = 0xFFFFFFFFu;
	static constexpr uint32_t kInline = 0x40000000u;
	// bits 24 .. 29 of an entry's word (unused by the step): the bytes of the one string that ends there, 0 = look it up
	// (several strings, or more than 63 bytes) -- the scan passes them on with its record, so that whoever reads the record
	// can tell where the string starts before any table is read
	static constexpr uint32_t kLenShift = 24;
};
// strings: codes[off[k] .. off[k + 1]) for k < n, no code 0 and none >= ncodes inside.  false: the automaton does not fit
// max_slots (the caller searches another way).
bool BuildSearchImage(const uint8_t *codes, const uint32_t *off, uint32_t n, uint32_t ncodes, uint32_t max_slots, SearchImage *out);
// CPU check of an image: every (string, position of its first byte) it reports on `text` (codes), as the kernels do
void SearchRun(const SearchImage &img, const uint8_t *text, uint64_t n, std::vector<std::pair<uint32_t, uint64_t>> *found);

// Host-side check of an image (CPU tests of the builder; never used by a scoring entry point): runs
// the automaton over documents exactly as the kernel does -- codes of the lower-cased bytes, a
// separator after every document -- and adds up counts[V + 3] and total_tokens from the events.
// *global_reads (optional): lookups the run repeated in the global array.
// *global_hits (optional): per entry of the global array (index - hot_slots), the hits of those lookups --
// what the scan's counters hold after an evaluation.
void DfaRun(const DfaImage &img, const uint8_t *bytes, const uint64_t *off, uint64_t D, uint64_t V, uint64_t *counts,
            uint64_t *total_tokens, uint64_t *global_reads = nullptr, std::vector<uint32_t> *global_hits = nullptr);

} // namespace tokscore
