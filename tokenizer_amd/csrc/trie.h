// trie.h -- host-side vocabulary trie and its packed double-array image.
//
// Device counterpart of SolutionFile::tokens_/GetId
// (/root/reference/src/files/SolutionFile.h:10, SolutionFile.cpp:74-77): instead of an
// unordered_map probed longest-first, the vocabulary becomes a byte trie whose walk
// yields the longest match directly.  The trie is flattened into a double array of
// 8-byte slots ordered hot-first so that the slots the scan kernel touches most sit
// at the lowest indices, which is the part the kernel stages in LDS.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace tokscore {

struct DfaImage;   // dfa.h
struct DfaOptions;

// Packed slot t (one byte-trie node), two 32-bit words:
//   w0 = label : 8   the byte on the edge from the parent
//        valid : 1   0 for empty slots and for the root
//        base  : 23  children live at slot base + byte; 0 = leaf
//   w1 = id          id emitted when a match ends here (token index + 2); 0 = none
// Slot 0 is the root.  Every non-leaf node has a base of its own (no two nodes share
// one), so a slot's owner is identified by base = t - label alone: a lookup of byte c
// under a node with base b reads slot t = b + c and hits iff that slot is valid and its
// label is c.  Hits AND misses cost exactly one 8-byte read and a 9-bit compare.
struct PackedTrie {
	static constexpr uint32_t kValid = 0x100u;
	static constexpr uint32_t kKeyMask = 0x1FFu;
	static constexpr uint32_t kBaseShift = 9;
	static constexpr uint64_t kMaxSlots = 1ull << 23;

	std::vector<uint64_t> slots;
	uint64_t nodes = 0;
	uint64_t max_len = 0;    // SolutionFile::max_len_ (bytes), duplicates included
	uint32_t unknown_id = 0; // V + 2

	static uint32_t w0(uint64_t e) { return (uint32_t)e; }
	static uint32_t base(uint64_t e) { return (uint32_t)e >> kBaseShift; }
	static uint32_t id(uint64_t e) { return (uint32_t)(e >> 32); }
	static bool matches(uint64_t e, uint32_t c) { return ((uint32_t)e & kKeyMask) == (c | kValid); }
};

// The stream scan's image of the vocabulary (kernels_stream.hip; VocabTrie::PackStream): a second
// double array, 3 bytes per slot instead of 8, small enough for a 32k-token vocabulary to sit in the
// 160 KB of LDS whole.  It lives in CODE space: the bytes that label some trie edge get the codes
// 1 .. k (k <= 125), byte 0 -- the separator the prepared corpus stores after every document -- is
// code 0, every other byte shares the code k + 1 ("no edge has this label"); the prepared corpus
// is stored as codes, so a lookup is slot t = base + code:
//   lt[t]      label code (7 bits) | 0x80 when a token ends on the slot
//   base[t]    base of the node's own children, 0 = none (full 16 bits: bases are unique per node,
//              and with most nodes having children every value is needed).  The root's base is 0
//              too: after a token's last slot the next lookup is at the root again.
//   slot_id[t] the id a token end on the slot counts for (only the histogram's reduction reads it)
// The root row is dense and every slot of it a token end: a byte that is no token of its own counts
// as UNKNOWN (SolutionFile.cpp:92-94), code 0 as nothing (kSkipSlot).  Edges labelled with byte 0 are
// left out: documents hold no NUL (DataFile.cpp:27-29), so they can never be followed, and no walk
// may continue over a separator (the window clip of SolutionFile.cpp:90).  Empty slots are all
// zero: only code 0 "hits" them, finds base 0 and no token end, which is what a miss does.
struct StreamImage {
	static constexpr uint64_t kMaxSlots = 65534;        // slot indices travel as u16 stream entries (0xFFFF = null)
	static constexpr uint32_t kSkipSlot = 0xFFFFFFFFu;
	static constexpr uint32_t kCodes = 128;
	std::vector<uint8_t> lt;
	std::vector<uint16_t> base;
	std::vector<uint32_t> slot_id;
	uint8_t code[256] = {};
	uint32_t first_cont_code = 0; // a byte starts a code point (or is a separator) iff code[byte] < first_cont_code
	uint32_t root_base = 0;
	bool valid = false;
};

class VocabTrie {
public:
	// Token i = bytes[off[i]..off[i+1]).  Returns false (with *err set) when no token
	// is non-empty: SolutionFile::Tokenize would never advance (SolutionFile.cpp:89-97).
	bool Build(const uint8_t *bytes, const uint64_t *off, uint64_t V, std::string *err);

	// Visit weights for the hot-first layout: greedy-segment `sample` (already a
	// concatenation of whole documents, boundaries ignored on purpose -- this only
	// steers the layout, never results) and count the lookups made at every node.
	void AddHotnessSample(const uint8_t *sample, uint64_t n);

	// Flatten to the double array.  Returns false when it would need more than 2^23 slots.
	bool Pack(PackedTrie *out, std::string *err);

	// Enable/disable token i; returns the slot whose id field changed (or -1) and the
	// new packed value for it so the caller can patch the device copy.
	int64_t Toggle(uint64_t token, bool enable, PackedTrie *packed, uint64_t *new_entry);

	// The stream image (hot-first like Pack).  False when it cannot be built: more than 125
	// distinct edge labels, or more than StreamImage::kMaxSlots slots.
	bool PackStream(StreamImage *out);
	// After Toggle(token): bring the image's slot of that token's node up to date; returns the
	// slot (or -1 when the node is not in the image / nothing changed).
	int64_t PatchStream(uint64_t token, StreamImage *img) const;
	// What PatchStream would do after Toggle(token, enable), with nothing changed: the token's slot
	// in the image (-1: none) and whether its token-end bit would flip (the only thing a toggle can
	// change about where tokens end; the id a slot counts for does not move a boundary).
	int64_t PeekStream(uint64_t token, bool enable, const StreamImage &img, bool *flips);
	// Closed-vocabulary occurrence counts from "where did the walk from every position end":
	// slot_ends[t] = walks (of at most max_cp code points, inside one document) whose deepest match
	// is slot t of the image.  A string occurs wherever a walk reaches its node, i.e. ends there or
	// below: counts[id] (V + 3 entries, cleared here) gets the subtree sum on the id the string's
	// first enabled entry has (Trie.cpp:114-128 counts strings of whole code points only: an entry
	// that is not valid UTF-8 on its own stays 0).
	void AccumulateEnds(const uint64_t *slot_ends, uint32_t max_cp, uint64_t *counts) const;
	bool Enabled(uint64_t token) const { return token < V_ && enabled_[token] != 0; }

	// The vocabulary as it is now (enabled entries only) as the one-transition-per-byte automaton of
	// dfa.h.  False (with *why set) when it cannot be built: too many distinct edge bytes, or more
	// slots than opt.max_slots.
	bool BuildDfa(DfaImage *out, const DfaOptions &opt, std::string *why) const;

	// Rows of a per-string table (one row per distinct non-empty token string): slot_row[t] = row
	// of the node at slot t or -1, tok_row[i] = row of token i's string or -1.  After Pack.
	uint32_t StringRows(uint64_t nslots, std::vector<int32_t> *slot_row, std::vector<int32_t> *tok_row) const;

	uint64_t vocab_size() const { return V_; }
	uint64_t max_len() const { return max_len_; }
	uint64_t node_count() const { return (uint64_t)first_child_.size(); }

	// Longest enabled token that prefixes text (host check of the packed image).
	static void Longest(const PackedTrie &p, const uint8_t *text, uint64_t n, uint64_t *len, uint64_t *id);

private:
	struct Edge {
		uint8_t byte;
		uint32_t child;
	};
	uint32_t Child(uint32_t node, uint8_t c) const;
	uint32_t ActiveId(uint32_t node) const;

	uint64_t V_ = 0;
	uint64_t max_len_ = 0;
	// node arrays (node 0 = root)
	std::vector<uint32_t> first_child_; // index into edges_ (contiguous per node after Build)
	std::vector<uint32_t> n_children_;
	std::vector<Edge> edges_;
	std::vector<uint32_t> depth_;
	std::vector<int64_t> head_tok_;   // first token index whose string ends here, -1 none
	std::vector<uint64_t> weight_;    // lookups made at this node in the hotness samples
	std::vector<uint32_t> root_child_; // direct table for the root
	// token arrays
	std::vector<int64_t> next_dup_;   // next token with the same string, -1 none
	std::vector<uint32_t> tok_node_;  // node of token i (0 for empty strings)
	std::vector<uint8_t> enabled_;
	std::vector<uint32_t> node_slot_; // filled by Pack
	std::vector<uint32_t> node_sslot_; // filled by PackStream: slot in the stream image, 0 = not in it
	void StreamSlotValues(uint32_t node, uint32_t code, uint8_t *lt, uint32_t *slot_id) const;
	bool sampled_ = false;
};

} // namespace tokscore
