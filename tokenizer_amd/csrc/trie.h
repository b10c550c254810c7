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

// Packed slot t (one byte-trie node), little end first:
//   parent : slot_bits   slot index of the parent node (all ones = empty slot)
//   base   : slot_bits   children live at slot base + byte; 0 = leaf
//   id     : id_bits     id emitted when a match ends here (token index + 2); 0 = none
// Slot 0 is the root.  A lookup of byte c under node n reads slot t = base(n) + c and
// hits iff parent(t) == n, so hits AND misses cost exactly one 8-byte read.
struct PackedTrie {
	std::vector<uint64_t> slots;
	uint32_t slot_bits = 0;
	uint32_t id_bits = 0;
	uint64_t nodes = 0;
	uint64_t max_len = 0;    // SolutionFile::max_len_ (bytes), duplicates included
	uint32_t unknown_id = 0; // V + 2

	uint32_t pmask() const { return slot_bits >= 32 ? 0xFFFFFFFFu : ((1u << slot_bits) - 1u); }
	uint32_t parent(uint64_t e) const { return (uint32_t)(e & pmask()); }
	uint32_t base(uint64_t e) const { return (uint32_t)((e >> slot_bits) & pmask()); }
	uint32_t id(uint64_t e) const { return (uint32_t)(e >> (2 * slot_bits)); }
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

	// Flatten to the double array.  Returns false when 2*slot_bits + id_bits > 64.
	bool Pack(PackedTrie *out, std::string *err);

	// Enable/disable token i; returns the slot whose id field changed (or -1) and the
	// new packed value for it so the caller can patch the device copy.
	int64_t Toggle(uint64_t token, bool enable, PackedTrie *packed, uint64_t *new_entry);

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
	bool sampled_ = false;
};

} // namespace tokscore
