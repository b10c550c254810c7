// candidates.h -- the candidate set the annealer works on: what the reference keeps as
// std::vector<annealing::Token> (src/tokenizer/Token.h:38-93), as parallel arrays.
//
//   Token i       one distinct window of 1 .. max_cp code points that occurs in the corpus,
//                 in trie pre-order with children sorted by code point (Trie.cpp:81-91,137-142)
//   chr[i]        FIRST code point of the string (BuildToken hands `fst` down, Trie.cpp:81-89)
//   uses[i]       occurrences, the trie node's freq (Trie.cpp:82,114-128)
//   lpar[i]       index of the string minus its first code point, rpar[i] of the string minus
//                 its last one (Trie.cpp:92-103); kNone for single code points
//
// Host code only (the move loop stays on the host).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace tokanneal {

constexpr uint32_t kNone = 0xFFFFFFFFu; // the file format's "no parent" (Token.cpp:187-190)

// annealing::TokenReadErrCode (Token.h:13-19), same values
enum ReadCode { READ_OK = 0, READ_BAD_STREAM = 1, READ_INVALID_VERSION = 2, READ_BAD_FILE = 3, READ_NOT_DONE = 4 };

struct CandidateSet {
	std::vector<uint32_t> chr;
	std::vector<uint64_t> uses;
	std::vector<uint32_t> lpar, rpar;

	size_t size() const { return chr.size(); }
	void clear() {
		chr.clear(), uses.clear(), lpar.clear(), rpar.clear();
	}
	// Token::size() (Token.cpp:56-64): code points, along the r (prefix) chain
	size_t TokenSize(size_t i) const;
	// Token::GetName() (Token.cpp:65-74): first code points along the l (suffix) chain, UTF-8
	std::string GetName(size_t i) const;
	// structural check of parent indices (both chains end, in range); false = corrupt input
	bool Validate(std::string *why) const;
};

// FileCandidates + Trie::BuildTokens (GetTokens.cpp:21-27,31-90; Trie.cpp:114-152) over documents
// packed back to back: every window of 1 .. max_cp code points starting at every code-point
// position of every document, clipped at the document end.  UTF-8 is decoded with the reference's
// rules (utf8/unchecked.h:104-131: the lead byte alone decides the length).
void BuildCandidates(const uint8_t *bytes, const uint64_t *off, uint64_t n_docs, uint32_t max_cp, CandidateSet *out);

// `.candidates-*.bin` (Token.cpp:109-193): "0.4.0\0", u64 count, per token UTF-8 of chr + LEB128
// of uses, then per token u32 l parent, u32 r parent (0xFFFFFFFF = none).
// The writer is byte-identical to WriteTokens (a token with uses == 0 gets no varint byte at
// all, Token.cpp:177-183, which ReadTokens cannot parse back: rejected here).
bool WriteCandidates(const CandidateSet &c, const std::string &path, std::string *why);
// The reader takes what the writer produces, with the full 64-bit varint (the reference's
// `(byte & 0x7F) << sft` is an int shift, Token.cpp:136, and breaks above 2^31 uses).
ReadCode ReadCandidates(const std::string &path, CandidateSet *out);

// The candidate set of two corpora together, from the sets of each: what GetTokens builds when
// FileCandidates feeds several files into one Trie (GetTokens.cpp:31-90).  Both inputs in
// the builder's order (trie pre-order, children by code point); false + *why otherwise.
// The inputs have passed Validate().
bool MergeCandidates(const CandidateSet &a, const CandidateSet &b, CandidateSet *out, std::string *why);

// utf8::append (utf8/core.h:357-377)
void AppendUtf8(uint32_t cp, std::string *s);

} // namespace tokanneal
