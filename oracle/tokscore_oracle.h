/*
 * tokscore_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's corpus-scoring algorithm:
 *   SolutionFile::Tokenize   /root/reference/src/files/SolutionFile.cpp:83-100
 *   SolutionFile::GetId      /root/reference/src/files/SolutionFile.cpp:74-77
 *   id assignment / max_len_ /root/reference/src/files/SolutionFile.cpp:43-72
 *   accumulation             /root/reference/src/main.cpp:31-46
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (tokenizer_amd/, libtokscore.so) never
 * links, imports or calls anything in oracle/.
 *
 * Parity pin: validated bit-for-bit against the compiled reference
 * (oracle/_ref/ref_harness, built from /root/reference sources in place by
 * oracle/Makefile) on every case under tests/golden/ -- see
 * tests/test_oracle_golden.py.  The reference ships no tests of its own
 * (SURVEY.md section 4), so those generated goldens are the pin.
 */
#ifndef TOKSCORE_ORACLE_H
#define TOKSCORE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_UNKNOWN ((uint64_t)-1) /* SolutionFile.cpp:76, (size_t)-1 */

typedef struct ora_vocab ora_vocab;

/* Vocabulary = V byte strings, token i = bytes[off[i] .. off[i+1]).
 * ids: 0=<START>, 1=<END>, token i -> i+2 (SolutionFile.cpp:49-57);
 * duplicate strings keep the FIRST id (unordered_map::insert fails, :56). */
ora_vocab *ora_vocab_new(const uint8_t *bytes, const uint64_t *off, uint64_t V);
void ora_vocab_free(ora_vocab *v);
uint64_t ora_vocab_size(const ora_vocab *v);    /* V */
uint64_t ora_vocab_max_len(const ora_vocab *v); /* max_len_ in BYTES (:55) */

/* SolutionFile::GetId (:74-77): exact-string lookup, ORA_UNKNOWN if absent. */
uint64_t ora_get_id(const ora_vocab *v, const uint8_t *s, uint64_t n);

/* SolutionFile::Tokenize (:83-100).  ids_out needs room for n+2 entries.
 * Returns the number of ids written (>= 2: START ... END). */
uint64_t ora_tokenize(const ora_vocab *v, const uint8_t *text, uint64_t n,
                      uint64_t *ids_out);

/* main.cpp:31-46 accumulation over D documents, doc d = corpus[off[d]..off[d+1]).
 * counts has V+3 entries: counts[id] for id in 0..V+1 (START and END are
 * counted once per document, exactly as Tokenize emits them) and
 * counts[V+2] = number of UNKNOWN ((size_t)-1) emissions.
 * total_tokens = sum(ids.size() - 2), total_bytes = sum(text.size()).
 * threads <= 1 runs single-threaded; otherwise documents are split into
 * `threads` contiguous ranges balanced by bytes (one pthread each). */
void ora_score(const ora_vocab *v, const uint8_t *corpus, const uint64_t *off,
               uint64_t D, uint64_t *counts, uint64_t *total_tokens,
               uint64_t *total_bytes, int threads);

/* Closed-vocabulary all-occurrence counting: annealing::ExtractCandidates +
 * Trie::AddString (/root/reference/src/tokenizer/GetTokens.cpp:21-27,
 * Trie.cpp:114-128) restricted to the strings of `v`: every document is cut
 * into code points (utf8::unchecked::utf8to32; lead byte -> sequence length as
 * utf8/core.h:134-147), and for every code-point position i and every length
 * k = 1 .. min(remaining, max_cp) code points the string [i, i+k) gets +1 if
 * it is in the vocabulary.  NO lower-casing on this path.  counts has V+3
 * entries indexed by id like ora_score (ids 0, 1 and V+2 stay 0); a duplicate
 * string counts on its first id.  *positions = number of code-point positions
 * (the trie root's freq, Trie.cpp:117).  Valid UTF-8 only: the reference's
 * unchecked decoder reads past the end of a truncated sequence.
 * Pinned by tests/golden/cands_* and dag_tiny.json (Token uses == freq,
 * Trie.cpp:82). */
void ora_count_occurrences(const ora_vocab *v, const uint8_t *corpus, const uint64_t *off,
                           uint64_t D, uint32_t max_cp, uint64_t *counts, uint64_t *positions,
                           int threads);

#ifdef __cplusplus
}
#endif
#endif
