/*
 * tokscore.h -- C-ABI of libtokscore.so, the MI355X (gfx950) corpus scorer.
 *
 * The reference (asdf2727/tokenizer) is one executable with no plugin or FFI
 * surface (CMakeLists.txt:8); the seams this library sits behind are C++ member
 * functions.  Each entry point below names the reference interface it replaces
 * (paths relative to /root/reference).  Plain pointers and sizes only: no C++
 * or torch types cross this boundary.  All functions return a TOKSCORE_* status
 * (the reference reports errors through flags/enums, never exceptions:
 * src/files/JsonFile.h:17, src/tokenizer/Token.h:13-19).
 *
 * Semantics (bit-exact with SolutionFile::Tokenize, src/files/SolutionFile.cpp:83-100):
 *   - ids: 0 = <START>, 1 = <END>, vocabulary entry i -> id i+2; duplicate strings
 *     keep the first id; UNKNOWN ((size_t)-1 in the reference) is reported in the
 *     extra bucket V+2.
 *   - the corpus is ASCII-lower-cased (A-Z only) before matching, the vocabulary is
 *     not; matching is on bytes, greedy longest-first, UNKNOWN advances one byte.
 *   - documents are independent; a document is text bytes [off[d], off[d+1]).
 *
 * One context drives one GPU and is not thread-safe; independent contexts are.
 */
#ifndef TOKSCORE_H
#define TOKSCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	TOKSCORE_OK = 0,
	TOKSCORE_E_ARG = 1,        /* null pointer / inconsistent sizes / offsets not monotone */
	TOKSCORE_E_NO_DEVICE = 2,  /* no usable HIP device: there is NO CPU fallback */
	TOKSCORE_E_HIP = 3,        /* a HIP runtime call failed (see tokscore_last_error) */
	TOKSCORE_E_STATE = 4,      /* vocabulary or corpus not set yet */
	TOKSCORE_E_VOCAB = 5,      /* vocabulary unusable: no non-empty token (the reference
	                              would loop forever, SolutionFile.cpp:89-97), or too large
	                              for the packed trie (more than 2^23 double-array slots) */
	TOKSCORE_E_CAPACITY = 6,   /* output buffer too small / internal stream overflow */
	TOKSCORE_E_INTERNAL = 7
};

typedef struct tokscore_ctx tokscore_ctx;

/* Library / context --------------------------------------------------------- */

/* ABI version of this header (major*100 + minor). */
int tokscore_abi_version(void);

/* Number of HIP devices visible (0 when there is no GPU). */
int tokscore_device_count(void);

/* Create a scorer bound to HIP device `device`.  Fails with TOKSCORE_E_NO_DEVICE
 * when no GPU is present.  Replaces: constructing the per-process scorer state the
 * reference keeps in SolutionFile (src/files/SolutionFile.h:9-12). */
int tokscore_create(tokscore_ctx **out, int device);
void tokscore_destroy(tokscore_ctx *ctx);

/* Human-readable text for a status code / the last failure inside ctx (ctx = NULL: why the last
 * tokscore_create on this thread failed). */
const char *tokscore_strerror(int status);
const char *tokscore_last_error(const tokscore_ctx *ctx);

/* Vocabulary in --------------------------------------------------------------
 * Replaces SolutionFile(const std::vector<std::string>&, path) /
 * SolutionFile(path) id assignment (src/files/SolutionFile.cpp:43-72) and the
 * tokens_ map behind GetId (:74-77).  Token i = bytes[off[i] .. off[i+1]). */
int tokscore_set_vocab(tokscore_ctx *ctx, const uint8_t *bytes, const uint64_t *off, uint64_t V);

/* max_len_ in BYTES over the whole vocabulary (SolutionFile.cpp:55,68). */
int tokscore_vocab_max_len(const tokscore_ctx *ctx, uint64_t *max_len);

/* Annealing move on the scored vocabulary: disable / re-enable entry `token_index`
 * without rebuilding the trie.  A disabled entry can never be emitted; if a later entry
 * holds the same string it takes over, exactly as in the vocabulary without the entry.
 * The call only queues the change (one 8-byte trie slot); the next evaluation applies it.
 * There is no reference counterpart on the corpus path: the reference judges moves on the
 * candidate DAG (src/tokenizer/Token.cpp:83-105); parity is defined as "equal to a full
 * score with the toggled vocabulary". */
int tokscore_toggle(tokscore_ctx *ctx, uint64_t token_index, int enable);

/* Corpus in --------------------------------------------------------------------
 * Replaces DataFile::GetEntries() (src/files/DataFile.cpp:24-31) as the document
 * source: D documents packed back to back, raw (NOT lower-cased) bytes, document d =
 * bytes[off[d] .. off[d+1]).  Host buffers; copied to HBM once and kept resident. */
int tokscore_set_corpus(tokscore_ctx *ctx, const uint8_t *bytes, const uint64_t *off, uint64_t D);

/* Same, for buffers that already live in this device's memory (no copy is made; the
 * caller keeps them alive and unchanged while the context uses them).  d_bytes must be
 * readable for 256 bytes past off[D] (the scan prefetches a halo). */
int tokscore_set_corpus_device(tokscore_ctx *ctx, const void *d_bytes, const void *d_off, uint64_t D,
                               uint64_t total_bytes);

/* Scoring ----------------------------------------------------------------------
 * One full-vocabulary evaluation of the resident corpus.  Replaces the loop at
 * src/main.cpp:38-43 (Tokenize per article, init_size/comp_size accumulation) plus the
 * histogram of the ids Tokenize returns.
 *   counts[V+3]   counts[id] for id 0..V+1 (START/END once per document) and
 *                 counts[V+2] = UNKNOWN emissions
 *   total_tokens  sum over documents of ids.size() - 2      (main.cpp:41)
 *   total_bytes   sum over documents of text.size()          (main.cpp:40) */
int tokscore_score(tokscore_ctx *ctx, uint64_t *counts, uint64_t *total_tokens, uint64_t *total_bytes);

/* Asynchronous form: results stay on the device in d_out[V+5] =
 * counts[V+3] | total_tokens | total_bytes (the vector a multi-GPU caller all-reduces),
 * enqueued on `stream` (a hipStream_t, or NULL for the context's own stream). */
int tokscore_score_async(tokscore_ctx *ctx, void *d_out, void *stream);

/* What an asynchronous caller cannot see in d_out: waits for `stream` (NULL: the context's own),
 * then *overflowed = 1 when the last tokscore_score_async enqueued there ran out of its internal
 * token-stream capacity -- d_out is then incomplete and the evaluation has to be redone with
 * tokscore_score, which falls back by itself (the reference has no capacity to exceed,
 * src/files/SolutionFile.cpp:84-99 grows a vector; the stream scan's capacity is a true upper
 * bound, so only the tile scan on adversarial input can report 1).  One context = one evaluation
 * in flight on one stream: the scratch buffers of an evaluation belong to the context. */
int tokscore_async_status(tokscore_ctx *ctx, void *stream, int *overflowed);

/* Re-score after tokscore_toggle calls: identical results to tokscore_score, bit for bit.
 * Only the work chunks that contain a toggled entry's string are scanned again (with the
 * trie as it was: subtracted; as it is: added) on top of the previous evaluation's counts,
 * which stay in device memory.  Falls back to a full evaluation by itself when there is no
 * previous evaluation, more than 8 toggles are pending, or the strings occur in more than
 * 1/3 of the chunks; tokscore_info.rescored_chunks reports what the call did. */
int tokscore_score_incremental(tokscore_ctx *ctx, uint64_t *counts, uint64_t *total_tokens,
                               uint64_t *total_bytes);

/* Batched move evaluation: what each of K annealing moves, taken alone, would do to the token
 * count of the resident corpus -- the cost a move is judged on.  Move k = "set entry
 * token_index[k] to enable[k]" against the vocabulary as it is now (toggles queued with
 * tokscore_toggle are evaluated first); delta_total_tokens[k] = total_tokens with the move minus
 * total_tokens without it, exactly what two full tokscore_score calls would give.  The moves are
 * independent and nothing is changed: no entry is toggled, counts and state stay as they were.
 * There is no reference counterpart on the corpus path: the reference estimates a move on the
 * candidate DAG (TokenGenerator::SimulateStep, src/tokenizer/TokenGenerator.cpp:87).
 * One batch = one look-up of the strings' chunks in the occurrence index, one search for the
 * streams (about 256 bytes between two guaranteed token starts) that hold an occurrence, and one
 * launch that walks each such stream with and without the move.  Vocabularies or corpora the
 * stream scan does not take are served move by move with tokscore_score_incremental. */
int tokscore_score_toggles(tokscore_ctx *ctx, const uint64_t *token_index, const int32_t *enable, uint64_t K,
                           int64_t *delta_total_tokens);

/* Closed-vocabulary all-occurrence counting: what
 * annealing::ExtractCandidates + Trie::AddString (src/tokenizer/GetTokens.cpp:21-27,
 * src/tokenizer/Trie.cpp:114-128) add up for the strings of the vocabulary -- for every
 * code-point position of every document of the resident corpus and every window of
 * 1 .. max_cp code points starting there (clipped at the document end), the window's string
 * gets +1 if it is an enabled vocabulary string.  These are the `uses` GetTokens seeds each
 * Token with (Trie.cpp:82).  No lower-casing on this path; the corpus must be valid UTF-8
 * (the reference's unchecked decoder is undefined otherwise).
 *   counts[V+3]   indexed by id like tokscore_score: counts[i + 2] for token i (a repeated
 *                 string counts on its first id); counts[0], counts[1], counts[V+2] are 0
 *   positions     number of code-point positions (the trie root's freq, Trie.cpp:117) */
int tokscore_count_occurrences(tokscore_ctx *ctx, uint32_t max_cp, uint64_t *counts, uint64_t *positions);

/* tokscore_count_occurrences for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Same counts, bit for bit, left in device memory: d_out[V+4] = counts[V+3] | positions
 * (uint64), enqueued on `stream` (a hipStream_t, or NULL for the context's own); the call does not
 * wait.  It calls no hipMalloc / hipFree: the kernels' scratch (tile queue heads, the first
 * document of every 1 KiB tile, one row of V+4 partial counts per workgroup) lives in a 16 MiB
 * array in the library's device image, one copy per GPU, held until the process ends.
 *   - TOKSCORE_E_CAPACITY when 8 bytes per KiB of corpus plus one row (4 * (V + 4) bytes) exceed
 *     the 16 MiB; fewer rows than tokscore_count_occurrences would take only make it slower.
 *   - TOKSCORE_E_STATE until the packed trie is in device memory: copying it there is set-up and
 *     allocates.  One tokscore_count_occurrences does it (so does any tokscore_toggle followed by an
 *     evaluation, or an evaluation of a vocabulary the automaton scan does not serve; a tokscore_score
 *     that the automaton scan serves does not: it needs no packed trie).  Also while toggles
 *     are pending (tokscore_toggle since the last evaluation): the next evaluation brings them to the
 *     device, this call changes nothing of the context's state.
 *   - Every call on a GPU, from any context, shares that GPU's array: two calls that overlap on
 *     one GPU (two streams, two contexts, two threads) race.  Calls on one stream are ordered. */
int tokscore_count_occurrences_static(tokscore_ctx *ctx, uint32_t max_cp, void *d_out, void *stream);

/* Open-vocabulary candidate extraction: the candidate set
 * annealing::GetTokens builds from the corpus -- FileCandidates + Trie::BuildTokens
 * (src/tokenizer/GetTokens.cpp:21-90, src/tokenizer/Trie.cpp:81-152) -- for the resident
 * corpus, computed on the GPU by sorting all windows of 1 .. max_cp code points (no trie is
 * built).  Token t of the result is the t-th node of the reference's trie in pre-order
 * (children by code point): chr[t] = FIRST code point of its string (Trie.cpp:81-89),
 * uses[t] = its occurrence count (Trie.cpp:82), lpar[t] / rpar[t] = index of the string minus
 * its first / last code point, 0xFFFFFFFF for single code points (Trie.cpp:92-103) -- the
 * arrays tokanneal_cands_from_arrays (include/tokanneal.h) takes, and what WriteTokens stores.
 * The corpus must be valid UTF-8 without NUL bytes, below 4 GiB; max_cp <= 32.
 *   tokscore_extract_candidates  runs the extraction; *n_tokens = number of tokens,
 *                                *positions = code points in the corpus (may be NULL)
 *   tokscore_candidates_copy     copies the arrays (n_tokens entries each; any pointer may be
 *                                NULL) to the host and releases the device copy */
int tokscore_extract_candidates(tokscore_ctx *ctx, uint32_t max_cp, uint64_t *n_tokens, uint64_t *positions);
int tokscore_candidates_copy(tokscore_ctx *ctx, uint32_t *chr, uint64_t *uses, uint32_t *lpar, uint32_t *rpar);

/* Per-document id sequence, the exact return value of
 * SolutionFile::Tokenize(std::string) (src/files/SolutionFile.cpp:83-100) for one
 * document of raw bytes; UNKNOWN is UINT64_MAX as in the reference.  ids_out has room
 * for `cap` entries (n+2 always suffices); *n_ids receives the count. */
int tokscore_tokenize(tokscore_ctx *ctx, const uint8_t *text, uint64_t n, uint64_t *ids_out, uint64_t cap,
                      uint64_t *n_ids);

/* Introspection for benchmarks / profiling -------------------------------------- */
typedef struct {
	uint64_t vocab_size;      /* V */
	uint64_t max_len;         /* bytes */
	uint64_t trie_nodes;      /* byte-trie nodes incl. root */
	uint64_t trie_slots;      /* double-array slots (8 B each) */
	uint64_t trie_lds_slots;  /* slots staged in LDS by the scan kernel */
	uint64_t corpus_bytes;
	uint64_t corpus_docs;
	uint64_t chunks;          /* wave work units of the scan kernel */
	uint32_t slot_bits;       /* packed slot = label:8 | valid:1 | base:slot_bits, then id:id_bits */
	uint32_t id_bits;
	uint32_t scan_path;       /* what the last full evaluation ran on: 2 = LDS stream scan (one range of the
	                           * corpus per lane, trie / automaton image in LDS), 1 = LDS tile scan kernel,
	                           * 0 = generic global-memory scan */
	uint32_t rescored_chunks; /* chunks re-scanned by the last tokscore_score_incremental;
	                           * UINT32_MAX when it ran (or fell back to) a full evaluation */
	uint64_t toggle_streams;  /* streams walked (twice each) by the batched path of tokscore_score_toggles
	                           * since the context was created; unchanged by a call the move-by-move
	                           * fallback served */
	/* the automaton scan (one transition per corpus byte; tokenizer_amd/csrc/dfa.h), which serves full
	 * evaluations whenever its image can be built for the vocabulary; 0 everywhere when it cannot */
	uint64_t dfa_slots;         /* 4-byte slots of the image, real + virtual + empty */
	uint64_t dfa_virtual_slots;
	uint32_t dfa_lds_slots;     /* slots of the array staged in LDS (the entries the corpus sample used); the rest is read from L2 */
	uint32_t dfa_active;        /* 1: the last full evaluation ran on it */
	uint32_t dfa_chains;        /* ranges of the corpus each lane steps side by side */
	uint32_t dfa_range_bytes;
	uint32_t base_scan_path;    /* the plan's own scan (incremental re-scores, moves, evaluations after toggles) */
	uint32_t dfa_lds_used;      /* entries in the LDS array */
	double dfa_hot_fraction;    /* share of the hotness sample's lookups that fall on the LDS prefix */
	double dfa_build_ms;        /* host time of the last image build */
	double dfa_plan_ms;         /* image + prepared corpus + buffers, once per corpus and vocabulary */
	uint64_t dfa_second_hits;   /* lookups the L2 array answered with a hit, when the scan's counters were last read */
	uint32_t dfa_relayouts;     /* times those counters were read to choose again what LDS holds (once per vocabulary state; TOKSCORE_DFA_RELAYOUTS) */
	uint32_t search_passes;     /* passes of a search automaton over the corpus that tokscore_score_toggles has made (up to 256 moves each) */
	uint32_t dfa_form;          /* 0: no image; 1: flattened (one lookup per byte, virtual entries); 2: failure form (the trie's own
	                             * edges + one failure entry per state, f followed state by state: vocabularies whose prefixes are
	                             * mostly not tokens, e.g. what TokenGenerator::GetSolution writes) */
	uint32_t reserved0;
} tokscore_info;
int tokscore_get_info(tokscore_ctx *ctx, tokscore_info *info);

/* Time of the last tokscore_score* call's kernels in milliseconds, measured with HIP
 * events on the stream the kernels ran on: [0] whole evaluation, [1] scan kernel,
 * [2] boundary fix-up kernels, [3] histogram + reduce kernels. */
int tokscore_last_timing(tokscore_ctx *ctx, float ms[4]);

/* Same four durations for every score call since the previous drain (oldest first, at most
 * the last 256 calls), so a benchmark can enqueue many evaluations back to back and read the
 * kernel times afterwards without a host sync inside its timed region.  ms holds 4*cap
 * floats; *n receives the number of calls reported. */
int tokscore_drain_timings(tokscore_ctx *ctx, float *ms, uint32_t cap, uint32_t *n);

/* Measurement aid; no counterpart in the reference.  Which HIP events tokscore_score_async and
 * tokscore_score record: mode 0 none, 1 all (the default; TOKSCORE_TIMING=0 starts a context at 0),
 * 2 only the two around the scan kernel, and only on every `every`-th full evaluation (0 = 1) --
 * the other evaluations carry none and are not reported by tokscore_drain_timings; a reported one
 * has [0] = [1] = the scan kernel, [2] = [3] = 0.  An event record holds the stream for 3 - 4 us,
 * so a benchmark that times many evaluations back to back samples instead of marking every one. */
int tokscore_set_timing(tokscore_ctx *ctx, int mode, uint32_t every);

/* Measurement aid ("report against a measured read-only streaming kernel
 * on the same box"); no counterpart in the reference.  Reads a zeroed scratch buffer of `bytes`
 * (>= 1 MiB; allocated and freed inside) `reps` times with a kernel that only loads, and
 * reports the rate in 10^9 bytes per second.  Buffers below the 256 MB last-level cache
 * measure that cache, larger ones HBM. */
int tokscore_probe_read_bandwidth(tokscore_ctx *ctx, uint64_t bytes, uint32_t reps, double *gb_per_s);

/* Host-side self-check of the packed double-array trie (no GPU involved, used by the
 * CPU test-suite to validate the builder): longest enabled vocabulary entry that is a
 * prefix of text[0..n).  *len = 0 when none.  Never used by any scoring entry point. */
int tokscore_hosttrie_longest(tokscore_ctx *ctx, const uint8_t *text, uint64_t n, uint64_t *len,
                              uint64_t *id);

/* Host-side self-check of the scan automaton's builder (tokenizer_amd/csrc/dfa.h; no GPU involved,
 * used by the CPU test-suite): builds the one-transition-per-byte image of the vocabulary as it is
 * (toggles included) with the documents as the hotness sample, runs it over the documents on the
 * host exactly as the scan kernel steps it, and returns what its events add up to -- counts[V+3]
 * and total_tokens as tokscore_score defines them (SolutionFile.cpp:83-100, main.cpp:38-43).
 * max_slots = 0: the library's default limit; lds_slots = 0: the LDS array's default size.  stats
 * (may be NULL): slots, real entries, virtual entries, codes, entries in the LDS array, reads
 * of the global array during the run, sample steps, of those on virtual entries.  TOKSCORE_E_VOCAB when the image cannot be built (the scoring entry points then use the
 * other scan paths).  Never used by any scoring entry point. */
int tokscore_hostdfa_check(tokscore_ctx *ctx, const uint8_t *bytes, const uint64_t *off, uint64_t D, uint64_t max_slots,
                           uint32_t lds_slots, uint64_t *counts, uint64_t *total_tokens, uint64_t stats[8]);

/* Host-side self-check of the search automaton tokscore_score_toggles builds for a batch of strings
 * (tokenizer_amd/csrc/dfa.h: BuildSearchImage; no GPU, no context): strings k = codes[off[k] .. off[k+1])
 * over the alphabet 1 .. ncodes-1, run over text[0 .. text_n) as the search kernel steps it.  Writes up
 * to cap (string, position of its first byte) pairs in the order found, *n_found = all there were;
 * stats (may be NULL): LDS slots, states, entries.  TOKSCORE_E_CAPACITY when the automaton does not
 * fit max_slots.  Never used by any scoring entry point. */
int tokscore_hostsearch_check(const uint8_t *codes, const uint32_t *off, uint32_t n, uint32_t ncodes, uint32_t max_slots,
                              const uint8_t *text, uint64_t text_n, uint64_t *pairs, uint64_t cap, uint64_t *n_found,
                              uint64_t stats[3]);

/* Contexts without a GPU, for the host-side checks above only: set_vocab and
 * hosttrie_longest work, every scoring call fails with TOKSCORE_E_NO_DEVICE. */
int tokscore_create_hostonly(tokscore_ctx **out);

#ifdef __cplusplus
}
#endif
#endif
