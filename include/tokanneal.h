/* tokanneal.h -- C ABI of libtokanneal.so: the host side of the vocabulary search.
 *
 * the reference judges annealing moves on a DAG of
 * candidate strings, on the CPU, and so does this library (no GPU code in here; the GPU
 * scorer is libtokscore.so, include/tokscore.h).  It restates
 *
 *   annealing::GetTokens / Trie::BuildTokens     src/tokenizer/GetTokens.cpp:21-115,
 *                                                src/tokenizer/Trie.cpp:81-152
 *   annealing::ReadTokens / WriteTokens          src/tokenizer/Token.cpp:109-193
 *   annealing::Token::SimulateStep / ApplyStep   src/tokenizer/Token.cpp:17-37,83-105
 *   annealing::TokenGenerator                    src/tokenizer/TokenGenerator.cpp:17-217
 *   LomaxDist                                    src/tokenizer/LomaxDist.cpp:5-51
 *
 * and hands its result over in the packed form tokscore_set_vocab takes.  The reference has
 * no FFI; the entry points below are what a binding of those classes would need.
 *
 * Conventions: int status (0 = ok, TOKANNEAL_E_*), no exceptions across the ABI, caller owns
 * all buffers, a handle is used by one thread at a time (tokanneal_generate runs its own
 * workers). */
#ifndef TOKANNEAL_H
#define TOKANNEAL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOKANNEAL_OK 0
#define TOKANNEAL_E_ARG 1   /* null / out-of-range argument */
#define TOKANNEAL_E_IO 2    /* file cannot be written */
#define TOKANNEAL_E_DATA 3  /* arrays do not describe a candidate DAG */
#define TOKANNEAL_E_SPACE 4 /* output buffer too small; the needed sizes are returned */
#define TOKANNEAL_NONE 0xFFFFFFFFu /* "no parent" (Token.cpp:187-190) */

int tokanneal_abi_version(void);
const char *tokanneal_strerror(int status);

/* Candidate set = std::vector<annealing::Token> -------------------------------------------- */
typedef struct tokanneal_cands tokanneal_cands;

/* GetTokens(metadata, max_len, ..., rebuild = true) without the file walk: FileCandidates +
 * Trie::BuildTokens (GetTokens.cpp:21-27,31-90, Trie.cpp:114-152) over documents packed back
 * to back (document d = bytes[off[d] .. off[d+1]), the form tokscore_set_corpus takes).
 * Every window of 1 .. max_cp code points at every code-point position, clipped at the
 * document end; tokens come out in the reference's order (trie pre-order, children by
 * code point). */
int tokanneal_cands_build(tokanneal_cands **out, const uint8_t *bytes, const uint64_t *off, uint64_t n_docs,
                          uint32_t max_cp);

/* From arrays: chr = FIRST code point of the string (Trie.cpp:81-89), uses = its count,
 * lpar / rpar = index of the string minus its first / last code point, TOKANNEAL_NONE for
 * single code points (Trie.cpp:92-103).  Arrays are copied and checked. */
int tokanneal_cands_from_arrays(tokanneal_cands **out, uint64_t n, const uint32_t *chr, const uint64_t *uses,
                                const uint32_t *lpar, const uint32_t *rpar);

/* The candidate set of two corpora together from the set of each: the strings of both, each
 * once, uses added up, numbered again in the reference's order.  It is what GetTokens builds
 * when FileCandidates feeds several files into one Trie (GetTokens.cpp:31-90, Trie.cpp:114-152),
 * for callers that extract shard by shard (one GPU extraction holds < 4 GiB) or GPU by GPU.
 * Both inputs have to be in the builder's order (TOKANNEAL_E_DATA otherwise); they stay valid. */
int tokanneal_cands_merge(tokanneal_cands **out, const tokanneal_cands *a, const tokanneal_cands *b);

/* ReadTokens (Token.cpp:109-161) on a `.candidates-*.bin`.  *read_code receives the
 * reference's TokenReadErrCode (Token.h:13-19: 0 OK, 1 BAD_STREAM, 2 INVALID_VERSION,
 * 3 BAD_FILE, 4 NOT_DONE); a handle is returned only for OK.  Unlike the reference the
 * varint is read at full 64-bit width (its `(byte & 0x7F) << sft` is an int shift,
 * Token.cpp:136). */
int tokanneal_cands_read(tokanneal_cands **out, const char *path, int *read_code);

/* WriteTokens (Token.cpp:163-193), byte for byte. */
int tokanneal_cands_write(const tokanneal_cands *c, const char *path);

uint64_t tokanneal_cands_size(const tokanneal_cands *c);
/* Copies the four arrays out (any pointer may be NULL); each has room for size() entries. */
int tokanneal_cands_arrays(const tokanneal_cands *c, uint32_t *chr, uint64_t *uses, uint32_t *lpar, uint32_t *rpar);
/* Token::size() (Token.cpp:56-64), in code points. */
uint64_t tokanneal_cands_token_size(const tokanneal_cands *c, uint64_t i);
/* Token::GetName() (Token.cpp:65-74) as UTF-8, no terminator; *len receives the length
 * (TOKANNEAL_E_SPACE when cap is too small). */
int tokanneal_cands_name(const tokanneal_cands *c, uint64_t i, uint8_t *buf, uint64_t cap, uint64_t *len);
/* All names, packed like a vocabulary: bytes back to back + n+1 offsets. */
int tokanneal_cands_names(const tokanneal_cands *c, uint8_t *bytes, uint64_t bytes_cap, uint64_t *off,
                          uint64_t *bytes_needed);
void tokanneal_cands_free(tokanneal_cands *c);

/* TokenGenerator --------------------------------------------------------------------------- */
typedef struct tokanneal_gen tokanneal_gen;

typedef struct {
	uint64_t gen_cnt;     /* moves proposed so far (gen_cnt_, TokenGenerator.cpp:157) */
	uint64_t enabled_cnt; /* enabled candidates, roots not counted */
	uint64_t tot_cand;    /* candidates of more than one code point */
	double raw_score;
	double score;         /* CalcScore(raw_score, enabled_cnt), TokenGenerator.cpp:64-69 */
	double temp;
	double moment1, moment2, beta, sigma; /* the LomaxDist fit */
} tokanneal_state;

/* TokenGenerator(std::move(tokens), pref_token_count, batch_size) (TokenGenerator.cpp:17-49).
 * Takes the candidate set over: *cands is freed and set to NULL, as the moved-from vector. */
int tokanneal_gen_create(tokanneal_gen **out, tokanneal_cands **cands, uint64_t pref_token_count,
                         uint64_t batch_size);
void tokanneal_gen_free(tokanneal_gen *g);

/* Token::SimulateStep() const / Token::ApplyStep<Enable>() of token i: the move's absolute
 * change of the raw score, as a double (Token.cpp:83-105). */
int tokanneal_simulate_step(const tokanneal_gen *g, uint64_t i, double *delta);
int tokanneal_apply_step(tokanneal_gen *g, uint64_t i, int enable, double *delta);
/* uses on the l / r side and the enabled flag of token i as they stand. */
int tokanneal_token_state(const tokanneal_gen *g, uint64_t i, uint64_t *l_uses, uint64_t *r_uses, int *enabled);

/* Moves judged on the corpus instead of on the DAG: when `evaluate` is set it replaces the estimate
 * TokenGenerator::RunBatch takes from Token::SimulateStep (TokenGenerator.cpp:87) for all the moves
 * of a batch at once -- delta_raw[k] = what candidate cand[k] is worth in the raw score's unit
 * (tokens the corpus is shorter by with it enabled), each judged alone against the state as it
 * stands; tokscore_score_toggles (include/tokscore.h) computes exactly that on the GPU.  `applied`
 * is told every accepted move, `resync` is asked for the exact raw score after every batch (both
 * optional).  The DAG keeps following the accepted moves (GetSolution orders by it).  Callbacks
 * return 0 on success; a failure surfaces as TOKANNEAL_E_DATA from tokanneal_worker_task /
 * tokanneal_generate.  With an evaluator tokanneal_generate runs one worker.  evaluate = NULL
 * restores the reference's DAG estimate. */
typedef int (*tokanneal_evaluate_fn)(void *user, const uint32_t *cand, uint64_t n, int enable, double *delta_raw);
typedef int (*tokanneal_applied_fn)(void *user, uint32_t cand, int enable);
typedef int (*tokanneal_resync_fn)(void *user, double *raw_score);
int tokanneal_set_evaluator(tokanneal_gen *g, tokanneal_evaluate_fn evaluate, tokanneal_applied_fn applied,
                            tokanneal_resync_fn resync, void *user);

/* The deterministic seeding option: (re)seeds the generator tokanneal_worker_task draws from
 * (std::mt19937, as TokenGenerator.cpp:51-53 but with a given seed instead of random_device). */
int tokanneal_seed(tokanneal_gen *g, uint32_t seed);
/* One TokenGenerator::WorkerTask (TokenGenerator.cpp:107-158) on the calling thread. */
int tokanneal_worker_task(tokanneal_gen *g);

/* TokenGenerator::Generate(pass_cnt) (TokenGenerator.cpp:174-188): passes 0 .. pass_cnt
 * inclusive, ceil(tot_cand / batch) WorkerTasks each.  threads = 0: as many as the reference
 * would start.  seed < 0: from std::random_device like the reference; otherwise worker k is
 * seeded with seed + k, and a run with threads = 1 is reproducible.  per_pass (may be NULL)
 * receives the state after every pass (what the reference prints, :184-185). */
int tokanneal_generate(tokanneal_gen *g, uint64_t pass_cnt, uint32_t threads, int64_t seed,
                       tokanneal_state *per_pass, uint64_t per_pass_cap, uint64_t *n_passes);

int tokanneal_get_state(const tokanneal_gen *g, tokanneal_state *s);

/* TokenGenerator::GetSolution() (TokenGenerator.cpp:190-217), packed the way
 * tokscore_set_vocab takes a vocabulary: token i = bytes[off[i] .. off[i+1]).  Call with
 * bytes = NULL to learn the sizes. */
int tokanneal_solution(const tokanneal_gen *g, uint8_t *bytes, uint64_t bytes_cap, uint64_t *off, uint64_t off_cap,
                       uint64_t *n_tokens, uint64_t *bytes_needed);

#ifdef __cplusplus
}
#endif
#endif
