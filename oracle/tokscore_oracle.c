/*
 * tokscore_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See tokscore_oracle.h for scope, citations and how parity is pinned.
 *
 * The algorithm is deliberately the reference's own: an exact-string hash map
 * probed longest-first at every position (SolutionFile.cpp:89-97), not a trie.
 */
#include "tokscore_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct ora_vocab {
	uint64_t V;
	uint64_t max_len;     /* SolutionFile.cpp:55 / :68 */
	const uint8_t *bytes; /* owned copy */
	uint64_t *off;        /* owned copy, V+1 */
	uint64_t cap;         /* power of two */
	uint64_t *slot_tok;   /* token index + 1, 0 = empty */
	uint64_t *slot_hash;
};

static uint64_t hash_bytes(const uint8_t *s, uint64_t n) {
	uint64_t h = 1469598103934665603ull; /* FNV-1a */
	for (uint64_t i = 0; i < n; i++) {
		h ^= s[i];
		h *= 1099511628211ull;
	}
	return h ^ (h >> 29);
}

ora_vocab *ora_vocab_new(const uint8_t *bytes, const uint64_t *off, uint64_t V) {
	ora_vocab *v = (ora_vocab *)calloc(1, sizeof *v);
	v->V = V;
	uint64_t nbytes = off[V] - off[0];
	uint8_t *b = (uint8_t *)malloc(nbytes + 1);
	memcpy(b, bytes + off[0], nbytes);
	v->bytes = b;
	v->off = (uint64_t *)malloc((V + 1) * sizeof(uint64_t));
	for (uint64_t i = 0; i <= V; i++) v->off[i] = off[i] - off[0];
	v->cap = 16;
	while (v->cap < 2 * V + 2) v->cap <<= 1;
	v->slot_tok = (uint64_t *)calloc(v->cap, sizeof(uint64_t));
	v->slot_hash = (uint64_t *)calloc(v->cap, sizeof(uint64_t));
	for (uint64_t i = 0; i < V; i++) {
		const uint8_t *s = v->bytes + v->off[i];
		uint64_t n = v->off[i + 1] - v->off[i];
		/* max_len_ is updated BEFORE the insert, duplicates included (:55). */
		if (n > v->max_len) v->max_len = n;
		uint64_t h = hash_bytes(s, n);
		uint64_t p = h & (v->cap - 1);
		int dup = 0;
		while (v->slot_tok[p]) {
			uint64_t j = v->slot_tok[p] - 1;
			uint64_t m = v->off[j + 1] - v->off[j];
			if (v->slot_hash[p] == h && m == n &&
			    memcmp(v->bytes + v->off[j], s, n) == 0) {
				dup = 1; /* insert fails: first id is kept (:56) */
				break;
			}
			p = (p + 1) & (v->cap - 1);
		}
		if (!dup) {
			v->slot_tok[p] = i + 1;
			v->slot_hash[p] = h;
		}
	}
	return v;
}

void ora_vocab_free(ora_vocab *v) {
	if (!v) return;
	free((void *)v->bytes);
	free(v->off);
	free(v->slot_tok);
	free(v->slot_hash);
	free(v);
}

uint64_t ora_vocab_size(const ora_vocab *v) { return v->V; }
uint64_t ora_vocab_max_len(const ora_vocab *v) { return v->max_len; }

uint64_t ora_get_id(const ora_vocab *v, const uint8_t *s, uint64_t n) {
	uint64_t h = hash_bytes(s, n);
	uint64_t p = h & (v->cap - 1);
	while (v->slot_tok[p]) {
		uint64_t j = v->slot_tok[p] - 1;
		if (v->slot_hash[p] == h && v->off[j + 1] - v->off[j] == n &&
		    memcmp(v->bytes + v->off[j], s, n) == 0)
			return j + 2; /* token i -> id i+2 (:49-57) */
		p = (p + 1) & (v->cap - 1);
	}
	return ORA_UNKNOWN;
}

/* std::tolower on unsigned char in the "C" locale: only A-Z change (:87-88). */
static inline uint8_t c_tolower(uint8_t c) {
	return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c;
}

/* Tokenize body on an already lower-cased buffer; emit(id) per token. */
static uint64_t tokenize_lowered(const ora_vocab *v, const uint8_t *in,
                                 uint64_t n, uint64_t *ids_out,
                                 uint64_t *counts) {
	uint64_t k = 0;
	uint64_t pos = 0;
	while (pos < n) { /* :89 */
		uint64_t len = v->max_len < n - pos ? v->max_len : n - pos; /* :90 */
		for (; len > 0; --len) {
			uint64_t id = ora_get_id(v, in + pos, len); /* :91 */
			if (id == ORA_UNKNOWN && len > 1) continue; /* :92 */
			if (ids_out) ids_out[k] = id;               /* :93 */
			if (counts) counts[id == ORA_UNKNOWN ? v->V + 2 : id]++;
			k++;
			pos += len; /* :94 */
			break;
		}
		/* max_len_ == 0 (empty vocabulary or only "" tokens) makes the
		 * reference loop forever at :89-97; the oracle refuses instead. */
		if (len == 0) abort();
	}
	return k;
}

uint64_t ora_tokenize(const ora_vocab *v, const uint8_t *text, uint64_t n,
                      uint64_t *ids_out) {
	uint8_t *in = (uint8_t *)malloc(n + 1);
	for (uint64_t i = 0; i < n; i++) in[i] = c_tolower(text[i]);
	uint64_t k = 0;
	ids_out[k++] = 0; /* <START>, :85 */
	k += tokenize_lowered(v, in, n, ids_out + k, NULL);
	ids_out[k++] = 1; /* <END>, :98 */
	free(in);
	return k;
}

typedef struct {
	const ora_vocab *v;
	const uint8_t *corpus;
	const uint64_t *off;
	uint64_t d0, d1;
	uint64_t *counts;
	uint64_t tokens, bytes;
} score_job;

static void *score_range(void *arg) {
	score_job *j = (score_job *)arg;
	uint64_t maxn = 0;
	for (uint64_t d = j->d0; d < j->d1; d++) {
		uint64_t n = j->off[d + 1] - j->off[d];
		if (n > maxn) maxn = n;
	}
	uint8_t *in = (uint8_t *)malloc(maxn + 1);
	for (uint64_t d = j->d0; d < j->d1; d++) {
		const uint8_t *text = j->corpus + j->off[d];
		uint64_t n = j->off[d + 1] - j->off[d];
		for (uint64_t i = 0; i < n; i++) in[i] = c_tolower(text[i]);
		j->bytes += n;                                          /* main.cpp:40 */
		j->tokens += tokenize_lowered(j->v, in, n, NULL, j->counts); /* :41 */
		j->counts[0]++; /* <START> */
		j->counts[1]++; /* <END> */
	}
	free(in);
	return NULL;
}

void ora_score(const ora_vocab *v, const uint8_t *corpus, const uint64_t *off,
               uint64_t D, uint64_t *counts, uint64_t *total_tokens,
               uint64_t *total_bytes, int threads) {
	uint64_t nc = v->V + 3;
	memset(counts, 0, nc * sizeof(uint64_t));
	if (threads < 1) threads = 1;
	if ((uint64_t)threads > D) threads = D ? (int)D : 1;
	score_job *jobs = (score_job *)calloc(threads, sizeof *jobs);
	pthread_t *tid = (pthread_t *)calloc(threads, sizeof *tid);
	uint64_t total = D ? off[D] - off[0] : 0;
	uint64_t d = 0;
	for (int t = 0; t < threads; t++) {
		jobs[t].v = v;
		jobs[t].corpus = corpus;
		jobs[t].off = off;
		jobs[t].d0 = d;
		uint64_t target = off[0] + total / threads * (t + 1);
		if (t == threads - 1) d = D;
		else
			while (d < D && off[d + 1] <= target) d++;
		jobs[t].d1 = d;
		jobs[t].counts = t == 0 ? counts : (uint64_t *)calloc(nc, sizeof(uint64_t));
	}
	if (threads == 1) score_range(&jobs[0]);
	else {
		for (int t = 0; t < threads; t++)
			pthread_create(&tid[t], NULL, score_range, &jobs[t]);
		for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
	}
	uint64_t tok = 0, byt = 0;
	for (int t = 0; t < threads; t++) {
		tok += jobs[t].tokens;
		byt += jobs[t].bytes;
		if (t > 0) {
			for (uint64_t i = 0; i < nc; i++) counts[i] += jobs[t].counts[i];
			free(jobs[t].counts);
		}
	}
	*total_tokens = tok;
	*total_bytes = byt;
	free(jobs);
	free(tid);
}

/* ---- closed-vocabulary all-occurrence counting (GetTokens.cpp:21-27, Trie.cpp:114-128) ---- */

static int seq_len(uint8_t lead) { /* utf8/core.h:134-147 */
	if (lead < 0x80) return 1;
	if ((lead >> 5) == 0x6) return 2;
	if ((lead >> 4) == 0xe) return 3;
	if ((lead >> 3) == 0x1e) return 4;
	return 0;
}

typedef struct {
	const ora_vocab *v;
	const uint8_t *corpus;
	const uint64_t *off;
	uint64_t d0, d1;
	uint32_t max_cp;
	uint64_t *counts;
	uint64_t positions;
} count_job;

static void *count_range(void *arg) {
	count_job *j = (count_job *)arg;
	uint64_t cap = 0;
	for (uint64_t d = j->d0; d < j->d1; d++) {
		uint64_t n = j->off[d + 1] - j->off[d];
		if (n > cap) cap = n;
	}
	uint64_t *start = (uint64_t *)malloc((cap + 1) * sizeof(uint64_t));
	for (uint64_t d = j->d0; d < j->d1; d++) {
		const uint8_t *text = j->corpus + j->off[d];
		const uint64_t n = j->off[d + 1] - j->off[d];
		/* utf8to32: one code point per lead byte (GetTokens.cpp:22-23) */
		uint64_t ncp = 0, p = 0;
		while (p < n) {
			int l = seq_len(text[p]);
			if (l == 0) l = 1; /* unchecked::next: no case taken, ++it */
			start[ncp++] = p;
			p += (uint64_t)l;
		}
		if (p > n) p = n; /* (truncated tail: outside the contract) */
		start[ncp] = p;
		j->positions += ncp;
		for (uint64_t i = 0; i < ncp; i++) { /* GetTokens.cpp:24-26 */
			uint64_t len = ncp - i < j->max_cp ? ncp - i : j->max_cp;
			for (uint64_t k = 1; k <= len; k++) { /* Trie.cpp:118-124: every prefix gets ++freq */
				uint64_t id = ora_get_id(j->v, text + start[i], start[i + k] - start[i]);
				if (id != ORA_UNKNOWN) j->counts[id]++;
			}
		}
	}
	free(start);
	return NULL;
}

void ora_count_occurrences(const ora_vocab *v, const uint8_t *corpus, const uint64_t *off,
                           uint64_t D, uint32_t max_cp, uint64_t *counts, uint64_t *positions,
                           int threads) {
	uint64_t nc = v->V + 3;
	memset(counts, 0, nc * sizeof(uint64_t));
	if (threads < 1) threads = 1;
	if ((uint64_t)threads > D) threads = D ? (int)D : 1;
	count_job *jobs = (count_job *)calloc(threads, sizeof *jobs);
	pthread_t *tid = (pthread_t *)calloc(threads, sizeof *tid);
	uint64_t total = D ? off[D] - off[0] : 0;
	uint64_t d = 0;
	for (int t = 0; t < threads; t++) {
		jobs[t].v = v;
		jobs[t].corpus = corpus;
		jobs[t].off = off;
		jobs[t].max_cp = max_cp;
		jobs[t].d0 = d;
		uint64_t target = off[0] + total / threads * (t + 1);
		if (t == threads - 1) d = D;
		else
			while (d < D && off[d + 1] <= target) d++;
		jobs[t].d1 = d;
		jobs[t].counts = t == 0 ? counts : (uint64_t *)calloc(nc, sizeof(uint64_t));
	}
	if (threads == 1) count_range(&jobs[0]);
	else {
		for (int t = 0; t < threads; t++) pthread_create(&tid[t], NULL, count_range, &jobs[t]);
		for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
	}
	uint64_t pos = 0;
	for (int t = 0; t < threads; t++) {
		pos += jobs[t].positions;
		if (t) {
			for (uint64_t i = 0; i < nc; i++) counts[i] += jobs[t].counts[i];
			free(jobs[t].counts);
		}
	}
	if (positions) *positions = pos;
	free(jobs);
	free(tid);
}
