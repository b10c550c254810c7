// tokanneal.cpp -- the C ABI of include/tokanneal.h over candidates.{h,cpp} and annealer.{h,cpp}.
#include "../../include/tokanneal.h"

#include <cstring>
#include <memory>
#include <new>

#include "annealer.h"
#include "candidates.h"

using namespace tokanneal;

struct tokanneal_cands {
	CandidateSet set;
};

struct tokanneal_gen {
	std::unique_ptr<TokenGenerator> gen;
	WorkerRng rng{std::random_device()()};
};

namespace {

template <typename F> int guarded(F &&f) {
	try {
		return f();
	} catch (const std::bad_alloc &) {
		return TOKANNEAL_E_DATA;
	} catch (...) {
		return TOKANNEAL_E_DATA;
	}
}

void fill_state(const GeneratorState &s, tokanneal_state *o) {
	o->gen_cnt = s.gen_cnt, o->enabled_cnt = s.enabled_cnt, o->tot_cand = s.tot_cand;
	o->raw_score = s.raw_score, o->score = s.score, o->temp = s.temp;
	o->moment1 = s.moment1, o->moment2 = s.moment2, o->beta = s.beta, o->sigma = s.sigma;
}

int pack_strings(const std::vector<std::string> &v, uint8_t *bytes, uint64_t bytes_cap, uint64_t *off, uint64_t off_cap,
                 uint64_t *n, uint64_t *needed) {
	uint64_t total = 0;
	for (const std::string &s : v) total += s.size();
	if (n) *n = v.size();
	if (needed) *needed = total;
	if (!bytes && !off) return TOKANNEAL_OK; // size query
	if (!off || off_cap < v.size() + 1 || (total && (!bytes || bytes_cap < total))) return TOKANNEAL_E_SPACE;
	uint64_t at = 0;
	for (size_t i = 0; i < v.size(); i++) {
		off[i] = at;
		if (!v[i].empty()) memcpy(bytes + at, v[i].data(), v[i].size());
		at += v[i].size();
	}
	off[v.size()] = at;
	return TOKANNEAL_OK;
}

} // namespace

extern "C" {

int tokanneal_abi_version(void) { return 100; }

const char *tokanneal_strerror(int status) {
	switch (status) {
	case TOKANNEAL_OK: return "ok";
	case TOKANNEAL_E_ARG: return "null or out-of-range argument";
	case TOKANNEAL_E_IO: return "file cannot be written";
	case TOKANNEAL_E_DATA: return "arrays do not describe a candidate DAG";
	case TOKANNEAL_E_SPACE: return "output buffer too small";
	}
	return "unknown status";
}

int tokanneal_cands_build(tokanneal_cands **out, const uint8_t *bytes, const uint64_t *off, uint64_t n_docs,
                          uint32_t max_cp) {
	if (!out || !off || max_cp == 0 || (!bytes && n_docs && off[n_docs] != off[0])) return TOKANNEAL_E_ARG;
	for (uint64_t d = 0; d < n_docs; d++)
		if (off[d + 1] < off[d]) return TOKANNEAL_E_ARG;
	return guarded([&] {
		auto c = std::make_unique<tokanneal_cands>();
		BuildCandidates(bytes, off, n_docs, max_cp, &c->set);
		*out = c.release();
		return TOKANNEAL_OK;
	});
}

int tokanneal_cands_from_arrays(tokanneal_cands **out, uint64_t n, const uint32_t *chr, const uint64_t *uses,
                                const uint32_t *lpar, const uint32_t *rpar) {
	if (!out || (n && (!chr || !uses || !lpar || !rpar)) || n >= TOKANNEAL_NONE) return TOKANNEAL_E_ARG;
	return guarded([&] {
		auto c = std::make_unique<tokanneal_cands>();
		c->set.chr.assign(chr, chr + n);
		c->set.uses.assign(uses, uses + n);
		c->set.lpar.assign(lpar, lpar + n);
		c->set.rpar.assign(rpar, rpar + n);
		if (!c->set.Validate(nullptr)) return TOKANNEAL_E_DATA;
		// a token is a root on both sides or on neither (Trie.cpp:145-146)
		for (uint64_t i = 0; i < n; i++)
			if ((lpar[i] == kNone) != (rpar[i] == kNone)) return TOKANNEAL_E_DATA;
		*out = c.release();
		return TOKANNEAL_OK;
	});
}

int tokanneal_cands_merge(tokanneal_cands **out, const tokanneal_cands *a, const tokanneal_cands *b) {
	if (!out || !a || !b) return TOKANNEAL_E_ARG;
	return guarded([&] {
		auto c = std::make_unique<tokanneal_cands>();
		if (!MergeCandidates(a->set, b->set, &c->set, nullptr)) return TOKANNEAL_E_DATA;
		*out = c.release();
		return TOKANNEAL_OK;
	});
}

int tokanneal_cands_read(tokanneal_cands **out, const char *path, int *read_code) {
	if (!out || !path) return TOKANNEAL_E_ARG;
	return guarded([&] {
		auto c = std::make_unique<tokanneal_cands>();
		const ReadCode rc = ReadCandidates(path, &c->set);
		if (read_code) *read_code = (int)rc;
		if (rc != READ_OK) return TOKANNEAL_E_DATA;
		*out = c.release();
		return TOKANNEAL_OK;
	});
}

int tokanneal_cands_write(const tokanneal_cands *c, const char *path) {
	if (!c || !path) return TOKANNEAL_E_ARG;
	return guarded([&] { return WriteCandidates(c->set, path, nullptr) ? TOKANNEAL_OK : TOKANNEAL_E_IO; });
}

uint64_t tokanneal_cands_size(const tokanneal_cands *c) { return c ? c->set.size() : 0; }

int tokanneal_cands_arrays(const tokanneal_cands *c, uint32_t *chr, uint64_t *uses, uint32_t *lpar, uint32_t *rpar) {
	if (!c) return TOKANNEAL_E_ARG;
	const size_t n = c->set.size();
	if (n == 0) return TOKANNEAL_OK;
	if (chr) memcpy(chr, c->set.chr.data(), n * sizeof *chr);
	if (uses) memcpy(uses, c->set.uses.data(), n * sizeof *uses);
	if (lpar) memcpy(lpar, c->set.lpar.data(), n * sizeof *lpar);
	if (rpar) memcpy(rpar, c->set.rpar.data(), n * sizeof *rpar);
	return TOKANNEAL_OK;
}

uint64_t tokanneal_cands_token_size(const tokanneal_cands *c, uint64_t i) {
	return c && i < c->set.size() ? c->set.TokenSize(i) : 0;
}

int tokanneal_cands_name(const tokanneal_cands *c, uint64_t i, uint8_t *buf, uint64_t cap, uint64_t *len) {
	if (!c || !len || i >= c->set.size()) return TOKANNEAL_E_ARG;
	return guarded([&] {
		const std::string s = c->set.GetName(i);
		*len = s.size();
		if (!buf || cap < s.size()) return TOKANNEAL_E_SPACE;
		memcpy(buf, s.data(), s.size());
		return TOKANNEAL_OK;
	});
}

int tokanneal_cands_names(const tokanneal_cands *c, uint8_t *bytes, uint64_t bytes_cap, uint64_t *off,
                          uint64_t *bytes_needed) {
	if (!c) return TOKANNEAL_E_ARG;
	return guarded([&] {
		const size_t n = c->set.size();
		uint64_t total = 0;
		for (size_t i = 0; i < n; i++) total += c->set.GetName(i).size();
		if (bytes_needed) *bytes_needed = total;
		if (!bytes && !off) return TOKANNEAL_OK; // size query
		if (!off || (total && (!bytes || bytes_cap < total))) return TOKANNEAL_E_SPACE;
		uint64_t at = 0;
		for (size_t i = 0; i < n; i++) {
			off[i] = at;
			const std::string s = c->set.GetName(i);
			memcpy(bytes + at, s.data(), s.size());
			at += s.size();
		}
		off[n] = at;
		return TOKANNEAL_OK;
	});
}

void tokanneal_cands_free(tokanneal_cands *c) { delete c; }

int tokanneal_gen_create(tokanneal_gen **out, tokanneal_cands **cands, uint64_t pref_token_count, uint64_t batch_size) {
	if (!out || !cands || !*cands || batch_size == 0 || pref_token_count == 0) return TOKANNEAL_E_ARG;
	return guarded([&] {
		auto g = std::make_unique<tokanneal_gen>();
		g->gen = std::make_unique<TokenGenerator>(std::move((*cands)->set), pref_token_count, batch_size);
		delete *cands;
		*cands = nullptr;
		*out = g.release();
		return TOKANNEAL_OK;
	});
}

void tokanneal_gen_free(tokanneal_gen *g) { delete g; }

int tokanneal_simulate_step(const tokanneal_gen *g, uint64_t i, double *delta) {
	if (!g || !delta || i >= g->gen->size()) return TOKANNEAL_E_ARG;
	if (g->gen->tokens().rpar[i] == kNone) return TOKANNEAL_E_ARG; // roots have no parents to walk to
	*delta = g->gen->SimulateStep(i);
	return TOKANNEAL_OK;
}

int tokanneal_apply_step(tokanneal_gen *g, uint64_t i, int enable, double *delta) {
	if (!g || i >= g->gen->size()) return TOKANNEAL_E_ARG;
	if (g->gen->tokens().rpar[i] == kNone) return TOKANNEAL_E_ARG;
	const double d = g->gen->ApplyStep(i, enable != 0);
	if (delta) *delta = d;
	return TOKANNEAL_OK;
}

int tokanneal_token_state(const tokanneal_gen *g, uint64_t i, uint64_t *l_uses, uint64_t *r_uses, int *enabled) {
	if (!g || i >= g->gen->size()) return TOKANNEAL_E_ARG;
	if (l_uses) *l_uses = g->gen->l_uses(i);
	if (r_uses) *r_uses = g->gen->r_uses(i);
	if (enabled) *enabled = g->gen->enabled(i) ? 1 : 0;
	return TOKANNEAL_OK;
}

int tokanneal_seed(tokanneal_gen *g, uint32_t seed) {
	if (!g) return TOKANNEAL_E_ARG;
	g->rng = WorkerRng(seed);
	return TOKANNEAL_OK;
}

int tokanneal_set_evaluator(tokanneal_gen *g, tokanneal_evaluate_fn evaluate, tokanneal_applied_fn applied,
                            tokanneal_resync_fn resync, void *user) {
	if (!g) return TOKANNEAL_E_ARG;
	MoveEvaluator e;
	if (evaluate) {
		e.evaluate = [=](const uint32_t *cand, size_t n, bool enable, double *delta) {
			return evaluate(user, cand, (uint64_t)n, enable ? 1 : 0, delta) == 0;
		};
		if (applied) e.applied = [=](uint32_t cand, bool enable) { return applied(user, cand, enable ? 1 : 0) == 0; };
		if (resync) e.resync = [=](double *raw) { return resync(user, raw) == 0; };
	}
	g->gen->SetEvaluator(std::move(e));
	return TOKANNEAL_OK;
}

int tokanneal_worker_task(tokanneal_gen *g) {
	if (!g) return TOKANNEAL_E_ARG;
	if (g->gen->State().tot_cand == 0) return TOKANNEAL_E_DATA;
	return guarded([&] {
		g->gen->WorkerTask(g->rng);
		return TOKANNEAL_OK;
	});
}

int tokanneal_generate(tokanneal_gen *g, uint64_t pass_cnt, uint32_t threads, int64_t seed, tokanneal_state *per_pass,
                       uint64_t per_pass_cap, uint64_t *n_passes) {
	if (!g || pass_cnt >= (1ull << 32)) return TOKANNEAL_E_ARG; // (no "until stdin" mode: there is no channel to stop it)
	return guarded([&] {
		uint64_t done = 0;
		g->gen->Generate(pass_cnt, threads, seed, [&](const GeneratorState &s) {
			if (per_pass && done < per_pass_cap) fill_state(s, &per_pass[done]);
			done++;
			return true;
		});
		if (n_passes) *n_passes = done;
		return TOKANNEAL_OK;
	});
}

int tokanneal_get_state(const tokanneal_gen *g, tokanneal_state *s) {
	if (!g || !s) return TOKANNEAL_E_ARG;
	fill_state(g->gen->State(), s);
	return TOKANNEAL_OK;
}

int tokanneal_solution(const tokanneal_gen *g, uint8_t *bytes, uint64_t bytes_cap, uint64_t *off, uint64_t off_cap,
                       uint64_t *n_tokens, uint64_t *bytes_needed) {
	if (!g) return TOKANNEAL_E_ARG;
	return guarded([&] { return pack_strings(g->gen->GetSolution(), bytes, bytes_cap, off, off_cap, n_tokens, bytes_needed); });
}

} // extern "C"
