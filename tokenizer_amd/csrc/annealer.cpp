// annealer.cpp -- see annealer.h.  Host code, no GPU involved.
#include "annealer.h"

#include <stdexcept>

#include <algorithm>
#include <cmath>
#include <thread>

namespace tokanneal {

// ---------------------------------------------------------------- LomaxDist (LomaxDist.cpp:5-51)

void LomaxDist::SetHalfLife(double half_life) { alpha_ = std::log(2) / half_life; }

void LomaxDist::SetMoments(double moment1, double moment2) {
	moment1_ = moment1;
	moment2_ = moment2;
}

void LomaxDist::AddPoint(double val, double weight) {
	// exponential moving first and second moment, step = weight * alpha
	const double w = weight * alpha_;
	moment1_ += (val - moment1_) * w;
	const double sq = val * val;
	moment2_ += (sq - moment2_) * w;
}

void LomaxDist::UpdateParams() {
	// method of moments for a Lomax distribution
	const double ratio = moment2_ / (moment2_ - 2 * moment1_ * moment1_);
	sigma_ = moment1_ * ratio;
	beta_ = ratio + 1;
}

void LomaxDist::GetParams(double *beta, double *sigma) const {
	if (beta) *beta = beta_;
	if (sigma) *sigma = sigma_;
}

double LomaxDist::GetPDF(double x) const { return beta_ / sigma_ * std::pow(1 + x / sigma_, -(beta_ + 1)); }
double LomaxDist::GetCDF(double x) const { return 1 - std::pow(1 + x / sigma_, -beta_); }

double LomaxDist::GetBest(double p) const {
	// integral of the inverse CDF over the best p of the mass
	const double e = 1.0 - 1 / beta_;
	return sigma_ * (std::pow(p, e) / e - p);
}

// ---------------------------------------------------------------- move primitives

namespace {
inline std::atomic_ref<uint64_t> A64(uint64_t &x) { return std::atomic_ref<uint64_t>(x); }
inline std::atomic_ref<uint8_t> A8(uint8_t &x) { return std::atomic_ref<uint8_t>(x); }
} // namespace

// Branch::SimulateStep<LeftBranch> (Token.cpp:17-22): 1 + the disabled ancestors on this side's
// chain before the first enabled one, times this token's uses on that side.  The product is
// taken in 64-bit integers and converted once, as `delta_len * uses` is there.
template <bool Left> double TokenGenerator::BranchSimulate(size_t i) const {
	const std::vector<uint32_t> &par = Left ? c_.lpar : c_.rpar;
	uint64_t len = 1;
	for (uint32_t a = par[i]; !A8(const_cast<uint8_t &>(enabled_[a])).load(std::memory_order_relaxed); a = par[a]) len++;
	const uint64_t uses = A64(const_cast<uint64_t &>((Left ? luses_ : ruses_)[i])).load(std::memory_order_relaxed);
	return (double)(len * uses);
}

double TokenGenerator::SimulateStep(size_t i) const {
	double score = 0;
	score += BranchSimulate<true>(i);
	score += BranchSimulate<false>(i);
	return score;
}

// Branch::ApplyStep<Enable, LeftBranch> (Token.cpp:26-37): the saved uses move off (enable) or
// back onto (disable) every ancestor up to and including the first enabled one.  The reference
// takes the ancestor's striped mutex around the update and the enabled test; here the update is
// one atomic add and the flag an atomic load after it.
template <bool Enable, bool Left> double TokenGenerator::BranchApply(size_t i, uint64_t saved) {
	const std::vector<uint32_t> &par = Left ? c_.lpar : c_.rpar;
	std::vector<uint64_t> &uses = Left ? luses_ : ruses_;
	uint64_t len = 1;
	for (uint32_t a = par[i];; a = par[a]) {
		if (Enable) A64(uses[a]).fetch_sub(saved, std::memory_order_acq_rel);
		else A64(uses[a]).fetch_add(saved, std::memory_order_acq_rel);
		if (A8(enabled_[a]).load(std::memory_order_acquire)) break;
		len++;
	}
	return (double)(len * saved);
}

double TokenGenerator::ApplyStep(size_t i, bool enable) {
	// Token::ApplyStep<Enable> (Token.cpp:90-105): flag first, then both sides with the uses
	// as they were at that moment
	A8(enabled_[i]).store(enable ? 1 : 0, std::memory_order_release);
	const uint64_t l = A64(luses_[i]).load(std::memory_order_acquire);
	const uint64_t r = A64(ruses_[i]).load(std::memory_order_acquire);
	double score = 0;
	if (enable) {
		score += BranchApply<true, true>(i, l);
		score += BranchApply<true, false>(i, r);
	} else {
		score += BranchApply<false, true>(i, l);
		score += BranchApply<false, false>(i, r);
	}
	return score;
}

// ---------------------------------------------------------------- TokenGenerator

TokenGenerator::TokenGenerator(CandidateSet &&tokens, size_t pref_token_count, size_t batch_size)
    : c_(std::move(tokens)), pref_cand_(pref_token_count), batch_size_(batch_size) {
	const size_t n = c_.size();
	luses_ = c_.uses;
	ruses_ = c_.uses;
	enabled_.assign(n, 0);
	// single code points are always on and never move; everything else starts disabled and feeds
	// the two moments of uses * (size - 1) (TokenGenerator.cpp:27-48)
	double moment1 = 0, moment2 = 0;
	for (size_t i = 0; i < n; i++) {
		if (c_.rpar[i] == kNone) { // size() == 1
			roots_.push_back((uint32_t)i);
			enabled_[i] = 1;
			continue;
		}
		disabled_list_.push_back((uint32_t)i);
		const size_t size = c_.TokenSize(i);
		double v = (double)(uint64_t)(c_.uses[i] * (uint64_t)(size - 1));
		moment1 += v;
		v *= v;
		moment2 += v;
	}
	tot_cand_ = disabled_list_.size();
	score_dist_.SetMoments(moment1 / tot_cand_, moment2 / tot_cand_);
	score_dist_.SetHalfLife((double)tot_cand_ * 0.5);
}

double TokenGenerator::CalcScore(double raw_score, size_t enabled_cnt) const {
	if (enabled_cnt == 0) return 0;
	const double contrib = tot_cand_ * score_dist_.GetBest((double)enabled_cnt / tot_cand_);
	const double fill = (double)enabled_cnt / pref_cand_;
	return raw_score / contrib * fill * (2 - fill);
}

// RunBatch<Enable> (TokenGenerator.cpp:71-105): draw up to work_cnt candidates from the list the
// move leaves, decide each move against the score as it stands, apply the accepted ones.
// Returns how many were drawn (the reference returns the vector).
template <bool Enable>
size_t TokenGenerator::RunBatch(size_t work_cnt, double *samples, WorkerRng &rng, std::vector<uint32_t> *working) {
	const size_t first = working->size();
	{
		std::vector<uint32_t> &from = Enable ? disabled_list_ : enabled_list_;
		std::lock_guard<std::mutex> lock(Enable ? disabled_mutex_ : enabled_mutex_);
		// RandCandidate (TokenGenerator.cpp:55-62), work_cnt times: uniform pick, swapped to the
		// back and popped.  The list shrinks by one per pick whatever is picked, so all positions
		// are drawn first (same draws, same order) and their cache lines requested together:
		// the lock is held for one memory round trip instead of one per pick.
		const size_t take = std::min(work_cnt, from.size());
		size_t pos[64];
		for (size_t base = 0; base < take; base += 64) {
			const size_t m = std::min<size_t>(64, take - base);
			for (size_t k = 0; k < m; k++) {
				pos[k] = std::uniform_int_distribution<size_t>(0, from.size() - 1 - k)(rng.gen);
				__builtin_prefetch(&from[pos[k]], 1);
			}
			for (size_t k = 0; k < m; k++) {
				std::swap(from[pos[k]], from.back());
				working->push_back(from.back());
				from.pop_back();
			}
		}
	}
	// the batch is known up front and its tokens are scattered over a DAG far larger than any cache:
	// start the loads of every candidate's own record, then of its two parents', before the moves
	for (size_t k = first; k < working->size(); k++) {
		const uint32_t cand = (*working)[k];
		__builtin_prefetch(&c_.lpar[cand]);
		__builtin_prefetch(&c_.rpar[cand]);
		__builtin_prefetch(&luses_[cand]);
		__builtin_prefetch(&ruses_[cand]);
	}
	for (size_t k = first; k < working->size(); k++) {
		const uint32_t cand = (*working)[k];
		const uint32_t l = c_.lpar[cand], r = c_.rpar[cand];
		__builtin_prefetch(&enabled_[l]);
		__builtin_prefetch(&enabled_[r]);
		__builtin_prefetch(&luses_[l], 1);
		__builtin_prefetch(&ruses_[r], 1);
		__builtin_prefetch(&c_.lpar[l]);
		__builtin_prefetch(&c_.rpar[r]);
	}
	// with an evaluator the whole batch is judged in one call, every move against the state as it
	// stands now (the reference's workers also judge against a state other moves are changing)
	std::vector<double> judged;
	if (eval_.evaluate && working->size() > first) {
		judged.resize(working->size() - first);
		if (!eval_.evaluate(working->data() + first, judged.size(), Enable, judged.data()))
			throw std::runtime_error("move evaluator failed");
	}
	for (size_t k = first; k < working->size(); k++) {
		const uint32_t cand = (*working)[k];
		const size_t cnt = enabled_cnt_;
		const double raw = raw_score_;
		const double score = CalcScore(raw, cnt);

		const double delta = judged.empty() ? SimulateStep(cand) : judged[k - first];
		const double new_score = CalcScore(raw + (Enable ? delta : -delta), cnt + (Enable ? 1 : -1));

		// accepted with the logistic probability of the score change at the current temperature
		const bool do_step = rng.chance(rng.gen) > 1 / (1 + std::exp((new_score - score) / temp_));
		if (do_step) {
			double applied = ApplyStep(cand, Enable);
			if (!judged.empty()) {
				applied = delta; // (the DAG only follows; WorkerTask puts the exact score in after the batch)
				if (eval_.applied && !eval_.applied(cand, Enable)) throw std::runtime_error("move evaluator failed");
			}
			raw_score_ += Enable ? applied : -applied;
			enabled_cnt_ += Enable ? 1 : -1;
		}
		*samples++ = delta;
	}
	return working->size() - first;
}

void TokenGenerator::WorkerTask(WorkerRng &rng) {
	// how many of this batch's moves enable: drawn so that the enabled count is pulled towards
	// pref_cand_ instead of half of the candidates (TokenGenerator.cpp:108-117); all in the
	// reference's unsigned 64-bit arithmetic
	const uint64_t cnt_now = enabled_cnt_;
	const uint64_t enabled_weight = cnt_now * (uint64_t)(tot_cand_ - pref_cand_);
	const uint64_t disabled_weight = (uint64_t)(tot_cand_ - cnt_now) * pref_cand_;
	const double total_weight = (double)(uint64_t)(enabled_weight + disabled_weight);
	const double corr_enable = total_weight / (double)(uint64_t)(tot_cand_ * pref_cand_);
	const double corr_disable = total_weight / (double)(uint64_t)(tot_cand_ * (tot_cand_ - pref_cand_));
	size_t enable_cnt = std::binomial_distribution<size_t>(batch_size_, (double)disabled_weight / total_weight)(rng.gen);
	if (enabled_cnt_ < batch_size_ - enable_cnt) enable_cnt = batch_size_ - enabled_cnt_;
	enable_cnt = std::min<size_t>(enable_cnt, tot_cand_ - enabled_cnt_);

	temp_ = 0.003 * std::exp(-(double)gen_cnt_ / tot_cand_ * 0.1); // TokenGenerator.cpp:119

	std::vector<uint32_t> working;
	std::vector<double> samples(batch_size_, 0.0); // (the reference's array is uninitialised)
	size_t drawn_on = 0, drawn_off = 0;
	if (enable_cnt > 0) drawn_on = RunBatch<true>(enable_cnt, samples.data(), rng, &working);
	if (enable_cnt < batch_size_) drawn_off = RunBatch<false>(batch_size_ - enable_cnt, samples.data() + enable_cnt, rng, &working);

	if (eval_.resync) {
		double raw = 0;
		if (!eval_.resync(&raw)) throw std::runtime_error("move evaluator failed");
		raw_score_ = raw;
	}
	{
		// the sampled deltas feed the fit of the score distribution (TokenGenerator.cpp:136-146);
		// a slot no candidate was drawn for is skipped (the reference would read stack garbage)
		std::lock_guard<std::mutex> lock(dist_mutex_);
		for (size_t k = 0; k < drawn_on; k++) score_dist_.AddPoint(samples[k], corr_enable);
		for (size_t k = 0; k < drawn_off; k++) score_dist_.AddPoint(samples[enable_cnt + k], corr_disable);
		score_dist_.UpdateParams();
	}

	// hand the candidates back to the list of the state they ended in (TokenGenerator.cpp:122-155)
	std::vector<uint32_t> on, off;
	for (uint32_t cand : working) (A8(enabled_[cand]).load(std::memory_order_relaxed) ? on : off).push_back(cand);
	if (!on.empty()) {
		std::lock_guard<std::mutex> lock(enabled_mutex_);
		enabled_list_.insert(enabled_list_.end(), on.begin(), on.end());
	}
	if (!off.empty()) {
		std::lock_guard<std::mutex> lock(disabled_mutex_);
		disabled_list_.insert(disabled_list_.end(), off.begin(), off.end());
	}
	gen_cnt_ += batch_size_;
}

GeneratorState TokenGenerator::State() const {
	GeneratorState s;
	s.gen_cnt = gen_cnt_;
	s.enabled_cnt = enabled_cnt_;
	s.tot_cand = tot_cand_;
	s.raw_score = raw_score_;
	s.score = CalcScore(s.raw_score, s.enabled_cnt);
	s.temp = temp_;
	s.moment1 = score_dist_.GetMean();
	s.moment2 = score_dist_.GetMoment2();
	score_dist_.GetParams(&s.beta, &s.sigma);
	return s;
}

void TokenGenerator::Generate(size_t pass_cnt, unsigned threads, int64_t seed,
                              const std::function<bool(const GeneratorState &)> &after_pass) {
	if (tot_cand_ == 0 || batch_size_ == 0) return;
	if (threads == 0)
		threads = (unsigned)std::min<uint64_t>(std::max(1u, std::thread::hardware_concurrency()), tot_cand_ / batch_size_);
	threads = std::max(1u, threads);
	if (eval_.evaluate) threads = 1; // one GPU context behind the evaluator
	const size_t tasks_per_pass = (tot_cand_ + batch_size_ - 1) / batch_size_; // `i * batch_size_ < tot_cand_`
	std::vector<WorkerRng> rngs;
	for (unsigned k = 0; k < threads; k++) rngs.emplace_back(seed < 0 ? std::random_device()() : (uint32_t)(seed + k));
	for (size_t pass = 0; pass <= pass_cnt; pass++) {
		if (threads == 1) {
			for (size_t t = 0; t < tasks_per_pass; t++) WorkerTask(rngs[0]);
		} else {
			std::atomic<size_t> next{0};
			std::vector<std::thread> pool;
			for (unsigned k = 0; k < threads; k++)
				pool.emplace_back([&, k] {
					while (next.fetch_add(1) < tasks_per_pass) WorkerTask(rngs[k]);
				});
			for (std::thread &t : pool) t.join();
		}
		if (after_pass && !after_pass(State())) break;
		if (pass == (size_t)-1) break; // pass_cnt == -1 means "until told to stop": never wraps
	}
}

std::vector<std::string> TokenGenerator::GetSolution() const {
	std::vector<std::pair<size_t, std::string>> order;
	order.reserve(enabled_list_.size());
	for (uint32_t cand : enabled_list_) order.emplace_back((size_t)SimulateStep(cand), c_.GetName(cand));
	std::sort(order.begin(), order.end(), [](const auto &x, const auto &y) {
		return x.first == y.first ? x.second < y.second : x.first > y.first;
	});
	std::vector<std::string> solution;
	solution.reserve(order.size() + roots_.size());
	for (auto &e : order) solution.emplace_back(std::move(e.second));
	for (uint32_t r : roots_) solution.emplace_back(c_.GetName(r));
	return solution;
}

} // namespace tokanneal
