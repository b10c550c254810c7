// annealer.h -- the reference's simulated annealing over the candidate DAG, restated on the
// host:
//   LomaxDist       src/tokenizer/LomaxDist.{h,cpp}
//   TokenGenerator  src/tokenizer/TokenGenerator.{h,cpp}; the move primitives
//                   Token::SimulateStep / Token::ApplyStep<bool> (Token.cpp:17-37,83-105)
// on CandidateSet's arrays instead of 48-byte Token objects with pointers and striped mutexes:
// parents are 32-bit indices, `uses` are updated with atomic adds, `enabled` is a byte.
//
// Same arithmetic in the same order as the reference, and the same standard-library random
// number generator and distributions in the same call order, so that with one worker and a
// given seed a run is reproducible step by step (the reference seeds from std::random_device).
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "candidates.h"

namespace tokanneal {

class LomaxDist { // LomaxDist.h:4-31
	double alpha_ = 1;
	double moment1_ = 0;
	double moment2_ = 0;
	std::atomic<double> beta_{1};
	std::atomic<double> sigma_{1};

public:
	void SetHalfLife(double half_life);
	void SetMoments(double moment1, double moment2);
	void AddPoint(double val, double weight);
	void UpdateParams();
	void GetParams(double *beta, double *sigma) const;
	double GetMean() const { return moment1_; }
	double GetVar() const { return moment2_ - (moment1_ * moment1_); }
	double GetMoment2() const { return moment2_; }
	double GetPDF(double x) const;
	double GetCDF(double x) const;
	double GetBest(double p) const;
};

// thread_local gen / chance of TokenGenerator.cpp:51-53, one per worker
struct WorkerRng {
	std::mt19937 gen;
	std::uniform_real_distribution<> chance{0, 1};
	explicit WorkerRng(uint32_t seed) : gen(seed) {}
};

struct GeneratorState { // what Generate prints after every pass (TokenGenerator.cpp:184-185), and the fit
	uint64_t gen_cnt;
	uint64_t enabled_cnt;
	uint64_t tot_cand;
	double raw_score;
	double score;
	double temp;
	double moment1, moment2, beta, sigma;
};

// What judges a move instead of Token::SimulateStep's estimate on the DAG (TokenGenerator.cpp:87)
// when one is set: the corpus itself (tokscore_score_toggles).  Values are in the raw score's own
// unit, tokens the corpus is shorter by with the candidate enabled.
struct MoveEvaluator {
	// delta[k] >= 0 mostly: what candidate cand[k] is worth -- tokens saved by enabling it
	// (enable) or lost by disabling it -- each judged alone against the state as it stands
	std::function<bool(const uint32_t *cand, size_t n, bool enable, double *delta)> evaluate;
	// an accepted move
	std::function<bool(uint32_t cand, bool enable)> applied;
	// after a batch: the exact raw score of the state as it now is (the moves of a batch were
	// judged independently, so the sum of their deltas is not it)
	std::function<bool(double *raw_score)> resync;
};

class TokenGenerator {
public:
	// TokenGenerator(std::vector<Token>&&, pref_token_count, batch_size) (TokenGenerator.cpp:17-49)
	TokenGenerator(CandidateSet &&tokens, size_t pref_token_count, size_t batch_size);

	size_t size() const { return c_.size(); }
	const CandidateSet &tokens() const { return c_; }
	bool enabled(size_t i) const { return enabled_[i] != 0; }
	uint64_t l_uses(size_t i) const { return luses_[i]; }
	uint64_t r_uses(size_t i) const { return ruses_[i]; }

	// Token::SimulateStep (Token.cpp:83-88) / Token::ApplyStep<Enable> (Token.cpp:90-105)
	double SimulateStep(size_t i) const;
	double ApplyStep(size_t i, bool enable);

	// One WorkerTask (TokenGenerator.cpp:107-158) on the calling thread.
	void WorkerTask(WorkerRng &rng);

	// Moves judged by `e` from now on (an empty one: by the DAG again).  The DAG is still kept up
	// to date by every accepted move (GetSolution orders by it).  One worker thread only; a
	// callback that fails makes WorkerTask / Generate throw std::runtime_error.
	void SetEvaluator(MoveEvaluator e) { eval_ = std::move(e); }
	bool has_evaluator() const { return (bool)eval_.evaluate; }

	// Generate (TokenGenerator.cpp:174-188): passes 0 .. pass_cnt INCLUSIVE (`pass <= pass_cnt`),
	// ceil(tot_cand / batch) WorkerTasks each, on `threads` workers (0 = as the reference:
	// min(hardware threads, tot_cand / batch)).  seed < 0: workers seed from std::random_device
	// as the reference does; otherwise worker k uses seed + k, and threads == 1 is reproducible.
	// after_pass (optional) sees the state after every pass; returning false stops the run (the
	// reference polls stdin for that, TokenGenerator.cpp:186).
	void Generate(size_t pass_cnt, unsigned threads, int64_t seed,
	              const std::function<bool(const GeneratorState &)> &after_pass = nullptr);

	// GetSolution (TokenGenerator.cpp:190-217): enabled candidates by (size_t)SimulateStep
	// descending, name ascending; then the single-code-point roots in token order.  (The
	// reference also drops a "solution.txt" into the working directory: not reproduced.)
	std::vector<std::string> GetSolution() const;

	GeneratorState State() const;
	double CalcScore(double raw_score, size_t enabled_cnt) const; // TokenGenerator.cpp:64-69

private:
	template <bool Left> double BranchSimulate(size_t i) const;
	template <bool Enable, bool Left> double BranchApply(size_t i, uint64_t saved_uses);
	template <bool Enable> size_t RunBatch(size_t work_cnt, double *samples, WorkerRng &rng, std::vector<uint32_t> *working);

	CandidateSet c_;
	std::vector<uint64_t> luses_, ruses_; // Branch::uses (Token.h:41), both seeded with the count (Token.cpp:45-48)
	std::vector<uint8_t> enabled_;        // Token::enabled_

	size_t tot_cand_ = 0;
	const size_t pref_cand_;
	const size_t batch_size_;

	std::vector<uint32_t> roots_;
	std::mutex enabled_mutex_;
	std::vector<uint32_t> enabled_list_;
	std::mutex disabled_mutex_;
	std::vector<uint32_t> disabled_list_;

	// read by every move of every worker, written by the accepted ones: a cache line of their own
	alignas(64) std::atomic<size_t> enabled_cnt_{0};
	std::atomic<double> raw_score_{0};
	std::atomic<double> temp_{0};
	alignas(64) char pad_[1] = {0};

	std::mutex dist_mutex_;
	LomaxDist score_dist_;
	MoveEvaluator eval_;

	std::atomic<size_t> gen_cnt_{0};
};

} // namespace tokanneal
