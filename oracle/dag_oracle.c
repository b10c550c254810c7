/*
 * dag_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See dag_oracle.h for scope and citations.
 */
#include "dag_oracle.h"

#include <math.h>

/* Token.cpp:17-22.  delta_len (int64) * uses (uint64) is evaluated in
 * uint64 and only then converted to double, exactly as the C++ does. */
static double branch_simulate(const ora_dag *g, int64_t i, int left) {
	int64_t delta_len = 1;
	const int64_t *par = left ? g->lpar : g->rpar;
	for (int64_t node = par[i]; !g->enabled[node]; node = par[node]) delta_len++;
	uint64_t uses = left ? g->luses[i] : g->ruses[i];
	return (double)((uint64_t)delta_len * uses);
}

/* Token.cpp:83-88 */
double ora_dag_simulate(const ora_dag *g, int64_t i) {
	double score = 0;
	score += branch_simulate(g, i, 1);
	score += branch_simulate(g, i, 0);
	return score;
}

/* Token.cpp:26-37: every ancestor up to AND INCLUDING the first enabled one
 * has its same-side uses adjusted by the saved snapshot. */
static double branch_apply(ora_dag *g, int64_t i, int enable, int left, uint64_t saved) {
	int64_t delta_len = 1;
	const int64_t *par = left ? g->lpar : g->rpar;
	uint64_t *uses = left ? g->luses : g->ruses;
	for (int64_t node = par[i];; node = par[node]) {
		uses[node] -= (uint64_t)(enable ? 1 : -1) * saved; /* :32, wraps mod 2^64 */
		if (g->enabled[node]) break;
		delta_len++;
	}
	return (double)((uint64_t)delta_len * saved);
}

/* Token.cpp:90-105 */
double ora_dag_apply(ora_dag *g, int64_t i, int enable) {
	g->enabled[i] = enable ? 1 : 0;
	uint64_t l = g->luses[i];
	uint64_t r = g->ruses[i];
	double score = 0;
	score += branch_apply(g, i, enable, 1, l);
	score += branch_apply(g, i, enable, 0, r);
	return score;
}

/* LomaxDist.h:5-12 defaults */
void ora_lomax_init(ora_lomax *d) {
	d->alpha = 1;
	d->moment1 = 0;
	d->moment2 = 0;
	d->beta = 1;
	d->sigma = 1;
}
/* LomaxDist.cpp:5-7 */
void ora_lomax_set_half_life(ora_lomax *d, double half_life) { d->alpha = log(2) / half_life; }
/* LomaxDist.cpp:9-12 */
void ora_lomax_set_moments(ora_lomax *d, double m1, double m2) {
	d->moment1 = m1;
	d->moment2 = m2;
}
/* LomaxDist.cpp:14-19 */
void ora_lomax_add_point(ora_lomax *d, double val, double weight) {
	weight *= d->alpha;
	d->moment1 += (val - d->moment1) * weight;
	val *= val;
	d->moment2 += (val - d->moment2) * weight;
}
/* LomaxDist.cpp:21-25 */
void ora_lomax_update_params(ora_lomax *d) {
	const double temp = d->moment2 / (d->moment2 - 2 * d->moment1 * d->moment1);
	d->sigma = d->moment1 * temp;
	d->beta = temp + 1;
}
/* LomaxDist.cpp:48-51 */
double ora_lomax_get_best(const ora_lomax *d, double p) {
	const double temp = 1.0 - 1 / d->beta;
	return d->sigma * (pow(p, temp) / temp - p);
}

/* TokenGenerator.cpp:64-69 */
double ora_calc_score(const ora_lomax *d, double raw_score, uint64_t enabled_cnt,
                      uint64_t tot_cand, uint64_t pref_cand) {
	if (enabled_cnt == 0) return 0;
	const double contrib = tot_cand * ora_lomax_get_best(d, (double)enabled_cnt / tot_cand);
	const double new_pref_fill = (double)enabled_cnt / pref_cand;
	return raw_score / contrib * new_pref_fill * (2 - new_pref_fill);
}
