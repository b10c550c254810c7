/*
 * dag_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's annealing move primitives on the
 * candidate DAG:
 *   Token::Branch::SimulateStep  /root/reference/src/tokenizer/Token.cpp:17-22
 *   Token::Branch::ApplyStep     /root/reference/src/tokenizer/Token.cpp:26-37
 *   Token::SimulateStep          /root/reference/src/tokenizer/Token.cpp:83-88
 *   Token::ApplyStep             /root/reference/src/tokenizer/Token.cpp:90-105
 *   LomaxDist                    /root/reference/src/tokenizer/LomaxDist.cpp:5-51
 *   TokenGenerator::CalcScore    /root/reference/src/tokenizer/TokenGenerator.cpp:64-69
 *
 * The DAG is struct-of-arrays: token i has suffix parent lpar[i] (string minus
 * first char) and prefix parent rpar[i] (string minus last char), -1 for the
 * single-character roots; luses/ruses are the two Branch::uses counters.
 * Pinned against the compiled reference by tests/golden/dag_*.json
 * (oracle/_ref/ref_harness dag ...).
 */
#ifndef DAG_ORACLE_H
#define DAG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int64_t n;
	const int64_t *lpar;
	const int64_t *rpar;
	uint64_t *luses;
	uint64_t *ruses;
	uint8_t *enabled;
} ora_dag;

double ora_dag_simulate(const ora_dag *g, int64_t i);
double ora_dag_apply(ora_dag *g, int64_t i, int enable);

typedef struct {
	double alpha, moment1, moment2, beta, sigma;
} ora_lomax;

void ora_lomax_init(ora_lomax *d);
void ora_lomax_set_half_life(ora_lomax *d, double half_life);
void ora_lomax_set_moments(ora_lomax *d, double m1, double m2);
void ora_lomax_add_point(ora_lomax *d, double val, double weight);
void ora_lomax_update_params(ora_lomax *d);
double ora_lomax_get_best(const ora_lomax *d, double p);

double ora_calc_score(const ora_lomax *d, double raw_score, uint64_t enabled_cnt,
                      uint64_t tot_cand, uint64_t pref_cand);

#ifdef __cplusplus
}
#endif
#endif
