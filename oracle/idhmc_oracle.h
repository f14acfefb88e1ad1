/*
 * idhmc_oracle.h -- TEST INFRASTRUCTURE.  CPU oracle for the many-chain NUTS
 * hot path of chriselrod/InplaceDHMC.jl.
 *
 * This is a plain-C restatement of the reference's in-place CPU algorithm
 * (reference src/kinetic_energy.jl, src/hamiltonian.jl, src/tree.jl,
 * src/NUTS.jl, src/stepsize.jl, src/warmup.jl, src/mcmc.jl), one chain per
 * host thread as in `threaded_mcmc` (reference src/mcmc.jl:150-157).  It is
 * the checker for the HIP product and the `cpu_baseline` ("port") leg of
 * bench.py.  Nothing in the product may import, link or call it.
 *
 * parity unpinned: the reference holds no golden vectors (test/runtests.jl:4-6
 * is an empty testset), is pure Julia (no `julia` in this image) and draws
 * from an un-pinned third-party RNG, so this oracle cannot be checked against
 * reference outputs.  It is pinned instead by analytic known-answer tests
 * (tests/test_oracle_*.py) and follows the reference line by line where cited.
 */
#ifndef IDHMC_ORACLE_H
#define IDHMC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODEL_ISO_GAUSSIAN = 0, ORC_MODEL_DIAG_GAUSSIAN = 1, ORC_MODEL_DENSE_MVN = 2, ORC_MODEL_CALLBACK = 3 };

/* a user density on the CPU side (the product's IDHMC_MODEL_CUSTOM counterpart in tests): fills grad (length L,
 * pads zero) and returns l(q) */
typedef double (*orc_density_fn)(const double *q, double *grad, int D, int L, const double *params);

/* the user log density: reference `logdensity_and_gradient!` contract,
 * src/kinetic_energy.jl:73 */
typedef struct {
    int32_t kind, D, L;      /* L = D rounded up to a multiple of 128; pads are zero */
    const double *mu;        /* L (DIAG, DENSE) or NULL */
    const double *tau;       /* L, DIAG: 1/sigma^2 */
    const double *prec;      /* L*L row-major, DENSE: Sigma^-1 */
    orc_density_fn fn;       /* CALLBACK */
    const double *params;    /* CALLBACK */
} orc_model;

/* reference TreeStatisticsNUTS, src/NUTS.jl:229-242 (32 bytes) */
typedef struct {
    double pi;
    double acceptance_rate;
    int32_t term_left, term_right;   /* InvalidTree, src/tree.jl:278-300; (1,0) = REACHED_MAX_DEPTH */
    int32_t depth;
    int32_t steps;
} orc_tree_stats;

/* reference option structs flattened: NUTS (src/NUTS.jl:214-219),
 * DualAveraging (src/stepsize.jl:191-193), InitialStepsizeSearch
 * (src/stepsize.jl:29-37), default_warmup_stages (src/warmup.jl:361-372) */
typedef struct {
    int32_t max_depth;        /* 10 */
    double min_delta;         /* -1000 */
    double da_delta, da_gamma, da_kappa; /* 0.8 0.05 0.75 */
    int32_t da_t0;            /* 10 */
    double ss_a_min, ss_a_max, ss_eps0, ss_C; /* .25 .75 1 2 */
    int32_t ss_maxiter_crossing, ss_maxiter_bisect; /* 400 400 */
    int32_t init_steps, middle_steps, doubling_stages, terminating_steps; /* 75 25 5 50 */
    int32_t adapt_metric;     /* 1 = Diagonal in the doubling stages, 0 = Nothing */
    int32_t stepsize_search;  /* 1 = InitialStepsizeSearch stage, 0 = use eps_init */
    double eps_init;
    int32_t local_opt_iterations; /* FindLocalOptimum stage (src/warmup.jl:137-150): 0 = skipped (default here), reference 50 */
    double local_opt_penalty;     /* magnitude_penalty, reference 1e-4 */
} orc_options;

void orc_default_options(orc_options *o);

/* dual averaging state, src/stepsize.jl:196-202 */
typedef struct { double mu; int64_t m; double Hbar, logeps, logeps_bar; } orc_da_state;

typedef struct orc_chain orc_chain;

/* one chain (one reference thread): owns q, p, grad, M^-1, W, tree arena */
orc_chain *orc_chain_create(const orc_model *model, const orc_options *opt, uint64_t seed, uint32_t chain_id);
void orc_chain_destroy(orc_chain *c);
int orc_chain_L(const orc_chain *c);
/* state access (vectors of length L) */
double *orc_chain_q(orc_chain *c);
double *orc_chain_p(orc_chain *c);
double *orc_chain_grad(orc_chain *c);
double *orc_chain_minv(orc_chain *c);
double *orc_chain_w(orc_chain *c);
double orc_chain_lq(const orc_chain *c);
/* set q, recompute lq and grad (reference evaluate_l!, src/kinetic_energy.jl:72-85) */
void orc_chain_set_q(orc_chain *c, const double *q);
/* set M^-1 and W = 1/sqrt(M^-1) (src/hamiltonian.jl:50-57) */
void orc_chain_set_minv(orc_chain *c, const double *minv);
/* q ~ U[-2,2]^D (src/warmup.jl:73), evaluate */
void orc_chain_random_position(orc_chain *c);

/* building blocks, each citing its reference function in the .c file */
double orc_model_logdensity_and_gradient(const orc_model *m, const double *q, double *grad);
double orc_kinetic_energy(const double *minv, const double *p, int L);
void orc_rand_p(orc_chain *c, uint32_t iter);
double orc_chain_logdensity(const orc_chain *c);            /* pi = lq - K */
void orc_chain_leapfrog(orc_chain *c, double eps);          /* in place on (q,p,grad,lq) */
int orc_sample_tree(orc_chain *c, double eps, uint32_t iter, orc_tree_stats *stats);
/* FindLocalOptimum (src/warmup.jl:137-187): own L-BFGS, see the .c file; 0 or -5 (failed after 100 restarts) */
int orc_find_local_optimum(orc_chain *c, double magnitude_penalty, int iterations);
/* same with injected directions (reference kwarg, src/NUTS.jl:251-252) and,
 * if refresh_p == 0, the momentum already in the chain */
int orc_sample_tree_ex(orc_chain *c, double eps, uint32_t iter, int use_directions,
                       uint32_t directions, int refresh_p, orc_tree_stats *stats);
/* smallest decision margin met inside the last transition (for parity tests) */
double orc_chain_last_margin(const orc_chain *c);

void orc_da_init(orc_da_state *s, double eps);
void orc_da_adapt(const orc_options *o, orc_da_state *s, double a);
double orc_da_current_eps(const orc_da_state *s);
double orc_da_final_eps(const orc_da_state *s);

/* regularised diagonal metric from N draws stored column-major with stride L
 * (src/hamiltonian.jl:119-189) */
void orc_metric_from_draws(double *minv, double *w, const double *draws, int L, int D, int N, double lambda);

/* src/stepsize.jl:111-126,150-164; returns 0 ok, <0 on max-iteration error */
int orc_find_initial_stepsize(orc_chain *c, double *eps_out);

/* one chain end to end: mcmc_with_warmup! (src/mcmc.jl:94-105).
 * chain: L x NS column-major, stats: NS records, NS = max(N, longest stage). */
int orc_mcmc_with_warmup(orc_chain *c, int N, double *chain, orc_tree_stats *stats,
                         double *eps_final);
int orc_num_stored(const orc_options *o, int N);            /* NS */

/* threaded_mcmc (src/mcmc.jl:130-159): chains = L x NS x nchains */
int orc_threaded_mcmc(const orc_model *model, const orc_options *opt, uint64_t seed,
                      uint32_t first_chain, int nchains, int N, int nthreads,
                      double *chains, orc_tree_stats *stats, double *eps_final);

/* CPU baseline for bench.py: `sweeps` fixed-eps leapfrog sweeps over nchains
 * chains on nthreads threads; returns seconds. */
double orc_bench_leapfrog(const orc_model *model, uint64_t seed, int nchains, int sweeps,
                          double eps, const double *minv, int nthreads);
double orc_bench_nuts(const orc_model *model, uint64_t seed, int nchains, int transitions, double eps, const double *minv,
                      const double *q0, int nthreads, long *steps_out);

/* math + rng exports for the known-answer tests */
double orc_log_export(double x);
double orc_exp_export(double x);
double orc_log1p_export(double x);
void orc_sincos2pi_export(double u, double *s, double *c);
double orc_logaddexp_export(double x, double y);
void orc_philox_export(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_dot_export(const double *a, const double *b, int L);
double orc_randexp_export(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t draw);
uint32_t orc_rand_directions_export(uint64_t seed, uint32_t chain, uint32_t iter);
void orc_randn_export(uint64_t seed, uint32_t chain, uint32_t iter, int L, double *out);
double orc_logprob2_export(int bias, double w1, double w2);
int orc_is_turning_export(const double *rho, const double *psm, const double *psp, int L);
double orc_acceptance_rate_export(double log_sum_a, int steps);

#ifdef __cplusplus
}
#endif
#endif
