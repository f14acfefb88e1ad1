/*
 * idhmc_oracle.c -- TEST INFRASTRUCTURE.  See idhmc_oracle.h.
 *
 * CPU restatement of the reference's in-place NUTS path.  Structure follows
 * the reference (recursive `adjacent_tree`, separate passes for loop A / loop B
 * / kinetic energy / p# / rho / U-turn dots) so that it can serve as the
 * "reference's in-place CPU path" baseline; the HIP product is organised
 * differently (iterative, fused, one chain per wavefront) and is checked
 * against this file.
 *
 * parity unpinned (see header): no reference fixture exists for this path.
 *
 * All file:line citations are into /root/reference/.
 */
#include "idhmc_oracle.h"
#include "orc_math.h"
#include <stdlib.h>
#include <stdio.h>
#include <assert.h>
#include <pthread.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
void orc_default_options(orc_options *o)
{
    o->max_depth = 10;            /* src/tree.jl:2 */
    o->min_delta = -1000.0;       /* src/NUTS.jl:214 */
    o->da_delta = 0.8; o->da_gamma = 0.05; o->da_kappa = 0.75; o->da_t0 = 10; /* src/stepsize.jl:191 */
    o->ss_a_min = 0.25; o->ss_a_max = 0.75; o->ss_eps0 = 1.0; o->ss_C = 2.0; /* src/stepsize.jl:29 */
    o->ss_maxiter_crossing = 400; o->ss_maxiter_bisect = 400;
    o->init_steps = 75; o->middle_steps = 25; o->doubling_stages = 5; o->terminating_steps = 50; /* src/warmup.jl:366 */
    o->adapt_metric = 1;
    o->stepsize_search = 1;
    o->eps_init = 1.0;
    o->local_opt_iterations = 0;  /* FindLocalOptimum off by default at this level (own optimiser, see orc_find_local_optimum) */
    o->local_opt_penalty = 1e-4;  /* src/warmup.jl:143 */
}

/* ---- user density: the three built-ins the configs need ----------------- */
/* contract: reference logdensity_and_gradient!(grad, model, q, sptr) -> lq,
 * src/kinetic_energy.jl:73.  Arithmetic (canonical order, orc_math.h):
 *   d = q - mu; t = tau*d (DIAG/ISO) or t_r = fma-chain_c P[r][c]*d_c (DENSE)
 *   grad = -t;  lq = -1/2 * canonical_sum(t*d)                               */
double orc_model_logdensity_and_gradient(const orc_model *m, const double *q, double *grad)
{
    const int L = m->L;
    double acc[128];
    for (int r = 0; r < 128; ++r) acc[r] = 0.0;
    if (m->kind == ORC_MODEL_CALLBACK) return m->fn(q, grad, m->D, L, m->params);
    if (m->kind == ORC_MODEL_DENSE_MVN) {
        double *d = (double *)malloc(sizeof(double) * (size_t)L);
        for (int i = 0; i < L; ++i) d[i] = q[i] - m->mu[i];
        for (int r = 0; r < L; ++r) {
            const double *row = m->prec + (size_t)r * L;
            double t = 0.0;
            for (int c = 0; c < L; ++c) t = fma(row[c], d[c], t);
            grad[r] = -t;
            acc[r & 127] = fma(t, d[r], acc[r & 127]);
        }
        free(d);
    } else {
        for (int j = 0; j < L; j += 128)
            for (int r = 0; r < 128; ++r) {
                int i = j + r;
                double d = m->mu ? q[i] - m->mu[i] : q[i] - 0.0;
                double t = m->tau ? m->tau[i] * d : 1.0 * d;
                grad[i] = -t;
                acc[r] = fma(t, d, acc[r]);
            }
    }
    return -0.5 * orc_tree128(acc);
}

/* kinetic_energy, src/kinetic_energy.jl:14-24: K = 1/2 sum p*Minv*p */
double orc_kinetic_energy(const double *minv, const double *p, int L)
{
    double acc[128];
    for (int r = 0; r < 128; ++r) acc[r] = 0.0;
    for (int j = 0; j < L; j += 128)
        for (int r = 0; r < 128; ++r) {
            double ps = p[j + r] * minv[j + r];
            acc[r] = fma(ps, p[j + r], acc[r]);
        }
    return 0.5 * orc_tree128(acc);
}

/* calculate_p#, src/kinetic_energy.jl:39-46 */
static void calculate_psharp(double *out, const double *minv, const double *p, int L)
{
    for (int i = 0; i < L; ++i) out[i] = minv[i] * p[i];
}

/* logdensity(H, z), src/kinetic_energy.jl:107-112 */
static double phase_logdensity(double lq, const double *minv, const double *p, int L)
{
    if (!isfinite(lq)) return -INFINITY;
    double K = orc_kinetic_energy(minv, p, L);
    return lq - (isfinite(K) ? K : INFINITY);
}

/* ---- arena: the liveness guarantees of the reference Tree (src/tree.jl:16-121)
 * kept by reference counts instead of its bitmask free-list ---------------- */
typedef struct {
    double *buf; int width; int n; int *rc;
} pool_t;
static void pool_init(pool_t *p, int width, int n)
{
    p->buf = (double *)aligned_alloc(64, sizeof(double) * (size_t)width * n);
    p->rc = (int *)calloc((size_t)n, sizeof(int));
    p->width = width; p->n = n;
}
static void pool_free(pool_t *p) { free(p->buf); free(p->rc); }
static void pool_reset(pool_t *p) { for (int i = 0; i < p->n; ++i) p->rc[i] = 0; }
static int pool_alloc(pool_t *p)
{
    for (int i = 0; i < p->n; ++i) if (p->rc[i] == 0) { p->rc[i] = 1; return i; }
    fprintf(stderr, "idhmc oracle: arena exhausted (cf. src/tree.jl:80)\n");
    abort();
}
static void pool_retain(pool_t *p, int s) { assert(p->rc[s] > 0); p->rc[s]++; }
static void pool_release(pool_t *p, int s) { assert(p->rc[s] > 0); p->rc[s]--; } /* cf. "Double free", src/tree.jl:115 */
static double *pool_ptr(pool_t *p, int s) { return p->buf + (size_t)s * p->width; }

struct orc_chain {
    orc_model model; orc_options opt;
    uint64_t seed; uint32_t id; int L, D;
    double *q, *p, *g, *minv, *w; double lq;
    pool_t zpool;            /* slots [p | q | grad], src/kinetic_energy.jl:135-139 */
    pool_t vpool;            /* rho and p# vectors */
    double *zlq, *zpi;       /* per z slot: l(q) and pi = l(q) - K */
    /* per-transition */
    double pi0, eps; uint32_t iter, draw; double margin;
};

static double *zp(orc_chain *c, int s) { return pool_ptr(&c->zpool, s); }
static double *zq(orc_chain *c, int s) { return pool_ptr(&c->zpool, s) + c->L; }
static double *zg(orc_chain *c, int s) { return pool_ptr(&c->zpool, s) + 2 * c->L; }

orc_chain *orc_chain_create(const orc_model *model, const orc_options *opt, uint64_t seed, uint32_t chain_id)
{
    orc_chain *c = (orc_chain *)calloc(1, sizeof(orc_chain));
    c->model = *model; c->opt = *opt; c->seed = seed; c->id = chain_id;
    c->L = model->L; c->D = model->D;
    size_t n = (size_t)c->L;
    c->q = (double *)calloc(n, 8); c->p = (double *)calloc(n, 8); c->g = (double *)calloc(n, 8);
    c->minv = (double *)calloc(n, 8); c->w = (double *)calloc(n, 8);
    /* GaussianKineticEnergy(sptr, Static{D}, 1.0): M^-1 = I, src/hamiltonian.jl:63-74 */
    for (int i = 0; i < c->L; ++i) { c->minv[i] = 1.0; c->w[i] = 1.0; }
    int nz = 2 * (opt->max_depth + 4), nv = 4 * (opt->max_depth + 4);
    pool_init(&c->zpool, 3 * c->L, nz);
    pool_init(&c->vpool, c->L, nv);
    c->zlq = (double *)calloc((size_t)nz, 8); c->zpi = (double *)calloc((size_t)nz, 8);
    c->lq = orc_model_logdensity_and_gradient(&c->model, c->q, c->g);
    c->margin = INFINITY;
    return c;
}
void orc_chain_destroy(orc_chain *c)
{
    if (!c) return;
    free(c->q); free(c->p); free(c->g); free(c->minv); free(c->w);
    pool_free(&c->zpool); pool_free(&c->vpool); free(c->zlq); free(c->zpi); free(c);
}
int orc_chain_L(const orc_chain *c) { return c->L; }
double *orc_chain_q(orc_chain *c) { return c->q; }
double *orc_chain_p(orc_chain *c) { return c->p; }
double *orc_chain_grad(orc_chain *c) { return c->g; }
double *orc_chain_minv(orc_chain *c) { return c->minv; }
double *orc_chain_w(orc_chain *c) { return c->w; }
double orc_chain_lq(const orc_chain *c) { return c->lq; }
double orc_chain_last_margin(const orc_chain *c) { return c->margin; }

/* evaluate_l!, src/kinetic_energy.jl:72-85: non-finite l(q) becomes -Inf */
static double evaluate_l(const orc_model *m, const double *q, double *g)
{
    double lq = orc_model_logdensity_and_gradient(m, q, g);
    return isfinite(lq) ? lq : -INFINITY;
}
void orc_chain_set_q(orc_chain *c, const double *q)
{
    for (int i = 0; i < c->L; ++i) c->q[i] = i < c->D ? q[i] : 0.0;
    c->lq = evaluate_l(&c->model, c->q, c->g);
}
void orc_chain_set_minv(orc_chain *c, const double *minv)
{
    for (int i = 0; i < c->L; ++i) {
        c->minv[i] = i < c->D ? minv[i] : 1.0;
        c->w[i] = 1.0 / sqrt(c->minv[i]);          /* src/hamiltonian.jl:53-55 */
    }
}
/* random_position!, src/warmup.jl:73: q ~ U[-2,2) */
void orc_chain_random_position(orc_chain *c)
{
    for (int k = 0; k < c->L / 2; ++k) {
        uint32_t x[4];
        orc_rng(c->seed, c->id, 0, ORC_STREAM_INITQ, (uint32_t)k, x);
        double u0 = orc_u01(x[0], x[1]), u1 = orc_u01(x[2], x[3]);
        c->q[2 * k] = 2 * k < c->D ? fma(4.0, u0, -2.0) : 0.0;
        c->q[2 * k + 1] = 2 * k + 1 < c->D ? fma(4.0, u1, -2.0) : 0.0;
    }
    c->lq = evaluate_l(&c->model, c->q, c->g);
}

/* rand_p!, src/kinetic_energy.jl:63: p = W .* randn */
void orc_rand_p(orc_chain *c, uint32_t iter)
{
    for (int k = 0; k < c->L / 2; ++k) {
        double n0, n1;
        orc_randn_pair(c->seed, c->id, iter, (uint32_t)k, &n0, &n1);
        c->p[2 * k] = 2 * k < c->D ? c->w[2 * k] * n0 : 0.0;
        c->p[2 * k + 1] = 2 * k + 1 < c->D ? c->w[2 * k + 1] * n1 : 0.0;
    }
}
double orc_chain_logdensity(const orc_chain *c) { return phase_logdensity(c->lq, c->minv, c->p, c->L); }

/* leapfrog, src/kinetic_energy.jl:126-163.  (p,q,g) -> (p',q',g'), returns l(q'). */
static double leapfrog_vec(const orc_model *model, const double *minv, int L, double eps,
                           const double *p, const double *q, const double *g,
                           double *p1, double *q1, double *g1)
{
    const double eh = 0.5 * eps;                              /* :145 */
    for (int l = 0; l < L; ++l) {                             /* loop A :146-150 */
        double pm = fma(eh, g[l], p[l]);
        p1[l] = pm;
        q1[l] = fma(eps * minv[l], pm, q[l]);
    }
    double lq = evaluate_l(model, q1, g1);                    /* :154 */
    for (int l = 0; l < L; ++l) p1[l] = fma(eh, g1[l], p1[l]); /* loop B :159-161 */
    return lq;
}
void orc_chain_leapfrog(orc_chain *c, double eps)
{
    c->lq = leapfrog_vec(&c->model, c->minv, c->L, eps, c->p, c->q, c->g, c->p, c->q, c->g);
}

/* ---- NUTS --------------------------------------------------------------- */
typedef struct { double log_sum_a; int32_t steps; } acc_stat;      /* src/NUTS.jl:58-66 */
static acc_stat combine_acc(acc_stat A, acc_stat B)                 /* :68-70 */
{
    acc_stat r = { orc_logaddexp(A.log_sum_a, B.log_sum_a), A.steps + B.steps };
    return r;
}
typedef struct { int psm, psp, rho; int rho_alias_z; } turn_stat;   /* src/NUTS.jl:93-97; rho_alias_z>=0: rho aliases that z slot's p (flag 0, :115) */
typedef struct { int zeta; double omega; turn_stat tau; int zlast; int32_t ilast; } subtree;
typedef struct { int32_t left, right; } invalid_tree;

static const double *rho_ptr(orc_chain *c, const turn_stat *t)
{
    return t->rho_alias_z >= 0 ? zp(c, t->rho_alias_z) : pool_ptr(&c->vpool, t->rho);
}
static void note_margin(orc_chain *c, double m) { m = fabs(m); if (m < c->margin) c->margin = m; }

/* rand_bool_logprob, src/NUTS.jl:32-34: no draw when logprob >= 0 */
static int rand_bool_logprob(orc_chain *c, double logprob)
{
    if (logprob >= 0) return 1;
    double e = orc_randexp(c->seed, c->id, c->iter, c->draw++);
    note_margin(c, e + logprob);
    return e > -logprob;
}
/* combine_proposals_and_logweights, src/tree.jl:238-245 with
 * biased_progressive_logprob2 (:261-263) and combine_proposals (src/NUTS.jl:40-45) */
static int combine_proposals(orc_chain *c, int z1, int z2, double w1, double w2, int is_doubling, double *w)
{
    *w = orc_logaddexp(w1, w2);
    double logprob2 = w2 - (is_doubling ? w1 : *w);
    if (rand_bool_logprob(c, logprob2)) { pool_release(&c->zpool, z1); return z2; }
    pool_release(&c->zpool, z2); return z1;
}
/* combine_turn_statistics, src/NUTS.jl:118-145 (x earlier in time than y) */
static turn_stat combine_turn(orc_chain *c, turn_stat x, turn_stat y)
{
    const double *rx = rho_ptr(c, &x), *ry = rho_ptr(c, &y);
    turn_stat r;
    double *rho;
    if (x.rho_alias_z >= 0) { r.rho = pool_alloc(&c->vpool); rho = pool_ptr(&c->vpool, r.rho); } /* :126-130 */
    else { r.rho = x.rho; rho = pool_ptr(&c->vpool, r.rho); }                                     /* :133 */
    for (int l = 0; l < c->L; ++l) rho[l] = rx[l] + ry[l];                                        /* :139-141 */
    if (x.rho_alias_z < 0 && y.rho_alias_z < 0) pool_release(&c->vpool, y.rho);                   /* :135 */
    else if (x.rho_alias_z >= 0 && y.rho_alias_z < 0) pool_release(&c->vpool, y.rho);
    pool_release(&c->vpool, x.psp);                                                               /* :136 */
    pool_release(&c->vpool, y.psm);                                                               /* :137 */
    r.psm = x.psm; r.psp = y.psp; r.rho_alias_z = -1;
    return r;
}
static void release_turn(orc_chain *c, turn_stat t)
{
    pool_release(&c->vpool, t.psm); pool_release(&c->vpool, t.psp);
    if (t.rho_alias_z < 0) pool_release(&c->vpool, t.rho);
}
/* is_turning, src/NUTS.jl:148-170 */
static int is_turning(orc_chain *c, const turn_stat *t)
{
    const double *rho = rho_ptr(c, t);
    double dm = orc_dot(rho, pool_ptr(&c->vpool, t->psm), c->L);
    double dp = orc_dot(rho, pool_ptr(&c->vpool, t->psp), c->L);
    note_margin(c, dm); note_margin(c, dp);
    return (dm < 0.0) | (dp < 0.0);
}
/* leaf, src/NUTS.jl:176-191 (non-initial); returns 1 if divergent */
static int leaf(orc_chain *c, int z, subtree *t, acc_stat *v)
{
    double delta = c->zpi[z] - c->pi0;                                  /* :179 */
    int isdiv = delta < c->opt.min_delta;                               /* :180 */
    v->log_sum_a = delta < 0.0 ? delta : 0.0; v->steps = 1;             /* :76-78 */
    if (isdiv) return 1;
    int ps = pool_alloc(&c->vpool);                                     /* leaf_turn_statistic :113-116 */
    calculate_psharp(pool_ptr(&c->vpool, ps), c->minv, zp(c, z), c->L);
    pool_retain(&c->vpool, ps);
    t->tau.psm = ps; t->tau.psp = ps; t->tau.rho = -1; t->tau.rho_alias_z = z;
    t->zeta = z; pool_retain(&c->zpool, z);
    t->omega = delta;
    return 0;
}
/* move, src/NUTS.jl:18-21 + leapfrog into a fresh slot */
static int move(orc_chain *c, int z, int fwd)
{
    int s = pool_alloc(&c->zpool);
    double e = fwd ? c->eps : -c->eps;
    c->zlq[s] = leapfrog_vec(&c->model, c->minv, c->L, e, zp(c, z), zq(c, z), zg(c, z), zp(c, s), zq(c, s), zg(c, s));
    c->zpi[s] = phase_logdensity(c->zlq[s], c->minv, zp(c, s), c->L);
    return s;
}

/* adjacent_tree, src/tree.jl:321-366.  Returns 1 when invalid (then *t is
 * unusable and all its storage has been released). */
static int adjacent_tree(orc_chain *c, int z, int32_t i, int32_t depth, int fwd,
                         subtree *t, acc_stat *v, invalid_tree *it)
{
    int32_t i1 = i + (fwd ? 1 : -1);                                    /* :322 */
    if (depth == 0) {                                                   /* :328-332 */
        int z1 = move(c, z, fwd);
        t->zlast = z1; t->ilast = i1;
        if (leaf(c, z1, t, v)) {
            it->left = i1; it->right = i1;
            pool_release(&c->zpool, z1);
            return 1;
        }
        return 0;
    }
    subtree tm, tp; acc_stat vm, vp;
    if (adjacent_tree(c, z, i, depth - 1, fwd, &tm, &vm, it)) { *v = vm; return 1; }   /* :335-340 */
    int inv = adjacent_tree(c, tm.zlast, tm.ilast, depth - 1, fwd, &tp, &vp, it);       /* :344-346 */
    /* a depth-0 tau aliases its z slot's momentum; keep that slot alive until tau is merged */
    *v = combine_acc(vm, vp);                                                           /* :347 */
    if (inv) {                                                                          /* :348 */
        pool_release(&c->zpool, tm.zlast); pool_release(&c->zpool, tm.zeta); release_turn(c, tm.tau);
        return 1;
    }
    turn_stat tau = fwd ? combine_turn(c, tm.tau, tp.tau) : combine_turn(c, tp.tau, tm.tau); /* :354, :230-236 */
    pool_release(&c->zpool, tm.zlast);
    if (is_turning(c, &tau)) {                                                          /* :358 */
        it->left = i1; it->right = tp.ilast;
        release_turn(c, tau);
        pool_release(&c->zpool, tm.zeta); pool_release(&c->zpool, tp.zeta); pool_release(&c->zpool, tp.zlast);
        return 1;
    }
    t->zeta = combine_proposals(c, tm.zeta, tp.zeta, tm.omega, tp.omega, 0, &t->omega); /* :361-363 */
    t->tau = tau; t->zlast = tp.zlast; t->ilast = tp.ilast;
    return 0;
}

/* sample_tree + sample_trajectory, src/NUTS.jl:251-264, src/tree.jl:382-444 */
int orc_sample_tree_ex(orc_chain *c, double eps, uint32_t iter, int use_directions,
                       uint32_t directions, int refresh_p, orc_tree_stats *stats)
{
    const int L = c->L;
    c->iter = iter; c->draw = 0; c->eps = eps; c->margin = INFINITY;
    uint32_t dirs = use_directions ? directions : orc_rand_directions(c->seed, c->id, iter); /* NUTS.jl:252 */
    if (refresh_p) orc_rand_p(c, iter);                                                      /* :254 */
    pool_reset(&c->zpool); pool_reset(&c->vpool);
    int z0 = pool_alloc(&c->zpool);
    memcpy(zp(c, z0), c->p, 8 * (size_t)L); memcpy(zq(c, z0), c->q, 8 * (size_t)L); memcpy(zg(c, z0), c->g, 8 * (size_t)L);
    c->zlq[z0] = c->lq;
    c->pi0 = phase_logdensity(c->lq, c->minv, c->p, L);                                      /* :260 */
    c->zpi[z0] = c->pi0;
    /* initial leaf, tree.jl:388 / NUTS.jl:176-191 with is_initial = true */
    int zeta = z0; pool_retain(&c->zpool, z0);
    double omega = 0.0;
    acc_stat v = { -INFINITY, 0 };
    turn_stat tau;
    tau.psm = pool_alloc(&c->vpool);
    calculate_psharp(pool_ptr(&c->vpool, tau.psm), c->minv, zp(c, z0), L);
    pool_retain(&c->vpool, tau.psm);
    tau.psp = tau.psm; tau.rho = -1; tau.rho_alias_z = z0;
    int zm = z0, zpl = z0; pool_retain(&c->zpool, z0);         /* z-, z+ (tree.jl:389) */
    int32_t im = 0, ip = 0, depth = 0;
    invalid_tree term = { 1, 0 };                              /* REACHED_MAX_DEPTH, tree.jl:300 */
    while (depth < c->opt.max_depth) {                         /* tree.jl:395 */
        int fwd = (int)(dirs & 1u); dirs >>= 1;                /* next_direction, :152-155 */
        int zi = fwd ? zpl : zm; int32_t ii = fwd ? ip : im;
        subtree t1; acc_stat v1; invalid_tree it;
        int inv = adjacent_tree(c, zi, ii, depth, fwd, &t1, &v1, &it);   /* :410 */
        v = combine_acc(v, v1);                                /* :414 */
        if (inv) { term = it; break; }                         /* :417 */
        if (fwd) { pool_release(&c->zpool, zpl); zpl = t1.zlast; ip = t1.ilast; }
        else { pool_release(&c->zpool, zm); zm = t1.zlast; im = t1.ilast; }   /* :424-428 */
        zeta = combine_proposals(c, zeta, t1.zeta, omega, t1.omega, 1, &omega); /* :431-433 */
        depth += 1;
        tau = fwd ? combine_turn(c, tau, t1.tau) : combine_turn(c, t1.tau, tau); /* :437 */
        if (is_turning(c, &tau)) { term.left = im; term.right = ip; break; }     /* :438 */
    }
    stats->pi = c->zpi[zeta];                                  /* logdensity(H, zeta), NUTS.jl:262 */
    double a = orc_exp(v.log_sum_a) / (double)v.steps;         /* acceptance_rate, :84 */
    stats->acceptance_rate = a < 1.0 ? a : 1.0;
    stats->term_left = term.left; stats->term_right = term.right;
    stats->depth = depth; stats->steps = v.steps;
    /* zeta becomes the next z (warmup.jl:298, 325) */
    memcpy(c->q, zq(c, zeta), 8 * (size_t)L); memcpy(c->g, zg(c, zeta), 8 * (size_t)L);
    memcpy(c->p, zp(c, zeta), 8 * (size_t)L);
    c->lq = c->zlq[zeta];
    return 0;
}
int orc_sample_tree(orc_chain *c, double eps, uint32_t iter, orc_tree_stats *stats)
{
    return orc_sample_tree_ex(c, eps, iter, 0, 0, 1, stats);
}

/* ---- dual averaging, src/stepsize.jl:196-241 ---------------------------- */
void orc_da_init(orc_da_state *s, double eps)                  /* :208-212 */
{
    double le = orc_log(eps);
    s->mu = orc_log(10.0) + le; s->m = 0; s->Hbar = 0.0; s->logeps = le; s->logeps_bar = 0.0;
}
void orc_da_adapt(const orc_options *o, orc_da_state *s, double a)   /* :220-229 */
{
    s->m += 1;
    double m = (double)s->m;
    s->Hbar += (o->da_delta - a - s->Hbar) / (m + (double)o->da_t0);
    s->logeps = s->mu - sqrt(m) / o->da_gamma * s->Hbar;
    s->logeps_bar += orc_exp(-o->da_kappa * orc_log(m)) * (s->logeps - s->logeps_bar);
}
double orc_da_current_eps(const orc_da_state *s) { return orc_exp(s->logeps); }     /* :235 */
double orc_da_final_eps(const orc_da_state *s) { return orc_exp(s->logeps_bar); }   /* :241 */

/* ---- GaussianKineticEnergy!(regs2, invs, sample, lambda), src/hamiltonian.jl:119-189 */
void orc_metric_from_draws(double *minv, double *w, const double *draws, int L, int D, int N, double lambda)
{
    double dN = (double)N;
    double Ninv = 1.0 / dN;                                        /* :156 */
    double mulreg = dN / ((dN + lambda) * (dN - 1.0));             /* :157 */
    double addreg = 1e-3 * lambda / (dN + lambda);                 /* :158 */
    for (int i = 0; i < L; ++i) {
        if (i >= D) { minv[i] = 1.0; w[i] = 1.0; continue; }
        double mu = draws[i], sd = 0.0, sd2 = 0.0;                 /* :86-88 */
        for (int n = 1; n < N; ++n) {                              /* :89-93 */
            double d = draws[(size_t)n * L + i] - mu;
            sd = d + sd;
            sd2 = fma(d, d, sd2);
        }
        double sdsd = sd * sd;                                     /* :94 */
        double s2 = fma(-sdsd, Ninv, sd2);                         /* vfnmadd :95 */
        double regs2 = fma(s2, mulreg, addreg);                    /* :96 */
        minv[i] = regs2;
        w[i] = 1.0 / sqrt(regs2);                                  /* rsqrt :97 (exact 1/sqrt here) */
    }
}

/* ---- initial stepsize, src/stepsize.jl:51-164 --------------------------- */
typedef struct { orc_chain *c; double target; double *p1, *q1, *g1; } ratio_ctx;
static double local_ratio(ratio_ctx *r, double eps)                /* :150-154 */
{
    orc_chain *c = r->c;
    double lq = leapfrog_vec(&c->model, c->minv, c->L, eps, c->p, c->q, c->g, r->p1, r->q1, r->g1);
    return orc_exp(phase_logdensity(lq, c->minv, r->p1, c->L) - r->target);
}
int orc_find_initial_stepsize(orc_chain *c, double *eps_out)
{
    const orc_options *o = &c->opt;
    ratio_ctx r; r.c = c; r.target = orc_chain_logdensity(c);
    if (!isfinite(r.target)) return -3;                            /* :152-153 */
    r.p1 = (double *)malloc(24 * (size_t)c->L); r.q1 = r.p1 + c->L; r.g1 = r.q1 + c->L;
    int rc = 0;
    double e0 = o->ss_eps0, A0 = local_ratio(&r, e0);              /* :113 */
    double result = e0;
    if (!(o->ss_a_min <= A0 && A0 <= o->ss_a_max)) {               /* :114 */
        /* find_crossing_stepsize :51-72 */
        double s = A0 > o->ss_a_max ? 1.0 : -1.0, a = A0 > o->ss_a_max ? o->ss_a_max : o->ss_a_min;
        double C = s < 0 ? 1.0 / o->ss_C : o->ss_C;
        double e1 = e0, A1 = A0; int found = 0;
        for (int it = 0; it < o->ss_maxiter_crossing; ++it) {
            double e = e0 * C, Ae = local_ratio(&r, e);
            if (s * (Ae - a) <= 0) { e1 = e; A1 = Ae; found = 1; break; }
            e0 = e; A0 = Ae;
        }
        if (!found) { rc = -1; goto done; }                        /* :71 */
        if (o->ss_a_min <= A1 && A1 <= o->ss_a_max) { result = e1; goto done; }   /* :118 */
        double lo = e0, hi = e1;                                   /* :120-124 */
        if (!(e0 < e1)) { lo = e1; hi = e0; }
        found = 0;
        for (int it = 0; it < o->ss_maxiter_bisect; ++it) {        /* bisect_stepsize :83-102 */
            double em = 0.5 * (lo + hi), Am = local_ratio(&r, em); /* middle() */
            if (o->ss_a_min <= Am && Am <= o->ss_a_max) { result = em; found = 1; break; }
            else if (Am < o->ss_a_min) hi = em;
            else lo = em;
        }
        if (!found) rc = -2;                                       /* :101 */
    }
done:
    free(r.p1);
    *eps_out = result;
    return rc;
}

/* ---- warmup stages + mcmc, src/warmup.jl:269-332, 361-408 --------------- */
static int longest_stage(const orc_options *o)
{
    int m = o->init_steps;
    for (int d = 0; d < o->doubling_stages; ++d) { int n = o->middle_steps << d; if (n > m) m = n; }
    if (o->terminating_steps > m) m = o->terminating_steps;
    return m;
}
int orc_num_stored(const orc_options *o, int N) { int w = longest_stage(o); return N > w ? N : w; } /* src/mcmc.jl:115-116 */

/* warmup!(TuningNUTS), src/warmup.jl:269-314 */
static int tuning_stage(orc_chain *c, int N, int adapt_metric, double *chain, orc_tree_stats *stats,
                        double *eps, uint32_t *iter)
{
    orc_da_state da; orc_da_init(&da, *eps);                       /* :284 */
    double lambda = 5.0 / (double)N;                               /* :229 */
    for (int n = 0; n < N; ++n) {                                  /* :288 */
        double e = orc_da_current_eps(&da);                        /* :289 */
        if (e < 1e-10) return -4;                                  /* :291-296 */
        orc_sample_tree(c, e, ++(*iter), &stats[n]);               /* :298 */
        memcpy(chain + (size_t)n * c->L, c->q, 8 * (size_t)c->L);  /* :299 */
        orc_da_adapt(&c->opt, &da, stats[n].acceptance_rate);      /* :303 */
    }
    if (adapt_metric) orc_metric_from_draws(c->minv, c->w, chain, c->L, c->D, N, lambda);  /* :308-311 */
    *eps = orc_da_final_eps(&da);                                  /* :313 */
    return 0;
}
/* ---- FindLocalOptimum, src/warmup.jl:137-187 ------------------------------------------------------------
 * Contract of the reference stage: maximise l(q) - 1/2 * magnitude_penalty * sum(q^2) for at most `iterations`
 * iterations of a quasi-Newton method; if the result is not finite, draw a new random position, double the
 * penalty and try again, at most 100 times (:162-171), else fail (:172); on success the chain's (q, l(q),
 * grad l(q)) is the optimum.  The reference's optimiser is QuasiNewtonMethods.proptimize! (:163), an external
 * package whose source is not in the reference tree (parity unpinned); the iteration below is this engine's
 * own: L-BFGS with ORC_LBFGS_M pairs and Armijo backtracking, every reduction in the canonical order, so the
 * HIP kernel (csrc/idhmc_optimum.hpp) reproduces it bit for bit.  Minimises F(x) = -l(x) + lam/2 x.x. */
#define ORC_LBFGS_M 5
static void random_position_attempt(orc_chain *c, uint32_t attempt)
{
    for (int k = 0; k < c->L / 2; ++k) {
        uint32_t x[4];
        orc_rng(c->seed, c->id, attempt, ORC_STREAM_INITQ, (uint32_t)k, x);
        double u0 = orc_u01(x[0], x[1]), u1 = orc_u01(x[2], x[3]);
        c->q[2 * k] = 2 * k < c->D ? fma(4.0, u0, -2.0) : 0.0;
        c->q[2 * k + 1] = 2 * k + 1 < c->D ? fma(4.0, u1, -2.0) : 0.0;
    }
    c->lq = evaluate_l(&c->model, c->q, c->g);
}
int orc_find_local_optimum(orc_chain *c, double magnitude_penalty, int iterations)
{
    /* ring of R = M + 1 slots, at most M of them valid: slot `head` is always free, so a new pair can be formed in
     * place and simply not committed when it fails the curvature test */
    const int L = c->L, M = ORC_LBFGS_M, R = ORC_LBFGS_M + 1;
    double *buf = (double *)aligned_alloc(64, sizeof(double) * (size_t)L * (5 + 2 * R));
    double *G = buf, *r = buf + L, *xn = buf + 2 * L, *gln = buf + 3 * L, *Gn = buf + 4 * L;
    double *S = buf + 5 * L, *Y = S + (size_t)R * L;
    double lam = magnitude_penalty;
    int rc = -5;
    for (uint32_t attempt = 0; attempt < 100; ++attempt) {                       /* :162 */
        double *x = c->q, *gl = c->g;
        double lq = c->lq;
        double xx = orc_dot(x, x, L);
        double F = fma(0.5 * lam, xx, -lq);
        for (int i = 0; i < L; ++i) G[i] = fma(lam, x[i], -gl[i]);
        double rho[ORC_LBFGS_M + 1], alpha[ORC_LBFGS_M + 1], gamma = 1.0;
        int k = 0, head = 0;
        for (int it = 0; it < iterations; ++it) {
            const double gg = orc_dot(G, G, L);
            if (!(gg > 1e-16 * (xx > 1.0 ? xx : 1.0))) break;                   /* converged (or not a number) */
            for (int i = 0; i < L; ++i) r[i] = G[i];
            for (int j = 0; j < k; ++j) {                                        /* two-loop recursion, newest first */
                const int i = (head + R - 1 - j) % R;
                alpha[i] = rho[i] * orc_dot(S + (size_t)i * L, r, L);
                for (int e = 0; e < L; ++e) r[e] = fma(-alpha[i], Y[(size_t)i * L + e], r[e]);
            }
            const double scale = k > 0 ? gamma : 1.0 / sqrt(gg);
            for (int e = 0; e < L; ++e) r[e] = scale * r[e];
            for (int j = k - 1; j >= 0; --j) {                                   /* oldest first */
                const int i = (head + R - 1 - j) % R;
                const double beta = rho[i] * orc_dot(Y + (size_t)i * L, r, L);
                for (int e = 0; e < L; ++e) r[e] = fma(alpha[i] - beta, S[(size_t)i * L + e], r[e]);
            }
            double gd = -orc_dot(G, r, L);                                       /* direction d = -r */
            if (!(gd < 0.0)) {                                                   /* not a descent direction: restart */
                k = 0;
                const double sc = 1.0 / sqrt(gg);
                for (int e = 0; e < L; ++e) r[e] = sc * G[e];
                gd = -orc_dot(G, r, L);
            }
            double t = 1.0, lqn = 0.0, xxn = 0.0, Fn = 0.0;
            int accepted = 0;
            for (int ls = 0; ls < 30; ++ls) {                                    /* Armijo backtracking */
                for (int e = 0; e < L; ++e) xn[e] = fma(-t, r[e], x[e]);
                lqn = evaluate_l(&c->model, xn, gln);
                xxn = orc_dot(xn, xn, L);
                Fn = fma(0.5 * lam, xxn, -lqn);
                if (isfinite(Fn) && Fn <= fma(1e-4 * t, gd, F)) { accepted = 1; break; }
                t *= 0.5;
            }
            if (!accepted) break;
            for (int e = 0; e < L; ++e) Gn[e] = fma(lam, xn[e], -gln[e]);
            double *Sh = S + (size_t)head * L, *Yh = Y + (size_t)head * L;
            for (int e = 0; e < L; ++e) { Sh[e] = xn[e] - x[e]; Yh[e] = Gn[e] - G[e]; }
            const double sy = orc_dot(Sh, Yh, L), yy = orc_dot(Yh, Yh, L);
            if (sy > 1e-10 * yy) {                                               /* curvature condition: keep the pair */
                rho[head] = 1.0 / sy;
                gamma = sy / yy;
                head = (head + 1) % R;
                if (k < M) ++k;
            }
            for (int e = 0; e < L; ++e) { x[e] = xn[e]; gl[e] = gln[e]; G[e] = Gn[e]; }
            lq = lqn; F = Fn; xx = xxn;
        }
        c->lq = lq;
        if (isfinite(lq)) { rc = 0; break; }                                     /* :168 */
        random_position_attempt(c, attempt + 1);                                 /* :169-170 */
        lam += lam;                                                              /* :171 */
    }
    free(buf);
    return rc;                                                                   /* -5: "Optimization failed to converge", :172 */
}

int orc_mcmc_with_warmup(orc_chain *c, int N, double *chain, orc_tree_stats *stats, double *eps_final)
{
    const orc_options *o = &c->opt;
    uint32_t iter = 0; int rc;
    orc_chain_random_position(c);                                  /* initialize_warmup_state, warmup.jl:100-129 */
    if (o->local_opt_iterations > 0 &&                             /* FindLocalOptimum, warmup.jl:152-186 */
        (rc = orc_find_local_optimum(c, o->local_opt_penalty, o->local_opt_iterations)) != 0) return rc;
    double eps = o->eps_init;
    if (o->stepsize_search) {                                      /* warmup.jl:188-200 */
        orc_rand_p(c, 0);
        if ((rc = orc_find_initial_stepsize(c, &eps)) != 0) return rc;
    }
    if ((rc = tuning_stage(c, o->init_steps, 0, chain, stats, &eps, &iter)) != 0) return rc;      /* warmup.jl:369 */
    for (int d = 0; d < o->doubling_stages; ++d)                                                  /* :341-344 */
        if ((rc = tuning_stage(c, o->middle_steps << d, o->adapt_metric, chain, stats, &eps, &iter)) != 0) return rc;
    if ((rc = tuning_stage(c, o->terminating_steps, 0, chain, stats, &eps, &iter)) != 0) return rc; /* :371 */
    for (int n = 0; n < N; ++n) {                                  /* mcmc!, warmup.jl:316-332 */
        orc_sample_tree(c, eps, ++iter, &stats[n]);
        memcpy(chain + (size_t)n * c->L, c->q, 8 * (size_t)c->L);
    }
    if (eps_final) *eps_final = eps;
    return 0;
}

/* ---- threaded_mcmc, src/mcmc.jl:130-159 --------------------------------- */
typedef struct {
    const orc_model *model; const orc_options *opt; uint64_t seed; uint32_t first; int nchains, N, NS;
    double *chains; orc_tree_stats *stats; double *eps_final; int *next; pthread_mutex_t *mu; int rc;
} tm_job;
static void *tm_worker(void *arg)
{
    tm_job *j = (tm_job *)arg;
    for (;;) {
        pthread_mutex_lock(j->mu); int t = (*j->next)++; pthread_mutex_unlock(j->mu);
        if (t >= j->nchains) break;
        orc_chain *c = orc_chain_create(j->model, j->opt, j->seed, j->first + (uint32_t)t);
        int rc = orc_mcmc_with_warmup(c, j->N, j->chains + (size_t)t * j->NS * c->L,
                                      j->stats + (size_t)t * j->NS, j->eps_final ? j->eps_final + t : NULL);
        if (rc) j->rc = rc;
        orc_chain_destroy(c);
    }
    return NULL;
}
int orc_threaded_mcmc(const orc_model *model, const orc_options *opt, uint64_t seed, uint32_t first_chain,
                      int nchains, int N, int nthreads, double *chains, orc_tree_stats *stats, double *eps_final)
{
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER; int next = 0;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nchains) nthreads = nchains;
    tm_job *jobs = (tm_job *)calloc((size_t)nthreads, sizeof(tm_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    int NS = orc_num_stored(opt, N);
    for (int i = 0; i < nthreads; ++i) {
        tm_job j = { model, opt, seed, first_chain, nchains, N, NS, chains, stats, eps_final, &next, &mu, 0 };
        jobs[i] = j;
        pthread_create(&th[i], NULL, tm_worker, &jobs[i]);
    }
    int rc = 0;
    for (int i = 0; i < nthreads; ++i) { pthread_join(th[i], NULL); if (jobs[i].rc) rc = jobs[i].rc; }
    free(jobs); free(th);
    return rc;
}

/* ---- CPU baseline: fixed-eps leapfrog sweeps ----------------------------- */
typedef struct { orc_chain **chains; int lo, hi, sweeps; double eps; } lf_job;
static void *lf_worker(void *arg)
{
    lf_job *j = (lf_job *)arg;
    for (int s = 0; s < j->sweeps; ++s)
        for (int t = j->lo; t < j->hi; ++t) orc_chain_leapfrog(j->chains[t], j->eps);
    return NULL;
}
double orc_bench_leapfrog(const orc_model *model, uint64_t seed, int nchains, int sweeps, double eps,
                          const double *minv, int nthreads)
{
    orc_options o; orc_default_options(&o);
    orc_chain **cs = (orc_chain **)calloc((size_t)nchains, sizeof(orc_chain *));
    for (int t = 0; t < nchains; ++t) {
        cs[t] = orc_chain_create(model, &o, seed, (uint32_t)t);
        if (minv) orc_chain_set_minv(cs[t], minv);
        orc_chain_random_position(cs[t]);
        orc_rand_p(cs[t], 0);
    }
    if (nthreads > nchains) nthreads = nchains;
    lf_job *jobs = (lf_job *)calloc((size_t)nthreads, sizeof(lf_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < nthreads; ++i) {
        jobs[i].chains = cs; jobs[i].lo = (int)((long)nchains * i / nthreads); jobs[i].hi = (int)((long)nchains * (i + 1) / nthreads);
        jobs[i].sweeps = sweeps; jobs[i].eps = eps;
        pthread_create(&th[i], NULL, lf_worker, &jobs[i]);
    }
    for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < nchains; ++t) orc_chain_destroy(cs[t]);
    free(cs); free(jobs); free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- CPU baseline: NUTS transitions at a fixed eps (the in-place path of sample_tree, src/NUTS.jl:251-264), one chain per
 * thread at a time; q0 = mu-like start handed in by the caller ([nchains][L]); returns seconds, *steps_out = leapfrogs taken */
typedef struct { orc_chain **chains; int lo, hi, transitions; double eps; long steps; } nuts_job;
static void *nuts_worker(void *arg)
{
    nuts_job *j = (nuts_job *)arg;
    orc_tree_stats st;
    for (int it = 1; it <= j->transitions; ++it)
        for (int t = j->lo; t < j->hi; ++t) { orc_sample_tree(j->chains[t], j->eps, (uint32_t)it, &st); j->steps += st.steps; }
    return NULL;
}
double orc_bench_nuts(const orc_model *model, uint64_t seed, int nchains, int transitions, double eps, const double *minv,
                      const double *q0, int nthreads, long *steps_out)
{
    orc_options o; orc_default_options(&o);
    orc_chain **cs = (orc_chain **)calloc((size_t)nchains, sizeof(orc_chain *));
    for (int t = 0; t < nchains; ++t) {
        cs[t] = orc_chain_create(model, &o, seed, (uint32_t)t);
        if (minv) orc_chain_set_minv(cs[t], minv);
        if (q0) orc_chain_set_q(cs[t], q0 + (size_t)t * (size_t)orc_chain_L(cs[t]));
        else orc_chain_random_position(cs[t]);
    }
    if (nthreads > nchains) nthreads = nchains;
    nuts_job *jobs = (nuts_job *)calloc((size_t)nthreads, sizeof(nuts_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < nthreads; ++i) {
        jobs[i].chains = cs; jobs[i].lo = (int)((long)nchains * i / nthreads); jobs[i].hi = (int)((long)nchains * (i + 1) / nthreads);
        jobs[i].transitions = transitions; jobs[i].eps = eps; jobs[i].steps = 0;
        pthread_create(&th[i], NULL, nuts_worker, &jobs[i]);
    }
    long steps = 0;
    for (int i = 0; i < nthreads; ++i) { pthread_join(th[i], NULL); steps += jobs[i].steps; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < nchains; ++t) orc_chain_destroy(cs[t]);
    free(cs); free(jobs); free(th);
    if (steps_out) *steps_out = steps;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- exports for known-answer tests ------------------------------------- */
double orc_log_export(double x) { return orc_log(x); }
double orc_exp_export(double x) { return orc_exp(x); }
double orc_log1p_export(double x) { return orc_log1p(x); }
void orc_sincos2pi_export(double u, double *s, double *c) { orc_sincos2pi(u, s, c); }
double orc_logaddexp_export(double x, double y) { return orc_logaddexp(x, y); }
void orc_philox_export(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    orc_philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
double orc_dot_export(const double *a, const double *b, int L) { return orc_dot(a, b, L); }
double orc_randexp_export(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t draw) { return orc_randexp(seed, chain, iter, draw); }
uint32_t orc_rand_directions_export(uint64_t seed, uint32_t chain, uint32_t iter) { return orc_rand_directions(seed, chain, iter); }
void orc_randn_export(uint64_t seed, uint32_t chain, uint32_t iter, int L, double *out)
{
    for (int k = 0; k < L / 2; ++k) orc_randn_pair(seed, chain, iter, (uint32_t)k, &out[2 * k], &out[2 * k + 1]);
}
/* biased_progressive_logprob2, src/tree.jl:261-263 */
double orc_logprob2_export(int bias, double w1, double w2)
{
    double w = orc_logaddexp(w1, w2);
    return w2 - (bias ? w1 : w);
}
/* is_turning on explicit vectors, src/NUTS.jl:148-170 */
int orc_is_turning_export(const double *rho, const double *psm, const double *psp, int L)
{
    double dm = orc_dot(rho, psm, L), dp = orc_dot(rho, psp, L);
    return (dm < 0.0) | (dp < 0.0);
}
/* acceptance_rate, src/NUTS.jl:84 */
double orc_acceptance_rate_export(double log_sum_a, int steps)
{
    double a = orc_exp(log_sum_a) / (double)steps;
    return a < 1.0 ? a : 1.0;
}
