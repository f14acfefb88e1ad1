// idhmc_internal.hpp -- host/device shared descriptors and launcher prototypes (not part of the ABI).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "../../include/idhmc.h"

namespace idhmc {

// dual-averaging state, reference DualAveragingState (src/stepsize.jl:196-202), SoA over chains
struct DaArrays {
    double *mu, *Hbar, *logeps, *logeps_bar;
    int64_t *m;
};

// device-side diagnostics (IDHMC_T_ACCUM_DIAG): per chain the running sums of EBFMI, shifted by the first pi; per
// context the integer counters of include/idhmc.h
struct DiagArrays {
    int32_t *n;                                // [C] transitions accumulated
    double *pi1, *prev, *s1, *s2, *d2;         // [C] first pi, previous pi, sum(pi - pi1), sum (pi - pi1)^2, sum (diff pi)^2
    unsigned long long *counters;              // [IDHMC_DIAG_COUNTERS]
};

// everything a kernel needs, passed by value
struct DevState {
    int64_t C;            // chains in this context
    int32_t L, D, nch;    // padded length, dimension, L/128
    int32_t model;        // IDHMC_MODEL_*
    uint32_t k0, k1;      // Philox key = seed
    uint32_t first_chain; // global id of chain 0
    double *q, *p, *g;    // [C][L]
    double *lq, *pi;      // [C]  l(q),  pi = l(q) - K(p)
    double *eps;          // [C]
    double *minv, *w;     // [C][L] (minv_stride = L) or [L] shared (minv_stride = 0)
    int64_t minv_stride;
    const double *mu, *tau;   // [L]
    const double *prec;       // [L][L]
    const double *user_params;   // IDHMC_MODEL_CUSTOM: the user's parameter blob
    int64_t user_nparams;
    const void *jit;             // host only: the hipRTC module of a custom density
    // NUTS
    int32_t max_depth;
    double min_delta;
    idhmc_tree_stats *stats;  // [C]
    uint32_t *directions;     // [C] injected directions
    double *arena;            // per wave slot tree storage
    int64_t arena_stride;     // doubles per slot
    int32_t nslots;           // persistent waves
    uint32_t *queue;          // chain work counter
    uint32_t n_iter;          // transitions per chain of the launch being made (launch_nuts sets it in its copy of the state; 0 / 1: one)
    uint32_t *iters_done;     // [C] n_iter > 1: transitions of this launch each chain has completed (zeroed by launch_nuts)
    double *fz_q;             // this launch's draws leave here, [n_iter][C][D] contiguous rows (null: nobody wants them); set per launch
    idhmc_tree_stats *fz_st;  // and its records, [n_iter][C]
    // adaptation
    DaArrays da;
    double da_delta, da_gamma, da_kappa;
    int32_t da_t0;
    int32_t eps_mode;
    double *da_global;        // [8] global dual-averaging state: mu, m, Hbar, logeps, logeps_bar, eps
    unsigned long long *xchg_acc;   // [3 * kXchgBlocks + 1] per-workgroup integer partials + ticket of the exchange sum (k_xchg_sum)
    // metric window: x1, sum delta, sum delta^2 ([C][L] each), draws in the window [C]
    double *mw_x1, *mw_s1, *mw_s2;
    int32_t *mw_n;
    // running moments: mean, M2 ([C][L]), count [C]
    double *mom_mean, *mom_m2;
    int64_t *mom_n;
    // stepsize search
    double ss_a_min, ss_a_max, ss_eps0, ss_C;
    int32_t ss_maxiter_crossing, ss_maxiter_bisect;
    int32_t *status;          // [C] per-chain error codes from the search / eps underflow
    DiagArrays diag;          // all null until idhmc_diag_reset
    unsigned long long *total_steps;  // [32]: the pulse the host polls = {[0] leapfrog steps, [1] abort code (an IDHMC_ERR_* a
                                      // chain raised: eps underflow)}; [2..9] cycle stamps of the diagnostic build (-DIDHMC_STAMPS)
};
constexpr int kPulseAt = 0;
constexpr int kXchgBlocks = 64;   // workgroups of k_xchg_sum; DevState::xchg_acc holds 3 * kXchgBlocks partials + 1 ticket

#ifndef __HIPCC_RTC__   // host side only (the header is also compiled by hipRTC for custom densities)
// ---- dispatch over the padded length: NCH = L / 128 = ceil(D / 128), every value 1..16 for the separable densities
// (a vector is padded to the next multiple of 128, like the reference pads to the SIMD width, src/mcmc.jl:117);
// the dense density's matrix kernels need a power of two (D <= 1024)
#define IDHMC_NCH_CASE(N, ...) case N: { constexpr int NCH = N; __VA_ARGS__; } break;
#define IDHMC_DISPATCH_NCH(NCHV, ...)                                                                              \
    switch (NCHV) {                                                                                                \
        IDHMC_NCH_CASE(1, __VA_ARGS__) IDHMC_NCH_CASE(2, __VA_ARGS__) IDHMC_NCH_CASE(3, __VA_ARGS__) IDHMC_NCH_CASE(4, __VA_ARGS__)    \
        IDHMC_NCH_CASE(5, __VA_ARGS__) IDHMC_NCH_CASE(6, __VA_ARGS__) IDHMC_NCH_CASE(7, __VA_ARGS__) IDHMC_NCH_CASE(8, __VA_ARGS__)    \
        IDHMC_NCH_CASE(9, __VA_ARGS__) IDHMC_NCH_CASE(10, __VA_ARGS__) IDHMC_NCH_CASE(11, __VA_ARGS__) IDHMC_NCH_CASE(12, __VA_ARGS__) \
        IDHMC_NCH_CASE(13, __VA_ARGS__) IDHMC_NCH_CASE(14, __VA_ARGS__) IDHMC_NCH_CASE(15, __VA_ARGS__) IDHMC_NCH_CASE(16, __VA_ARGS__) \
    default: return hipErrorInvalidValue;                                                                          \
    }
#define IDHMC_DISPATCH_NCH_POW2(NCHV, ...)                                                                         \
    switch (NCHV) {                                                                                                \
        IDHMC_NCH_CASE(1, __VA_ARGS__) IDHMC_NCH_CASE(2, __VA_ARGS__) IDHMC_NCH_CASE(4, __VA_ARGS__) IDHMC_NCH_CASE(8, __VA_ARGS__)    \
    default: return hipErrorInvalidValue;                                                                          \
    }
// ---- custom densities through hipRTC (idhmc_jit.hip) -------------------------------------------------
struct JitModule;
// compiles `source` against the kernel templates for this state's shape; on failure returns non-zero and
// fills `log` (compiler output, truncated)
int jit_build(const DevState &s, const char *source, JitModule **out, char *log, size_t log_cap);
void jit_destroy(JitModule *m);
hipError_t launch_eval_jit(const DevState &s, int random_q, hipStream_t st);
hipError_t launch_leapfrog_jit(const DevState &s, double eps, int own, int n_steps, hipStream_t st);

// ---- RCCL communicator for the global-eps exchange (idhmc_comm.hip; RCCL bound with dlopen) ----------
struct Comm;
int comm_unique_id(void *out128, char *err, size_t cap);
Comm *comm_create(int nranks, int rank, const void *id128, char *err, size_t cap);   // on the current device
int comm_allreduce_sum(Comm *c, double *dev_buf, int n, hipStream_t st, char *err, size_t cap);
void comm_destroy(Comm *c);
void comm_info(const Comm *c, int *nranks, int *rank, long long *allreduces);

// ---- launchers (idhmc_kernels.hip / idhmc_nuts.hip) ------------------------------------------------
hipError_t launch_eval(const DevState &s, hipStream_t st);                 // lq, grad from q
hipError_t launch_random_position(const DevState &s, hipStream_t st);
hipError_t launch_refresh(const DevState &s, uint32_t iter, hipStream_t st);   // p = W randn; pi
hipError_t launch_logdensity(const DevState &s, hipStream_t st);               // pi from (lq, p)
// regrad != 0 (separable densities, single step): the gradient array is neither read nor written and goes stale
hipError_t launch_leapfrog(const DevState &s, double eps, int use_own_eps, int n_steps, int regrad, hipStream_t st);
constexpr int kRowPad = 32;   // rows of slack after q, p, g and a per-chain M^-1 of a dense context (ragged last tile, never stored)
// dense density, matrix-core kernel over the 16-chain tiles [tile_begin, tile_end) with at most max_grid workgroups (0: as many
// as are resident); hipErrorNotSupported outside the kernel's range (idhmc_dense_mfma.hip)
hipError_t launch_leapfrog_dense_mfma_tiles(const DevState &s, double eps, int own, int n_steps, int64_t tile_begin,
                                            int64_t tile_end, int64_t max_grid, hipStream_t st);
int dense_mfma_tile_align(const DevState &s);   // a single-step range must begin at a multiple of this many 16-chain tiles
hipError_t launch_set_w(const DevState &s, hipStream_t st);                    // W = 1/sqrt(M^-1)
hipError_t launch_fill(double *p, double v, int64_t n, hipStream_t st);
hipError_t launch_xcc_probe(uint32_t *out, int grid, hipStream_t st);   // out[b] = XCD id of workgroup b
hipError_t launch_spin(long long ticks_100MHz, hipStream_t st);   // an idle wavefront for that long (hardware-queue discovery)
hipError_t launch_placement_probe(double *const *v, int nvec, int64_t C, int L, hipStream_t st);   // reads and rewrites v[k][0 .. C L)
hipError_t launch_pack_draw(const DevState &s, double *q_out, idhmc_tree_stats *st_out, hipStream_t st);
hipError_t launch_broadcast_row(double *a, int L, int64_t C, hipStream_t st);
hipError_t launch_nuts(const DevState &s, uint32_t iter, uint32_t flags, int wide, hipStream_t st, uint32_t n_iter = 1,
                       double *fz_q = nullptr, idhmc_tree_stats *fz_st = nullptr);
hipError_t launch_stepsize_search(const DevState &s, hipStream_t st);
hipError_t launch_local_optimum(const DevState &s, double penalty, int iterations, hipStream_t st);
hipError_t launch_da_init(const DevState &s, hipStream_t st);
hipError_t launch_da_finalize(const DevState &s, hipStream_t st);
hipError_t launch_xchg_sum(const DevState &s, int kind, double *dev_xchg, hipStream_t st);   // IDHMC_XCHG_* record
hipError_t launch_da_adapt_global(const DevState &s, const double *dev_xchg, hipStream_t st);
hipError_t launch_eps_from_logeps(const DevState &s, const double *dev_xchg, hipStream_t st);
hipError_t launch_metric_update(const DevState &s, double lambda, hipStream_t st);
hipError_t launch_moments_get(const DevState &s, double *mean_out, double *var_out, hipStream_t st);
// pooled metric (IDHMC_METRIC_POOLED): partial sums per segment of global chain ids, added in segment order (idhmc_kernels.hip)
size_t pool_scratch_doubles(int L);
hipError_t launch_pool_partials(const DevState &s, int pass, const double *scratch, double *table, long long seg_lo, long long seg_hi,
                                hipStream_t st);
hipError_t launch_pool_consume(const DevState &s, int pass, double *scratch, const double *table, long long nseg, double lambda,
                               hipStream_t st);
hipError_t launch_status_max(const DevState &s, int32_t *dev_out, hipStream_t st);
hipError_t launch_ebfmi(const DevState &s, double *out, hipStream_t st);
#endif  // !__HIPCC_RTC__

}  // namespace idhmc
