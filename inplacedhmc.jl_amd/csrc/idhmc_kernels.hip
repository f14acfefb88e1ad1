// idhmc_kernels.hip -- streaming kernels: density evaluation, momentum refresh, fused leapfrog,
// dual-averaging / metric / moment bookkeeping.  One chain per wavefront (see idhmc_device.hpp).
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"
#include "idhmc_xchg.hpp"
#include <cstdlib>

namespace idhmc {

static inline int blocks_for(int64_t C, int waves_per_block, int max_blocks)
{
    int64_t b = (C + waves_per_block - 1) / waves_per_block;
    if (b > max_blocks) b = max_blocks;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------------------------------------------
// evaluate_l! for every chain (reference src/kinetic_energy.jl:72-85)
template <int NCH, class Model>
__global__ __launch_bounds__(256) void k_eval(DevState s)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Model mdl;
    mdl.load(s.mu, s.tau, lane);
    for (int64_t c = wave; c < s.C; c += nw) {
        const Vec<NCH> q = vload<NCH>(s.q + c * s.L, lane);
        Vec<NCH> g;
        const double lq = eval_density<NCH>(mdl, q, g);
        vstore<NCH>(s.g + c * s.L, lane, g);
        if (lane == 0) s.lq[c] = lq;
    }
}

// random_position! (reference src/warmup.jl:73): q ~ U[-2,2)^D, then evaluate
template <int NCH, class Model>
__global__ __launch_bounds__(256) void k_random_position(DevState s)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Model mdl;
    mdl.load(s.mu, s.tau, lane);
    for (int64_t c = wave; c < s.C; c += nw) {
        const RngKey key{s.k0, s.k1, s.first_chain + (uint32_t)c};
        Vec<NCH> q, g;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int pair = j * 64 + lane;
            const u32x4 x = rng_draw(key, 0u, kStreamInitQ, (uint32_t)pair);
            const double u0 = u01(x.x, x.y), u1 = u01(x.z, x.w);
            q.c[j].x = (2 * pair < s.D) ? dfma(4.0, u0, -2.0) : 0.0;
            q.c[j].y = (2 * pair + 1 < s.D) ? dfma(4.0, u1, -2.0) : 0.0;
        }
        const double lq = eval_density<NCH>(mdl, q, g);
        vstore<NCH>(s.q + c * s.L, lane, q);
        vstore<NCH>(s.g + c * s.L, lane, g);
        if (lane == 0) s.lq[c] = lq;
    }
}

// rand_p! (reference src/kinetic_energy.jl:63) + pi = logdensity(H, z) (:107-112)
template <int NCH>
__global__ __launch_bounds__(256) void k_refresh(DevState s, uint32_t iter, int draw)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < s.C; c += nw) {
        const Vec<NCH> minv = vload<NCH>(s.minv + c * s.minv_stride, lane);
        Vec<NCH> p;
        if (draw) {
            const RngKey key{s.k0, s.k1, s.first_chain + (uint32_t)c};
            const Vec<NCH> w = vload<NCH>(s.w + c * s.minv_stride, lane);
            p = rand_momentum<NCH>(key, iter, w, lane, s.D);
            vstore<NCH>(s.p + c * s.L, lane, p);
        } else {
            p = vload<NCH>(s.p + c * s.L, lane);
        }
        const double K = kinetic_energy<NCH>(minv, p);
        if (lane == 0) s.pi[c] = phase_logdensity(s.lq[c], K);
    }
}

// Fused leapfrog (reference src/kinetic_energy.jl:126-163), n_steps steps per launch.
// HBM traffic per chain and launch: read q, p, grad, write q', p', grad' = 6*L*8 bytes
// (M^-1, mu, tau are L2-resident: 3*L*8 bytes shared by all chains).
template <int NCH, class Model>
__global__ __launch_bounds__(256) void k_leapfrog(DevState s, double eps_arg, int own_eps, int n_steps)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Model mdl;
    mdl.load(s.mu, s.tau, lane);
    for (int64_t c = wave; c < s.C; c += nw) {
        const int64_t off = c * s.L;
        Vec<NCH> q = vload<NCH>(s.q + off, lane);
        Vec<NCH> p = vload<NCH>(s.p + off, lane);
        Vec<NCH> g = vload<NCH>(s.g + off, lane);
        const Vec<NCH> minv = vload<NCH>(s.minv + c * s.minv_stride, lane);
        const double eps = own_eps ? s.eps[c] : eps_arg;
        double lq = 0.0, K = 0.0;
        for (int it = 0; it < n_steps; ++it) leapfrog_step<NCH>(mdl, minv, eps, q, p, g, lq, K);
        vstore<NCH>(s.q + off, lane, q);
        vstore<NCH>(s.p + off, lane, p);
        vstore<NCH>(s.g + off, lane, g);
        if (lane == 0) {
            s.lq[c] = lq;
            s.pi[c] = phase_logdensity(lq, K);
        }
    }
}

// Single-step form of the same kernel, the HBM-bound headline path (BASELINE.json configs[1]).
// VAR bit 0: issue every load of the chain (q, p, grad: 3 KiB per chunk from HBM; M^-1, mu, tau from L2)
//            before the first store, so one wave keeps 24 KiB of HBM reads in flight (2 waves/SIMD);
//            otherwise chunk-by-chunk at 4 waves/SIMD.
// VAR bit 1: non-temporal loads/stores for the streamed state.
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool NT> IDHMC_DEV v2d ld2(const v2d *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> IDHMC_DEV void st2(v2d *p, v2d v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int NCH, class Model, int VAR>
__global__ __launch_bounds__(256, (VAR & 1) ? 2 : 4) void k_leapfrog1(DevState s, double eps_arg, int own_eps)
{
    // VAR bit 2 (REGRAD): the gradient stream is dropped -- grad l(q) is re-derived from q (2 flops per element, the
    // very bits the previous step would have stored) and grad l(q') is not written: 4 D 8 bytes per step instead of 6 D 8
    constexpr bool PRE = (VAR & 1) != 0, NT = (VAR & 2) != 0, REGRAD = (VAR & 4) != 0;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const v2d *__restrict__ mu2 = reinterpret_cast<const v2d *>(s.mu) + lane;
    const v2d *__restrict__ tau2 = reinterpret_cast<const v2d *>(s.tau) + lane;
    for (int64_t c = wave; c < s.C; c += nw) {
        const int64_t off = c * s.L;
        v2d *__restrict__ q2 = reinterpret_cast<v2d *>(s.q + off) + lane;
        v2d *__restrict__ p2 = reinterpret_cast<v2d *>(s.p + off) + lane;
        v2d *__restrict__ g2 = reinterpret_cast<v2d *>(s.g + off) + lane;
        const v2d *__restrict__ m2 = reinterpret_cast<const v2d *>(s.minv + c * s.minv_stride) + lane;
        const double eps = own_eps ? s.eps[c] : eps_arg;
        const double eh = 0.5 * eps;
        double l0 = 0.0, l1 = 0.0, k0 = 0.0, k1 = 0.0;
        v2d qv[NCH], pv[NCH], gv[NCH], mvv[NCH], muv[NCH], tav[NCH];
        if (PRE) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                qv[j] = ld2<NT>(q2 + j * 64);
                pv[j] = ld2<NT>(p2 + j * 64);
                if (!REGRAD) gv[j] = ld2<NT>(g2 + j * 64);
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                mvv[j] = m2[j * 64];
                if (Model::kHasParams) { muv[j] = mu2[j * 64]; tav[j] = tau2[j * 64]; }
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            v2d q, p, g, mv, mu = {0.0, 0.0}, tau = {1.0, 1.0};
            if (PRE) {
                q = qv[j]; p = pv[j]; mv = mvv[j];
                if (!REGRAD) g = gv[j];
                if (Model::kHasParams) { mu = muv[j]; tau = tav[j]; }
            } else {
                q = ld2<NT>(q2 + j * 64); p = ld2<NT>(p2 + j * 64);
                if (!REGRAD) g = ld2<NT>(g2 + j * 64);
                mv = m2[j * 64];
                if (Model::kHasParams) { mu = mu2[j * 64]; tau = tau2[j * 64]; }
            }
            if (REGRAD) g = v2d{-(tau.x * (q.x - mu.x)), -(tau.y * (q.y - mu.y))};
            const double pmx = dfma(eh, g.x, p.x), pmy = dfma(eh, g.y, p.y);
            const double qx = dfma(eps * mv.x, pmx, q.x), qy = dfma(eps * mv.y, pmy, q.y);
            const double dx = qx - mu.x, dy = qy - mu.y;
            const double tx = tau.x * dx, ty = tau.y * dy;
            l0 = dfma(tx, dx, l0);
            l1 = dfma(ty, dy, l1);
            const double px = dfma(eh, -tx, pmx), py = dfma(eh, -ty, pmy);
            k0 = dfma(px * mv.x, px, k0);
            k1 = dfma(py * mv.y, py, k1);
            st2<NT>(q2 + j * 64, v2d{qx, qy});
            st2<NT>(p2 + j * 64, v2d{px, py});
            if (!REGRAD) st2<NT>(g2 + j * 64, v2d{-tx, -ty});
        }
        double sl, sk;
        wave_sum2(l0, l1, k0, k1, sl, sk);
        double lq = -0.5 * sl;
        lq = dfinite(lq) ? lq : -kInf;
        if (lane == 0) {
            s.lq[c] = lq;
            s.pi[c] = phase_logdensity(lq, 0.5 * sk);
        }
    }
}

// W = 1/sqrt(M^-1) (GaussianKineticEnergy, reference src/hamiltonian.jl:50-57)
__global__ void k_set_w(double *w, const double *minv, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = 1.0 / __builtin_sqrt(minv[i]);
}
// copy row 0 of a [C][L] array into rows 1..C-1
__global__ void k_broadcast_row(double *a, int L, int64_t C)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (C - 1) * L) return;
    a[L + i] = a[i % L];
}
// the draw of every chain ([C][L] padded rows -> [C][D] contiguous) and its record into a staging buffer: the host copy of
// transition n then runs while transition n + 1 computes (idhmc_api.hip, fetch pipeline)
__global__ void k_pack_draw(DevState s, double *q_out, idhmc_tree_stats *st_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = s.C * s.D;
    if (q_out && i < n) {
        const int64_t c = i / s.D;
        q_out[i] = s.q[c * s.L + (i - c * s.D)];
    }
    if (st_out && i < s.C) st_out[i] = s.stats[i];
}
__global__ void k_fill(double *p, double v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// initial_adaptation_state (reference src/stepsize.jl:208-212); eps <- current_eps = exp(log eps) (:235)
__global__ void k_da_init(DevState s)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.C) return;
    const double le = dlog(s.eps[c]);
    s.da.mu[c] = dlog(10.0) + le;
    s.da.m[c] = 0;
    s.da.Hbar[c] = 0.0;
    s.da.logeps[c] = le;
    s.da.logeps_bar[c] = 0.0;
    s.eps[c] = dexp(le);
}
// global mode: one state for all chains, started from chain 0's eps (all chains hold the same value)
__global__ void k_da_init_global(DevState s)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double le = dlog(s.eps[0]);
        s.da_global[0] = dlog(10.0) + le;
        s.da_global[1] = 0.0;
        s.da_global[2] = 0.0;
        s.da_global[3] = le;
        s.da_global[4] = 0.0;
        s.da_global[5] = dexp(le);
    }
}
// final_eps = exp(logeps_bar) (reference src/stepsize.jl:241)
__global__ void k_da_finalize(DevState s)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.C) return;
    s.eps[c] = (s.eps_mode == IDHMC_EPS_GLOBAL) ? dexp(s.da_global[4]) : dexp(s.da.logeps_bar[c]);
}

// The exchange record of a per-chain statistic (include/idhmc.h, idhmc_xchg.hpp): fixed-point limbs summed as INTEGERS,
// so neither the order of this reduction nor that of the all-reduce behind it can change a bit of the result.
// KIND = IDHMC_XCHG_ACCEPT: the last transition's acceptance rates; IDHMC_XCHG_LOGEPS: log of each chain's eps.
template <int KIND>
__global__ __launch_bounds__(1024) void k_xchg_sum(DevState s, double *out4)
{
    // Up to kXchgBlocks workgroups of 1024 threads (one chain per thread at configs[1]), integer partial sums: inside a workgroup
    // through LDS, across workgroups through a table of per-workgroup partials in s.xchg_acc -- the workgroup that draws the last
    // ticket (ONE atomic per workgroup: contended atomics on one address cost ~15 ns each, 1024 of them were the whole 17 us of
    // the first multi-block form) adds the table in index order and writes the record.  Integer addition is associative, so the
    // record is exact and the same in any order anyway.  xchg_acc = [3 * kXchgBlocks partials][ticket], ticket left at zero.
    __shared__ long long sh[3][16];
    long long v[3] = {0, 0, 0};                      // hi, lo, chains with a pending status
    for (int64_t c = (int64_t)blockIdx.x * 1024 + threadIdx.x; c < s.C; c += (int64_t)gridDim.x * 1024) {
        if (KIND != IDHMC_XCHG_STATUS) {
            const double x = (KIND == IDHMC_XCHG_ACCEPT) ? s.stats[c].acceptance_rate : dlog(s.eps[c]);
            long long h, l;
            xchg_limbs(KIND, x, h, l);
            v[0] += h; v[1] += l;
        }
        v[2] += s.status[c] != 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
        if (lane == 0) sh[k][wave] = v[k];
    }
    __syncthreads();
    if (wave != 0) return;                           // the rest is one wavefront's work
    for (int k = 0; k < 3; ++k) {
        v[k] = lane < 16 ? sh[k][lane] : 0;
        for (int o = 8; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    }
    unsigned long long *acc = s.xchg_acc;
    int last = 0;
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) __hip_atomic_store(acc + 3 * blockIdx.x + k, (unsigned long long)v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(acc + 3 * kXchgBlocks, 1ull) == (unsigned long long)gridDim.x - 1ull;
    }
    if (!__shfl(last, 0)) return;
    __threadfence();
    // the last workgroup: lane b reads workgroup b's partials (kXchgBlocks <= 64: one load per lane and value, all in flight together)
    for (int k = 0; k < 3; ++k) {
        v[k] = (unsigned)lane < gridDim.x ? (long long)__hip_atomic_load(acc + 3 * lane + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if (lane == 0) {
        out4[0] = (double)v[0];
        out4[1] = (double)v[1];
        out4[2] = (double)s.C;
        out4[3] = (double)v[2];
        __hip_atomic_store(acc + 3 * kXchgBlocks, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
static_assert(kXchgBlocks <= 64, "the finisher reads one workgroup's partials per lane");
// adapt_stepsize (reference src/stepsize.jl:220-229) on the pooled mean acceptance
__global__ void k_da_adapt_global(DevState s, const double *xchg)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double a = xchg_mean(IDHMC_XCHG_ACCEPT, xchg[0], xchg[1], xchg[2]);
        double mu = s.da_global[0], m = s.da_global[1], Hbar = s.da_global[2], lb = s.da_global[4];
        m += 1.0;
        Hbar += (s.da_delta - a - Hbar) / (m + (double)s.da_t0);
        const double le = mu - __builtin_sqrt(m) / s.da_gamma * Hbar;
        lb += dexp(-s.da_kappa * dlog(m)) * (le - lb);
        s.da_global[1] = m;
        s.da_global[2] = Hbar;
        s.da_global[3] = le;
        s.da_global[4] = lb;
        s.da_global[5] = dexp(le);
    }
}
// global mode after the per-chain searches: one eps for everybody, exp(pooled mean of log eps)
__global__ void k_eps_from_logeps(DevState s, const double *xchg)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.C) return;
    s.eps[c] = dexp(xchg_mean(IDHMC_XCHG_LOGEPS, xchg[0], xchg[1], xchg[2]));
}
__global__ void k_eps_from_global(DevState s)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.C) return;
    const double e = s.da_global[5];
    s.eps[c] = e;
    if (e < 1e-10) {
        s.status[c] = IDHMC_ERR_EPS_UNDERFLOW;
        if (c == 0) atomicMax(s.total_steps + 1, (unsigned long long)IDHMC_ERR_EPS_UNDERFLOW);   // the host's pulse
    }
}

// GaussianKineticEnergy!(kappa, chain, lambda) (reference src/hamiltonian.jl:119-189) from the running
// window sums {x1, sum(x-x1), sum(x-x1)^2} kept by the transition kernel
__global__ void k_metric_update(DevState s, double lambda)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.C * s.L) return;
    const int64_t c = i / s.L;
    const int d = (int)(i - c * s.L);
    double mv = 1.0, wv = 1.0;
    if (d < s.D) {
        const double N = (double)s.mw_n[c];
        const double Ninv = 1.0 / N;                                  // :156
        const double mulreg = N / ((N + lambda) * (N - 1.0));         // :157
        const double addreg = 1e-3 * lambda / (N + lambda);           // :158
        const double s1 = s.mw_s1[i], s2 = s.mw_s2[i];
        const double s2nm1 = dfma(-(s1 * s1), Ninv, s2);              // :94-95
        mv = dfma(s2nm1, mulreg, addreg);                             // :96
        wv = 1.0 / __builtin_sqrt(mv);                                // :97
    }
    s.minv[i] = mv;
    s.w[i] = wv;
}
// ---- pooled metric (IDHMC_METRIC_POOLED): one M^-1 for all chains, estimated from every chain's window --------------
// Many chains sample the same posterior, so their windows are pooled: two passes over the per-chain running sums
// {n, x1, sum(x - x1), sum(x - x1)^2}:
//   pass 0: N = sum n_c,  A_d = sum_c (n_c x1 + s1)                      -> mean_d = A_d / N
//   pass 1: S_d = sum_c [ (s2 - s1^2 / n_c) + n_c (m_c - mean_d)^2 ],  m_c = x1 + s1 / n_c
// then the reference's regularisation with the pooled count (src/hamiltonian.jl:156-158,94-97).
// So that the result does not depend on how the chains are sharded, the column sums are taken over SEGMENTS OF GLOBAL CHAIN
// IDS (IDHMC_POOL_SEGMENT chains each): a segment's partial sum runs over its chains in ascending id, and the partials are
// added in ascending segment order.  Ranks exchange the [segments][L + 1] table of partials with a SUM all-reduce in which
// every entry is non-zero on one rank only (shards aligned to the segment size) -- adding zeros is exact in any order -- and
// every rank then adds the same table in the same order.
__global__ void k_pool_partial(DevState s, int pass, const double *mean, double *table, long long seg_lo, long long seg_hi)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const long long seg = seg_lo + blockIdx.y;
    if (d > s.L || seg >= seg_hi) return;
    const long long g0 = seg * IDHMC_POOL_SEGMENT, g1 = g0 + IDHMC_POOL_SEGMENT;         // global chain ids of the segment
    long long c0 = g0 - (long long)s.first_chain, c1 = g1 - (long long)s.first_chain;    // local indices, clipped to this context
    c0 = c0 < 0 ? 0 : c0;
    c1 = c1 > s.C ? s.C : c1;
    double acc = 0.0;
    if (d < s.L) {
        const double mu = pass ? mean[d] : 0.0;
        for (long long c = c0; c < c1; ++c) {
            const double n = (double)s.mw_n[c];
            if (!(n > 0.0)) continue;
            const int64_t i = c * s.L + d;
            const double x1 = s.mw_x1[i], s1 = s.mw_s1[i], s2 = s.mw_s2[i];
            if (pass == 0) {
                acc += dfma(n, x1, s1);
            } else {
                const double m = x1 + s1 / n, dm = m - mu;
                acc += dfma(-(s1 * s1), 1.0 / n, s2) + n * (dm * dm);
            }
        }
    } else {                                    // column L: the number of draws (an integer: exact)
        for (long long c = c0; c < c1; ++c) acc += (double)s.mw_n[c];
    }
    table[(seg - seg_lo) * (s.L + 1) + d] = acc;
}
// out[0..L] = column sums of the table in ascending segment order
__global__ void k_pool_finish(DevState s, const double *table, long long nseg, double *out)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > s.L) return;
    double acc = 0.0;
    for (long long g = 0; g < nseg; ++g) acc += table[g * (s.L + 1) + d];
    out[d] = acc;
}
__global__ void k_pool_mean(DevState s, const double *acc0, double *mean)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < s.L) mean[d] = acc0[d] / acc0[s.L];
}
__global__ void k_pool_apply(DevState s, const double *acc0, const double *acc1, double lambda)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= s.L) return;
    double mv = 1.0, wv = 1.0;
    if (d < s.D) {
        const double N = acc0[s.L];
        const double mulreg = N / ((N + lambda) * (N - 1.0));         // src/hamiltonian.jl:157
        const double addreg = 1e-3 * lambda / (N + lambda);           // :158
        mv = dfma(acc1[d], mulreg, addreg);                           // :96
        wv = 1.0 / __builtin_sqrt(mv);                                // :97
    }
    s.minv[d] = mv;
    s.w[d] = wv;
}
__global__ void k_moments_get(DevState s, double *mean, double *var)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.C * s.L) return;
    const int64_t c = i / s.L;
    const double n = (double)s.mom_n[c];
    mean[i] = s.mom_mean[i];
    var[i] = n > 1.0 ? s.mom_m2[i] / (n - 1.0) : 0.0;
}
// EBFMI per chain from the running sums (reference src/diagnostics.jl:28-32): mean(abs2, diff(pi)) / var(pi)
__global__ void k_ebfmi(DevState s, double *out)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.C) return;
    const double n = (double)s.diag.n[c];
    const double s1 = s.diag.s1[c];
    const double var = dfma(-(s1 * s1), 1.0 / n, s.diag.s2[c]) / (n - 1.0);
    out[c] = (s.diag.d2[c] / (n - 1.0)) / var;
}
__global__ void k_status_max(DevState s, int32_t *out)
{
    __shared__ int sh[256];
    int m = 0;
    for (int64_t c = threadIdx.x; c < s.C; c += 256) m = max(m, s.status[c]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ---- launchers ---------------------------------------------------------------------------------------
#define IDHMC_LAUNCH_SEPARABLE(KERNEL, GRID, ...)                                                     \
    IDHMC_DISPATCH_NCH(s.nch, {                                                                       \
        if (s.model == IDHMC_MODEL_ISO_GAUSSIAN)                                                      \
            hipLaunchKernelGGL((KERNEL<NCH, IsoGaussian<NCH>>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); \
        else if (s.model == IDHMC_MODEL_DIAG_GAUSSIAN)                                                \
            hipLaunchKernelGGL((KERNEL<NCH, DiagGaussian<NCH>>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); \
        else                                                                                          \
            return hipErrorNotSupported;                                                              \
    })

// streaming kernels: 4 chains per 256-thread block, capped so that every CU keeps several blocks
static constexpr int kMaxStreamBlocks = 256 * 16;

hipError_t launch_eval_dense(const DevState &s, hipStream_t st);
hipError_t launch_leapfrog_dense(const DevState &s, double eps, int own, int n_steps, hipStream_t st);
hipError_t launch_random_position_dense(const DevState &s, hipStream_t st);

hipError_t launch_eval(const DevState &s, hipStream_t st)
{
    if (s.model == IDHMC_MODEL_DENSE_MVN) return launch_eval_dense(s, st);
    if (s.model == IDHMC_MODEL_CUSTOM) return launch_eval_jit(s, 0, st);
    const int grid = blocks_for(s.C, 4, kMaxStreamBlocks);
    IDHMC_LAUNCH_SEPARABLE(k_eval, grid, s);
    return hipGetLastError();
}
hipError_t launch_random_position(const DevState &s, hipStream_t st)
{
    if (s.model == IDHMC_MODEL_DENSE_MVN) return launch_random_position_dense(s, st);
    if (s.model == IDHMC_MODEL_CUSTOM) return launch_eval_jit(s, 1, st);
    const int grid = blocks_for(s.C, 4, kMaxStreamBlocks);
    IDHMC_LAUNCH_SEPARABLE(k_random_position, grid, s);
    return hipGetLastError();
}
hipError_t launch_refresh(const DevState &s, uint32_t iter, hipStream_t st)
{
    const int grid = blocks_for(s.C, 4, kMaxStreamBlocks);
    IDHMC_DISPATCH_NCH(s.nch, hipLaunchKernelGGL((k_refresh<NCH>), dim3(grid), dim3(256), 0, st, s, iter, 1));
    return hipGetLastError();
}
hipError_t launch_logdensity(const DevState &s, hipStream_t st)
{
    const int grid = blocks_for(s.C, 4, kMaxStreamBlocks);
    IDHMC_DISPATCH_NCH(s.nch, hipLaunchKernelGGL((k_refresh<NCH>), dim3(grid), dim3(256), 0, st, s, 0u, 0));
    return hipGetLastError();
}
static int leapfrog_blocks(int64_t C)
{
    int cap = 0;
    if (const char *e = getenv("IDHMC_LF_BLOCKS")) cap = atoi(e);
    return blocks_for(C, 4, cap > 0 ? cap : (1 << 30));
}
hipError_t launch_leapfrog(const DevState &s, double eps, int own, int n_steps, int regrad, hipStream_t st)
{
    if (s.model == IDHMC_MODEL_DENSE_MVN) return launch_leapfrog_dense(s, eps, own, n_steps, st);
    if (s.model == IDHMC_MODEL_CUSTOM) return launch_leapfrog_jit(s, eps, own, n_steps, st);
    if (n_steps == 1) {
        const int grid = leapfrog_blocks(s.C);
        // measured on MI355X (tools/tune_leapfrog.py, 65 536 chains x 1024): diag 6.03 TB/s with 3, iso 5.89 TB/s with 2
        int var = (s.model == IDHMC_MODEL_ISO_GAUSSIAN) ? 2 : 3;
        if (const char *e = getenv("IDHMC_LF_VARIANT")) var = 2 | (atoi(e) & 1);
        if (regrad) var = 7;
        if (s.nch > 8) var = regrad ? 6 : 2;      // L = 2048: preloading the whole chain would spill, chunk by chunk instead
#define IDHMC_LF1(V)                                                                                          \
    IDHMC_DISPATCH_NCH(s.nch, {                                                                               \
        if (s.model == IDHMC_MODEL_ISO_GAUSSIAN)                                                              \
            hipLaunchKernelGGL((k_leapfrog1<NCH, IsoGaussian<NCH>, V>), dim3(grid), dim3(256), 0, st, s, eps, own); \
        else                                                                                                  \
            hipLaunchKernelGGL((k_leapfrog1<NCH, DiagGaussian<NCH>, V>), dim3(grid), dim3(256), 0, st, s, eps, own); \
    })
        switch (var) {      // (variants 0 and 1, temporal loads, lost the round-1 sweep and are no longer built)
        case 2: IDHMC_LF1(2); break;
        case 6: IDHMC_LF1(6); break;
        case 7: IDHMC_LF1(7); break;
        default: IDHMC_LF1(3); break;
        }
        return hipGetLastError();
    }
    const int grid = blocks_for(s.C, 4, kMaxStreamBlocks);
    IDHMC_LAUNCH_SEPARABLE(k_leapfrog, grid, s, eps, own, n_steps);
    return hipGetLastError();
}
hipError_t launch_set_w(const DevState &s, hipStream_t st)
{
    const int64_t n = s.minv_stride ? s.C * s.L : (int64_t)s.L;
    hipLaunchKernelGGL(k_set_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.w, s.minv, n);
    return hipGetLastError();
}
hipError_t launch_broadcast_row(double *a, int L, int64_t C, hipStream_t st)
{
    const int64_t n = (C - 1) * L;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_broadcast_row, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, L, C);
    return hipGetLastError();
}
hipError_t launch_pack_draw(const DevState &s, double *q_out, idhmc_tree_stats *st_out, hipStream_t st)
{
    const int64_t n = q_out ? s.C * s.D : s.C;
    hipLaunchKernelGGL(k_pack_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, q_out, st_out);
    return hipGetLastError();
}
// Placement probe (idhmc_create): the access pattern of the single-step leapfrog on nvec state arrays -- one chain of L doubles per
// wavefront, 16 bytes per lane and chunk, every array read and written in place, element i of all of them at the same time.
__global__ __launch_bounds__(256) void k_placement_probe(double *a0, double *a1, double *a2, double *a3, int nvec, int64_t C, int L)
{
    const int lane = threadIdx.x & 63;
    const int64_t chain = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (chain >= C) return;
    double *v[4] = {a0, a1, a2, a3};
    for (int j = 0; j < L / 128; ++j) {
        double2 x[4];
        for (int k = 0; k < nvec; ++k) x[k] = reinterpret_cast<const double2 *>(v[k] + chain * L)[j * 64 + lane];
        for (int k = 0; k < nvec; ++k) reinterpret_cast<double2 *>(v[k] + chain * L)[j * 64 + lane] = x[k];
    }
}
hipError_t launch_placement_probe(double *const *v, int nvec, int64_t C, int L, hipStream_t st)
{
    hipLaunchKernelGGL(k_placement_probe, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, st, v[0], v[1], nvec > 2 ? v[2] : nullptr,
                       nvec > 3 ? v[3] : nullptr, nvec, C, L);
    return hipGetLastError();
}
// one wavefront that does nothing for `ticks` of the 100 MHz wall clock: two of these on two streams take once or twice that,
// which tells whether the streams share a hardware queue (idhmc_api.hip, pick_lane_streams)
__global__ void k_spin(long long ticks, unsigned long long *sink)
{
    const long long t0 = wall_clock64();
    long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (ticks < 0) *sink = (unsigned long long)t;
}
// which XCD each workgroup of a grid of that size runs on (HW_REG_XCC_ID); several transitions per launch (idhmc_nuts_kernel.hpp) hand a
// chain's state from one workgroup to another through the L2 that workgroups b and b + 8 share -- idhmc_create verifies that they do
__global__ void k_xcc_probe(uint32_t *out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)__builtin_amdgcn_s_getreg(6164) & 15u;      // hwreg(HW_REG_XCC_ID, 0, 4)
}
hipError_t launch_xcc_probe(uint32_t *out, int grid, hipStream_t st)
{
    hipLaunchKernelGGL(k_xcc_probe, dim3(grid), dim3(64), 0, st, out);
    return hipGetLastError();
}
hipError_t launch_spin(long long ticks, hipStream_t st)
{
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks, (unsigned long long *)nullptr);
    return hipGetLastError();
}
hipError_t launch_fill(double *p, double v, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, v, n);
    return hipGetLastError();
}
__global__ void k_eps_from_global(DevState s);
hipError_t launch_da_init(const DevState &s, hipStream_t st)
{
    if (s.eps_mode == IDHMC_EPS_GLOBAL) {
        hipLaunchKernelGGL(k_da_init_global, dim3(1), dim3(64), 0, st, s);
        hipLaunchKernelGGL(k_eps_from_global, dim3((unsigned)((s.C + 255) / 256)), dim3(256), 0, st, s);
    } else {
        hipLaunchKernelGGL(k_da_init, dim3((unsigned)((s.C + 255) / 256)), dim3(256), 0, st, s);
    }
    return hipGetLastError();
}
hipError_t launch_da_finalize(const DevState &s, hipStream_t st)
{
    hipLaunchKernelGGL(k_da_finalize, dim3((unsigned)((s.C + 255) / 256)), dim3(256), 0, st, s);
    return hipGetLastError();
}
hipError_t launch_xchg_sum(const DevState &s, int kind, double *dev_xchg, hipStream_t st)
{
    const int64_t nb = (s.C + 1023) / 1024;
    const dim3 grid((unsigned)(nb < kXchgBlocks ? nb : kXchgBlocks));
    if (kind == IDHMC_XCHG_ACCEPT) hipLaunchKernelGGL(k_xchg_sum<IDHMC_XCHG_ACCEPT>, grid, dim3(1024), 0, st, s, dev_xchg);
    else if (kind == IDHMC_XCHG_LOGEPS) hipLaunchKernelGGL(k_xchg_sum<IDHMC_XCHG_LOGEPS>, grid, dim3(1024), 0, st, s, dev_xchg);
    else hipLaunchKernelGGL(k_xchg_sum<IDHMC_XCHG_STATUS>, grid, dim3(1024), 0, st, s, dev_xchg);
    return hipGetLastError();
}
hipError_t launch_da_adapt_global(const DevState &s, const double *dev_xchg, hipStream_t st)
{
    hipLaunchKernelGGL(k_da_adapt_global, dim3(1), dim3(64), 0, st, s, dev_xchg);
    hipLaunchKernelGGL(k_eps_from_global, dim3((unsigned)((s.C + 255) / 256)), dim3(256), 0, st, s);
    return hipGetLastError();
}
hipError_t launch_eps_from_logeps(const DevState &s, const double *dev_xchg, hipStream_t st)
{
    hipLaunchKernelGGL(k_eps_from_logeps, dim3((unsigned)((s.C + 255) / 256)), dim3(256), 0, st, s, dev_xchg);
    return hipGetLastError();
}
hipError_t launch_metric_update(const DevState &s, double lambda, hipStream_t st)
{
    const int64_t n = s.C * s.L;
    hipLaunchKernelGGL(k_metric_update, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, lambda);
    return hipGetLastError();
}
// small scratch of the pooled metric: [L + 1 acc0][L + 1 acc1][L mean]
size_t pool_scratch_doubles(int L) { return 2 * (size_t)(L + 1) + L; }
hipError_t launch_pool_partials(const DevState &s, int pass, const double *scratch, double *table, long long seg_lo, long long seg_hi,
                                hipStream_t st)
{
    const double *mean = scratch + 2 * (size_t)(s.L + 1);
    const long long nseg = seg_hi - seg_lo;
    if (nseg < 1 || nseg > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pool_partial, dim3((unsigned)((s.L + 256) / 256), (unsigned)nseg), dim3(256), 0, st, s, pass, mean, table, seg_lo, seg_hi);
    return hipGetLastError();
}
// pass 0: acc0 and the mean; pass 1: acc1 and the metric
hipError_t launch_pool_consume(const DevState &s, int pass, double *scratch, const double *table, long long nseg, double lambda,
                               hipStream_t st)
{
    double *acc0 = scratch, *acc1 = scratch + (s.L + 1), *mean = scratch + 2 * (size_t)(s.L + 1);
    const unsigned g1 = (unsigned)((s.L + 256) / 256), g0 = (unsigned)((s.L + 255) / 256);
    hipLaunchKernelGGL(k_pool_finish, dim3(g1), dim3(256), 0, st, s, table, nseg, pass ? acc1 : acc0);
    if (pass == 0) hipLaunchKernelGGL(k_pool_mean, dim3(g0), dim3(256), 0, st, s, acc0, mean);
    else hipLaunchKernelGGL(k_pool_apply, dim3(g0), dim3(256), 0, st, s, acc0, acc1, lambda);
    return hipGetLastError();
}
hipError_t launch_moments_get(const DevState &s, double *mean_out, double *var_out, hipStream_t st)
{
    const int64_t n = s.C * s.L;
    hipLaunchKernelGGL(k_moments_get, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, mean_out, var_out);
    return hipGetLastError();
}
hipError_t launch_ebfmi(const DevState &s, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_ebfmi, dim3((unsigned)((s.C + 255) / 256)), dim3(256), 0, st, s, out);
    return hipGetLastError();
}
hipError_t launch_status_max(const DevState &s, int32_t *dev_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_status_max, dim3(1), dim3(256), 0, st, s, dev_out);
    return hipGetLastError();
}

}  // namespace idhmc
