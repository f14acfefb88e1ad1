// idhmc_dense.hip -- dense multivariate-normal density (BASELINE.json configs[3]): evaluation, fused
// leapfrog and initial-stepsize search with the general (non-separable) density form, one chain per
// wavefront; the gradient is a per-wave GEMV streaming the symmetric precision matrix from L2
// (DenseMvn::grad, idhmc_device.hpp).  The NUTS transition for this density is k_nuts<.., DenseMvn, ..>
// in idhmc_nuts.hip.
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"
#include <cstdlib>

namespace idhmc {

constexpr int kDenseWaves = 4;

template <int NCH>
IDHMC_DEV DenseMvn<NCH> make_dense(const DevState &s, double *dbuf, int lane)
{
    DenseMvn<NCH> m;
    m.prec = s.prec;
    m.mu2 = reinterpret_cast<const double2 *>(s.mu) + lane;
    m.dbuf = dbuf;
    m.D = s.D;
    m.lane = lane;
    return m;
}

#define IDHMC_DENSE_PROLOGUE                                                              \
    __shared__ __attribute__((aligned(16))) double dshare[kDenseWaves][128 * NCH];        \
    const int lane = threadIdx.x & 63;                                                    \
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;           \
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;                            \
    const DenseMvn<NCH> mdl = make_dense<NCH>(s, dshare[threadIdx.x >> 6], lane)

// evaluate_l! (src/kinetic_energy.jl:72-85); mode 1: q ~ U[-2,2) first (random_position!, src/warmup.jl:73)
template <int NCH>
__global__ __launch_bounds__(kDenseWaves * 64) void k_eval_dense(DevState s, int random_q)
{
    IDHMC_DENSE_PROLOGUE;
    for (int64_t c = wave; c < s.C; c += nw) {
        Vec<NCH> q, g;
        if (random_q) {
            const RngKey key{s.k0, s.k1, s.first_chain + (uint32_t)c};
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int pair = j * 64 + lane;
                const u32x4 x = rng_draw(key, 0u, kStreamInitQ, (uint32_t)pair);
                const double u0 = u01(x.x, x.y), u1 = u01(x.z, x.w);
                q.c[j].x = (2 * pair < s.D) ? dfma(4.0, u0, -2.0) : 0.0;
                q.c[j].y = (2 * pair + 1 < s.D) ? dfma(4.0, u1, -2.0) : 0.0;
            }
            vstore<NCH>(s.q + c * s.L, lane, q);
        } else {
            q = vload<NCH>(s.q + c * s.L, lane);
        }
        const double lq = mdl.grad(q, g);
        vstore<NCH>(s.g + c * s.L, lane, g);
        if (lane == 0) s.lq[c] = lq;
    }
}

// leapfrog (src/kinetic_energy.jl:126-163), n_steps per launch, state in registers
template <int NCH>
__global__ __launch_bounds__(kDenseWaves * 64) void k_leapfrog_dense(DevState s, double eps_arg, int own_eps, int n_steps)
{
    IDHMC_DENSE_PROLOGUE;
    for (int64_t c = wave; c < s.C; c += nw) {
        const int64_t off = c * s.L;
        Vec<NCH> q = vload<NCH>(s.q + off, lane);
        Vec<NCH> p = vload<NCH>(s.p + off, lane);
        Vec<NCH> g = vload<NCH>(s.g + off, lane);
        const Vec<NCH> minv = vload<NCH>(s.minv + c * s.minv_stride, lane);
        const double eps = own_eps ? s.eps[c] : eps_arg;
        double lq = 0.0, K = 0.0;
        for (int it = 0; it < n_steps; ++it) leapfrog_step_general<NCH>(mdl, minv, eps, q, p, g, lq, K);
        vstore<NCH>(s.q + off, lane, q);
        vstore<NCH>(s.p + off, lane, p);
        vstore<NCH>(s.g + off, lane, g);
        if (lane == 0) {
            s.lq[c] = lq;
            s.pi[c] = phase_logdensity(lq, K);
        }
    }
}

// find_initial_stepsize (src/stepsize.jl:111-164)
template <int NCH>
IDHMC_DEV double local_ratio_dense(const DenseMvn<NCH> &mdl, const Vec<NCH> &minv, const Vec<NCH> &q, const Vec<NCH> &p,
                                   const Vec<NCH> &g, double eps, double target)
{
    Vec<NCH> q1 = q, p1 = p, g1 = g;
    double lq, K;
    leapfrog_step_general<NCH>(mdl, minv, eps, q1, p1, g1, lq, K);
    return dexp(phase_logdensity(lq, K) - target);
}
template <int NCH>
__global__ __launch_bounds__(kDenseWaves * 64) void k_stepsize_search_dense(DevState s)
{
    IDHMC_DENSE_PROLOGUE;
    for (int64_t c = wave; c < s.C; c += nw) {
        const int64_t off = c * s.L;
        const Vec<NCH> q = vload<NCH>(s.q + off, lane);
        const Vec<NCH> p = vload<NCH>(s.p + off, lane);
        const Vec<NCH> g = vload<NCH>(s.g + off, lane);
        const Vec<NCH> minv = vload<NCH>(s.minv + c * s.minv_stride, lane);
        const double target = phase_logdensity(s.lq[c], kinetic_energy<NCH>(minv, p));
        int rc = 0;
        double e0 = s.ss_eps0, result = s.ss_eps0;
        if (!dfinite(target)) {
            rc = IDHMC_ERR_NONFINITE_START;
        } else {
            double A0 = local_ratio_dense<NCH>(mdl, minv, q, p, g, e0, target);
            if (!(s.ss_a_min <= A0 && A0 <= s.ss_a_max)) {
                const double sg = A0 > s.ss_a_max ? 1.0 : -1.0;
                const double a = A0 > s.ss_a_max ? s.ss_a_max : s.ss_a_min;
                const double Cf = sg < 0.0 ? 1.0 / s.ss_C : s.ss_C;
                double e1 = e0, A1 = A0;
                bool found = false;
                for (int it = 0; it < s.ss_maxiter_crossing; ++it) {
                    const double e = e0 * Cf;
                    const double Ae = local_ratio_dense<NCH>(mdl, minv, q, p, g, e, target);
                    if (sg * (Ae - a) <= 0.0) { e1 = e; A1 = Ae; found = true; break; }
                    e0 = e; A0 = Ae;
                }
                if (!found) {
                    rc = IDHMC_ERR_STEPSIZE_SEARCH;
                } else if (s.ss_a_min <= A1 && A1 <= s.ss_a_max) {
                    result = e1;
                } else {
                    double lo = e0, hi = e1;
                    if (!(e0 < e1)) { lo = e1; hi = e0; }
                    found = false;
                    for (int it = 0; it < s.ss_maxiter_bisect; ++it) {
                        const double em = 0.5 * (lo + hi);
                        const double Am = local_ratio_dense<NCH>(mdl, minv, q, p, g, em, target);
                        if (s.ss_a_min <= Am && Am <= s.ss_a_max) { result = em; found = true; break; }
                        else if (Am < s.ss_a_min) hi = em;
                        else lo = em;
                    }
                    if (!found) rc = IDHMC_ERR_STEPSIZE_SEARCH;
                }
            }
        }
        if (lane == 0) {
            s.eps[c] = result;
            if (rc) s.status[c] = rc;
        }
    }
}

#define IDHMC_DISPATCH_NCH(NCHV, ...)                                  \
    switch (NCHV) {                                                    \
    case 1: { constexpr int NCH = 1; __VA_ARGS__; } break;             \
    case 2: { constexpr int NCH = 2; __VA_ARGS__; } break;             \
    case 4: { constexpr int NCH = 4; __VA_ARGS__; } break;             \
    case 8: { constexpr int NCH = 8; __VA_ARGS__; } break;             \
    default: return hipErrorInvalidValue;                              \
    }

static int dense_grid(int64_t C)
{
    int64_t b = (C + kDenseWaves - 1) / kDenseWaves;
    if (b > 256 * 8) b = 256 * 8;
    return (int)(b < 1 ? 1 : b);
}

hipError_t launch_eval_dense(const DevState &s, hipStream_t st)
{
    IDHMC_DISPATCH_NCH(s.nch, hipLaunchKernelGGL((k_eval_dense<NCH>), dim3(dense_grid(s.C)), dim3(kDenseWaves * 64), 0, st, s, 0));
    return hipGetLastError();
}
hipError_t launch_random_position_dense(const DevState &s, hipStream_t st)
{
    IDHMC_DISPATCH_NCH(s.nch, hipLaunchKernelGGL((k_eval_dense<NCH>), dim3(dense_grid(s.C)), dim3(kDenseWaves * 64), 0, st, s, 1));
    return hipGetLastError();
}
hipError_t launch_leapfrog_dense_mfma(const DevState &s, double eps, int own, hipStream_t st);

hipError_t launch_leapfrog_dense(const DevState &s, double eps, int own, int n_steps, hipStream_t st)
{
    // single step: the matrix-core kernel (L <= 256); IDHMC_DENSE_MFMA=0 selects the per-wave GEMV kernel
    if (n_steps == 1) {
        const char *e = getenv("IDHMC_DENSE_MFMA");
        if (!(e && e[0] == '0')) {
            const hipError_t r = launch_leapfrog_dense_mfma(s, eps, own, st);
            if (r != hipErrorNotSupported) return r;
        }
    }
    IDHMC_DISPATCH_NCH(s.nch, hipLaunchKernelGGL((k_leapfrog_dense<NCH>), dim3(dense_grid(s.C)), dim3(kDenseWaves * 64), 0, st,
                                                 s, eps, own, n_steps));
    return hipGetLastError();
}
hipError_t launch_stepsize_search_dense(const DevState &s, hipStream_t st)
{
    IDHMC_DISPATCH_NCH(s.nch, hipLaunchKernelGGL((k_stepsize_search_dense<NCH>), dim3(dense_grid(s.C)), dim3(kDenseWaves * 64), 0, st, s));
    return hipGetLastError();
}

}  // namespace idhmc
