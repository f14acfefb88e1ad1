// idhmc_general.hpp -- streaming kernels for a GENERAL (non-separable) density: evaluation, fused leapfrog,
// initial-stepsize search, one chain per wavefront.  A general density is a type with
//     template <class State> __device__ void init(const State &s, double *lds_vec, int lane);
//     __device__ double grad(const Vec<NCH> &q, Vec<NCH> &g) const;     // returns l(q), fills grad l(q)
// -- the device form of the reference's logdensity_and_gradient!(grad, model, q, sptr) (src/kinetic_energy.jl:73).
// Instantiated ahead of time for the dense multivariate normal (idhmc_dense.hip) and at run time through
// hipRTC for a user-supplied density (idhmc_jit.hip).  Bodies are __device__ templates; the __global__
// wrappers own the per-wavefront LDS vector.
#pragma once
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"

namespace idhmc {

constexpr int kGeneralWaves = 4;   // wavefronts per workgroup of the general streaming kernels

// evaluate_l! (src/kinetic_energy.jl:72-85); random_q = 1: q ~ U[-2,2) first (random_position!, src/warmup.jl:73)
template <int NCH, class Model>
IDHMC_DEV void eval_general_body(const DevState &s, int random_q, double *lds_vec)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Model mdl;
    mdl.init(s, lds_vec, lane);
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
template <int NCH, class Model>
IDHMC_DEV void leapfrog_general_body(const DevState &s, double eps_arg, int own_eps, int n_steps, double *lds_vec)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Model mdl;
    mdl.init(s, lds_vec, lane);
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

// A(eps) of find_initial_stepsize (src/stepsize.jl:150-154)
template <int NCH, class Model>
IDHMC_DEV double local_ratio_general(const Model &mdl, const Vec<NCH> &minv, const Vec<NCH> &q, const Vec<NCH> &p,
                                     const Vec<NCH> &g, double eps, double target)
{
    Vec<NCH> q1 = q, p1 = p, g1 = g;
    double lq, K;
    leapfrog_step_general<NCH>(mdl, minv, eps, q1, p1, g1, lq, K);
    return dexp(phase_logdensity(lq, K) - target);
}

// find_initial_stepsize (src/stepsize.jl:111-164)
template <int NCH, class Model>
IDHMC_DEV void stepsize_general_body(const DevState &s, double *lds_vec)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Model mdl;
    mdl.init(s, lds_vec, lane);
    for (int64_t c = wave; c < s.C; c += nw) {
        const int64_t off = c * s.L;
        const Vec<NCH> q = vload<NCH>(s.q + off, lane);
        const Vec<NCH> p = vload<NCH>(s.p + off, lane);
        const Vec<NCH> g = vload<NCH>(s.g + off, lane);
        const Vec<NCH> minv = vload<NCH>(s.minv + c * s.minv_stride, lane);
        const double target = phase_logdensity(s.lq[c], kinetic_energy<NCH>(minv, p));     // :151
        int rc = 0;
        double e0 = s.ss_eps0, result = s.ss_eps0;
        if (!dfinite(target)) {
            rc = IDHMC_ERR_NONFINITE_START;                                                // :152-153
        } else {
            double A0 = local_ratio_general<NCH>(mdl, minv, q, p, g, e0, target);          // :113
            if (!(s.ss_a_min <= A0 && A0 <= s.ss_a_max)) {                                 // :114
                const double sg = A0 > s.ss_a_max ? 1.0 : -1.0;                            // find_crossing_stepsize :51-72
                const double a = A0 > s.ss_a_max ? s.ss_a_max : s.ss_a_min;
                const double Cf = sg < 0.0 ? 1.0 / s.ss_C : s.ss_C;
                double e1 = e0, A1 = A0;
                bool found = false;
                for (int it = 0; it < s.ss_maxiter_crossing; ++it) {
                    const double e = e0 * Cf;
                    const double Ae = local_ratio_general<NCH>(mdl, minv, q, p, g, e, target);
                    if (sg * (Ae - a) <= 0.0) { e1 = e; A1 = Ae; found = true; break; }
                    e0 = e; A0 = Ae;
                }
                if (!found) {
                    rc = IDHMC_ERR_STEPSIZE_SEARCH;                                        // :71
                } else if (s.ss_a_min <= A1 && A1 <= s.ss_a_max) {
                    result = e1;                                                           // :118
                } else {
                    double lo = e0, hi = e1;                                               // :120-124
                    if (!(e0 < e1)) { lo = e1; hi = e0; }
                    found = false;
                    for (int it = 0; it < s.ss_maxiter_bisect; ++it) {                     // bisect_stepsize :83-102
                        const double em = 0.5 * (lo + hi);
                        const double Am = local_ratio_general<NCH>(mdl, minv, q, p, g, em, target);
                        if (s.ss_a_min <= Am && Am <= s.ss_a_max) { result = em; found = true; break; }
                        else if (Am < s.ss_a_min) hi = em;
                        else lo = em;
                    }
                    if (!found) rc = IDHMC_ERR_STEPSIZE_SEARCH;                            // :101
                }
            }
        }
        if (lane == 0) {
            s.eps[c] = result;
            if (rc) s.status[c] = rc;
        }
    }
}

// __global__ wrappers (each owns one LDS vector per wavefront)
template <int NCH, class Model>
__global__ __launch_bounds__(kGeneralWaves * 64) void k_eval_general(DevState s, int random_q)
{
    __shared__ __attribute__((aligned(16))) double dshare[kGeneralWaves][128 * NCH];
    eval_general_body<NCH, Model>(s, random_q, dshare[threadIdx.x >> 6]);
}
template <int NCH, class Model>
__global__ __launch_bounds__(kGeneralWaves * 64) void k_leapfrog_general(DevState s, double eps, int own_eps, int n_steps)
{
    __shared__ __attribute__((aligned(16))) double dshare[kGeneralWaves][128 * NCH];
    leapfrog_general_body<NCH, Model>(s, eps, own_eps, n_steps, dshare[threadIdx.x >> 6]);
}
template <int NCH, class Model>
__global__ __launch_bounds__(kGeneralWaves * 64) void k_stepsize_general(DevState s)
{
    __shared__ __attribute__((aligned(16))) double dshare[kGeneralWaves][128 * NCH];
    stepsize_general_body<NCH, Model>(s, dshare[threadIdx.x >> 6]);
}

inline int general_grid(int64_t C)
{
    int64_t b = (C + kGeneralWaves - 1) / kGeneralWaves;
    if (b > 256 * 8) b = 256 * 8;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace idhmc
