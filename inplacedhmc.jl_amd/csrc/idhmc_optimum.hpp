// idhmc_optimum.hpp -- the FindLocalOptimum warm-up stage (reference src/warmup.jl:137-187), one chain per
// wavefront, for separable and general densities alike.
//
// Contract of the reference stage: maximise l(q) - 1/2 * magnitude_penalty * sum(q^2) for at most `iterations`
// quasi-Newton iterations; a non-finite result draws a new random position, doubles the penalty and tries
// again, at most 100 times (:162-171), else the stage fails (:172); on success the chain's (q, l(q), grad l(q))
// is the optimum.  The reference's optimiser is QuasiNewtonMethods.proptimize! (:163) -- an external package
// whose source is not in the reference tree -- so the iteration is this engine's own: L-BFGS with kLbfgsM
// pairs and Armijo backtracking, stated once in oracle/idhmc_oracle.c (orc_find_local_optimum) and once here,
// operation for operation, with every reduction in the canonical order: the two agree bit for bit.
//
// Where the state lives: x, grad l, G = grad F, the search direction, the trial point and its gradient in
// VGPRs (6 vectors); the (s, y) history as a ring of kLbfgsM + 1 slots in the wavefront's tree arena (L2) --
// one slot is always free, so a new pair is formed in place and simply not committed when it fails the
// curvature test; rho, alpha and the loop state in SGPR-uniform scalars.
#pragma once
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"

namespace idhmc {

constexpr int kLbfgsM = 5, kLbfgsR = kLbfgsM + 1;
constexpr int kOptimumWaves = 4;

template <int NCH>
IDHMC_DEV double vdot(const Vec<NCH> &a, const Vec<NCH> &b)
{
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        a0 = dfma(a.c[j].x, b.c[j].x, a0);
        a1 = dfma(a.c[j].y, b.c[j].y, a1);
    }
    return wave_sum(a0, a1);
}
// r = fma(a, y, r) elementwise
template <int NCH>
IDHMC_DEV void vaxpy(double a, const Vec<NCH> &y, Vec<NCH> &r)
{
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        r.c[j].x = dfma(a, y.c[j].x, r.c[j].x);
        r.c[j].y = dfma(a, y.c[j].y, r.c[j].y);
    }
}
template <int NCH>
IDHMC_DEV Vec<NCH> vscale(double a, const Vec<NCH> &v)
{
    Vec<NCH> r;
#pragma unroll
    for (int j = 0; j < NCH; ++j) r.c[j] = make_double2(a * v.c[j].x, a * v.c[j].y);
    return r;
}
template <int NCH>
IDHMC_DEV Vec<NCH> vsub(const Vec<NCH> &a, const Vec<NCH> &b)
{
    Vec<NCH> r;
#pragma unroll
    for (int j = 0; j < NCH; ++j) r.c[j] = make_double2(a.c[j].x - b.c[j].x, a.c[j].y - b.c[j].y);
    return r;
}
// small wave-uniform arrays indexed by a run-time slot number: selects instead of scratch memory
IDHMC_DEV double ring_get(const double (&a)[kLbfgsR], int i)
{
    double v = a[0];
#pragma unroll
    for (int u = 1; u < kLbfgsR; ++u) v = (i == u) ? a[u] : v;
    return v;
}
IDHMC_DEV void ring_set(double (&a)[kLbfgsR], int i, double v)
{
#pragma unroll
    for (int u = 0; u < kLbfgsR; ++u) a[u] = (i == u) ? v : a[u];
}
template <int NCH, class Model>
IDHMC_DEV double density_eval(const Model &mdl, const Vec<NCH> &q, Vec<NCH> &g)
{
    double lq;
    if constexpr (Model::kSeparable) lq = eval_density<NCH>(mdl, q, g);
    else lq = mdl.grad(q, g);
    return dfinite(lq) ? lq : -kInf;                    // evaluate_l!, src/kinetic_energy.jl:80-84
}

template <int NCH, class Model>
IDHMC_DEV void local_optimum_body(const DevState &s, const Model &mdl, double penalty, int iterations, double *hist)
{
    constexpr int L = 128 * NCH, M = kLbfgsM, R = kLbfgsR;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double *const S = hist, *const Y = hist + (size_t)R * L;
    for (int64_t c = wave; c < s.C; c += nw) {
        const int64_t off = c * L;
        const RngKey key{s.k0, s.k1, s.first_chain + (uint32_t)c};
        Vec<NCH> x = vload<NCH>(s.q + off, lane);
        Vec<NCH> gl = vload<NCH>(s.g + off, lane);
        double lq = s.lq[c];
        double lam = penalty;
        int rc = IDHMC_ERR_OPTIMIZATION;
        for (uint32_t attempt = 0; attempt < 100u; ++attempt) {                      // :162
            double xx = vdot<NCH>(x, x);
            double F = dfma(0.5 * lam, xx, -lq);
            Vec<NCH> G;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                G.c[j] = make_double2(dfma(lam, x.c[j].x, -gl.c[j].x), dfma(lam, x.c[j].y, -gl.c[j].y));
            double rho[R], alpha[R], gamma = 1.0;
#pragma unroll
            for (int u = 0; u < R; ++u) { rho[u] = 0.0; alpha[u] = 0.0; }
            int k = 0, head = 0;
            for (int it = 0; it < iterations; ++it) {
                const double gg = vdot<NCH>(G, G);
                if (!(gg > 1e-16 * (xx > 1.0 ? xx : 1.0))) break;                     // converged (or not a number)
                Vec<NCH> r = G;
                for (int j = 0; j < k; ++j) {                                         // two-loop recursion, newest first
                    const int i = (head + R - 1 - j) % R;
                    const double al = ring_get(rho, i) * vdot<NCH>(vload<NCH>(S + (size_t)i * L, lane), r);
                    ring_set(alpha, i, al);
                    vaxpy<NCH>(-al, vload<NCH>(Y + (size_t)i * L, lane), r);
                }
                r = vscale<NCH>(k > 0 ? gamma : 1.0 / __builtin_sqrt(gg), r);
                for (int j = k - 1; j >= 0; --j) {                                    // oldest first
                    const int i = (head + R - 1 - j) % R;
                    const double beta = ring_get(rho, i) * vdot<NCH>(vload<NCH>(Y + (size_t)i * L, lane), r);
                    vaxpy<NCH>(ring_get(alpha, i) - beta, vload<NCH>(S + (size_t)i * L, lane), r);
                }
                double gd = -vdot<NCH>(G, r);                                         // direction d = -r
                if (!(gd < 0.0)) {                                                    // not a descent direction: restart
                    k = 0;
                    r = vscale<NCH>(1.0 / __builtin_sqrt(gg), G);
                    gd = -vdot<NCH>(G, r);
                }
                double t = 1.0, lqn = 0.0, xxn = 0.0, Fn = 0.0;
                bool accepted = false;
                Vec<NCH> xn, gln;
                for (int ls = 0; ls < 30; ++ls) {                                     // Armijo backtracking
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
                        xn.c[j] = make_double2(dfma(-t, r.c[j].x, x.c[j].x), dfma(-t, r.c[j].y, x.c[j].y));
                    lqn = density_eval<NCH>(mdl, xn, gln);
                    xxn = vdot<NCH>(xn, xn);
                    Fn = dfma(0.5 * lam, xxn, -lqn);
                    if (dfinite(Fn) && Fn <= dfma(1e-4 * t, gd, F)) { accepted = true; break; }
                    t *= 0.5;
                }
                if (!accepted) break;
                Vec<NCH> Gn;
#pragma unroll
                for (int j = 0; j < NCH; ++j)
                    Gn.c[j] = make_double2(dfma(lam, xn.c[j].x, -gln.c[j].x), dfma(lam, xn.c[j].y, -gln.c[j].y));
                const Vec<NCH> sv = vsub<NCH>(xn, x), yv = vsub<NCH>(Gn, G);
                vstore<NCH>(S + (size_t)head * L, lane, sv);                          // the ring's free slot
                vstore<NCH>(Y + (size_t)head * L, lane, yv);
                const double sy = vdot<NCH>(sv, yv), yy = vdot<NCH>(yv, yv);
                if (sy > 1e-10 * yy) {                                                // curvature condition: commit the pair
                    ring_set(rho, head, 1.0 / sy);
                    gamma = sy / yy;
                    head = (head + 1) % R;
                    if (k < M) ++k;
                }
                x = xn; gl = gln; G = Gn;
                lq = lqn; F = Fn; xx = xxn;
            }
            if (dfinite(lq)) { rc = 0; break; }                                       // :168
            // random_position! with this attempt's draws, evaluate, double the penalty (:169-171)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int pair = j * 64 + lane;
                const u32x4 u = rng_draw(key, attempt + 1u, kStreamInitQ, (uint32_t)pair);
                const double u0 = u01(u.x, u.y), u1 = u01(u.z, u.w);
                x.c[j].x = (2 * pair < s.D) ? dfma(4.0, u0, -2.0) : 0.0;
                x.c[j].y = (2 * pair + 1 < s.D) ? dfma(4.0, u1, -2.0) : 0.0;
            }
            lq = density_eval<NCH>(mdl, x, gl);
            lam += lam;
        }
        vstore<NCH>(s.q + off, lane, x);
        vstore<NCH>(s.g + off, lane, gl);
        if (lane == 0) {
            s.lq[c] = lq;
            if (rc) s.status[c] = rc;                                                 // "Optimization failed to converge", :172
        }
    }
}

// separable densities: parameters in registers
template <int NCH, class Model>
__global__ __launch_bounds__(kOptimumWaves * 64) void k_local_optimum(DevState s, double penalty, int iterations)
{
    Model mdl;
    mdl.load(s.mu, s.tau, threadIdx.x & 63);
    double *hist = s.arena + ((int64_t)blockIdx.x * kOptimumWaves + (threadIdx.x >> 6)) * s.arena_stride;
    local_optimum_body<NCH, Model>(s, mdl, penalty, iterations, hist);
}
// general densities: one LDS vector per wavefront
template <int NCH, class Model>
__global__ __launch_bounds__(kOptimumWaves * 64) void k_local_optimum_general(DevState s, double penalty, int iterations)
{
    __shared__ __attribute__((aligned(16))) double dshare[kOptimumWaves][128 * NCH];
    Model mdl;
    mdl.init(s, dshare[threadIdx.x >> 6], threadIdx.x & 63);
    double *hist = s.arena + ((int64_t)blockIdx.x * kOptimumWaves + (threadIdx.x >> 6)) * s.arena_stride;
    local_optimum_body<NCH, Model>(s, mdl, penalty, iterations, hist);
}

// every launched wavefront owns one arena slot (>= 2 kLbfgsR vectors: the arena holds at least 12)
inline int optimum_grid(const DevState &s)
{
    int64_t b = (s.C + kOptimumWaves - 1) / kOptimumWaves;
    const int64_t have = s.nslots / kOptimumWaves;
    if (b > have) b = have;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace idhmc
