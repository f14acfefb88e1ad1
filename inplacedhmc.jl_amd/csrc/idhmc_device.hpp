// idhmc_device.hpp -- per-wavefront building blocks: one chain per 64-lane wavefront.
//
// Data layout in HBM: every per-chain vector (q, p, grad l, M^-1, W, tree arena vectors) is a
// contiguous run of L = 128*NCH doubles, chain-major ([chain][L]); fields are separate arrays
// (SoA).  Lane l of the wavefront owns elements {128 j + 2l, 128 j + 2l + 1 : j < NCH}, i.e. one
// 16-byte global_load_dwordx4 per 128-element chunk, 1 KiB contiguous per wave instruction.
// Pads (elements D..L-1) are zero in q, p, grad, mu, tau and one in M^-1, W and stay so.
#pragma once
#include "idhmc_math.hpp"

namespace idhmc {

// nothing is scheduled across this point (register-pressure control in the NUTS kernel)
IDHMC_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

template <int NCH>
struct Vec {
    double2 c[NCH];
    IDHMC_DEV double2 get(int j) const { return c[j]; }
};

// a read-only vector living in LDS (shared density parameters, M^-1): lane-offset pointer, one
// ds_read_b128 per chunk, conflict-free (16 consecutive bytes per lane)
struct LdsVec {
    const double2 *p;
    IDHMC_DEV double2 get(int j) const { return p[j * 64]; }
};

template <int NCH>
IDHMC_DEV Vec<NCH> vload(const double *base, int lane)
{
    Vec<NCH> v;
    const double2 *b = reinterpret_cast<const double2 *>(base) + lane;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v.c[j] = b[j * 64];
    return v;
}
template <int NCH>
IDHMC_DEV void vstore(double *base, int lane, const Vec<NCH> &v)
{
    double2 *b = reinterpret_cast<double2 *>(base) + lane;
#pragma unroll
    for (int j = 0; j < NCH; ++j) b[j * 64] = v.c[j];
}
// The same accesses through a raw buffer resource: the wave-uniform base lives in SGPRs, every chunk of every
// vector shares ONE per-lane offset register (lane * 16; the chunk's j * 1024 goes into the instruction's
// immediate field, with one more register for chunks past 4 KiB).  With flat 64-bit addressing the compiler
// keeps a VGPR pair per (vector, chunk) address, hoists them all out of the NUTS kernel's loops and spills them.
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
IDHMC_DEV __amdgpu_buffer_rsrc_t buf_rsrc(const void *p)
{
    // the base is wave-uniform; readfirstlane hands the compiler the proof (no waterfall loop)
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(a >> 32));
    void *u = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(u, 0, -1, 0x00020000);   // raw buffer, 4 GiB window, DATA_FORMAT_32 (gfx9 family)
}
// AUX: the instruction's cache-policy bits; kAuxNt (nt) marks data that is touched once per launch (the chain state,
// the regeneration checkpoints) so that it does not displace the tree arena from L2 / Infinity Cache
constexpr int kAuxNt = 2;
template <int NCH, int AUX = 0>
IDHMC_DEV Vec<NCH> bload(const double *base, int lane)
{
    Vec<NCH> v;
    const __amdgpu_buffer_rsrc_t r = buf_rsrc(base);
    const int vo = lane * 16;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const v4u32 w = __builtin_amdgcn_raw_buffer_load_b128(r, vo + j * 1024, 0, AUX);
        v.c[j] = __builtin_bit_cast(double2, w);
    }
    return v;
}
template <int NCH, int AUX = 0>
IDHMC_DEV void bstore(double *base, int lane, const Vec<NCH> &v)
{
    const __amdgpu_buffer_rsrc_t r = buf_rsrc(base);
    const int vo = lane * 16;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v.c[j]), r, vo + j * 1024, 0, AUX);
}
template <int NCH>
IDHMC_DEV Vec<NCH> vfill(double x)
{
    Vec<NCH> v;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v.c[j] = make_double2(x, x);
    return v;
}

// ---- user densities (reference contract: logdensity_and_gradient!, src/kinetic_energy.jl:73) -----
// A separable density supplies (mu, tau) per element; grad = -(tau*(q-mu)), l = -1/2 sum tau (q-mu)^2.
template <int NCH>
struct IsoGaussian {
    static constexpr bool kHasParams = false;
    static constexpr bool kSeparable = true;
    static constexpr bool kCooperative = false;
    IDHMC_DEV void load(const double *, const double *, int) {}
    IDHMC_DEV double2 mu(int) const { return make_double2(0.0, 0.0); }
    IDHMC_DEV double2 tau(int) const { return make_double2(1.0, 1.0); }
};
template <int NCH>
struct DiagGaussian {
    static constexpr bool kHasParams = true;
    static constexpr bool kSeparable = true;
    static constexpr bool kCooperative = false;
    Vec<NCH> m, t;
    IDHMC_DEV void load(const double *mu_, const double *tau_, int lane)
    {
        m = vload<NCH>(mu_, lane);
        t = vload<NCH>(tau_, lane);
    }
    IDHMC_DEV double2 mu(int j) const { return m.c[j]; }
    IDHMC_DEV double2 tau(int j) const { return t.c[j]; }
};

// the same density with mu, tau staged in LDS once per workgroup (NUTS kernel)
template <int NCH>
struct DiagGaussianLds {
    static constexpr bool kHasParams = true;
    static constexpr bool kSeparable = true;
    static constexpr bool kCooperative = false;
    const double2 *m, *t;   // lane-offset LDS pointers
    IDHMC_DEV double2 mu(int j) const { return m[j * 64]; }
    IDHMC_DEV double2 tau(int j) const { return t[j * 64]; }
};

// Dense multivariate normal, l(q) = -1/2 (q-mu)' P (q-mu), P = Sigma^-1 symmetric (BASELINE.json
// configs[3]).  General (non-separable) density form: grad() needs the whole vector.
//   d = q - mu;  t_r = fma chain over c = 0, 1, ... of P[c][r] * d_c  (ascending c; P symmetric, so row c
//   of P is read as one coalesced 16-byte-per-lane stream);  grad = -t;  l = -1/2 canonical_sum(t .* d).
// One wavefront per chain: d is staged in this wave's LDS vector and broadcast one element per step.
template <int NCH>
struct DenseMvn {
    static constexpr bool kHasParams = true;
    static constexpr bool kSeparable = false;
    static constexpr bool kCooperative = false;
    const double *prec;      // [L][L] row-major, device
    const double2 *mu2;      // lane-offset, device
    double *dbuf;            // this wavefront's LDS staging vector, L doubles
    int D, lane;
    // every general density is set up the same way: the kernel state and one LDS vector of this wavefront
    template <class State>
    IDHMC_DEV void init(const State &s, double *lds_vec, int lane_)
    {
        prec = s.prec;
        mu2 = reinterpret_cast<const double2 *>(s.mu) + lane_;
        dbuf = lds_vec;
        D = s.D;
        lane = lane_;
    }
    IDHMC_DEV double grad(const Vec<NCH> &q, Vec<NCH> &g) const
    {
        constexpr int L = 128 * NCH;
        Vec<NCH> d, t;
        double2 *db2 = reinterpret_cast<double2 *>(dbuf) + lane;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const double2 m = mu2[j * 64];
            d.c[j] = make_double2(q.c[j].x - m.x, q.c[j].y - m.y);
            db2[j * 64] = d.c[j];
            t.c[j] = make_double2(0.0, 0.0);
        }
        const double2 *row = reinterpret_cast<const double2 *>(prec) + lane;
#pragma unroll 4
        for (int c = 0; c < D; ++c) {
            const double dc = dbuf[c];                       // LDS broadcast
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const double2 pr = row[(size_t)c * (L / 2) + j * 64];
                t.c[j].x = dfma(pr.x, dc, t.c[j].x);
                t.c[j].y = dfma(pr.y, dc, t.c[j].y);
            }
        }
        double l0 = 0.0, l1 = 0.0;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            g.c[j] = make_double2(-t.c[j].x, -t.c[j].y);
            l0 = dfma(t.c[j].x, d.c[j].x, l0);
            l1 = dfma(t.c[j].y, d.c[j].y, l1);
        }
        const double lq = -0.5 * wave_sum(l0, l1);
        return dfinite(lq) ? lq : -kInf;
    }
};

// The same density for the NUTS kernel at L <= 256, evaluated COOPERATIVELY by the 16 wavefronts of a
// workgroup on the fp64 matrix cores.  Each wavefront still runs its own chain's tree; a gradient request is
// one *service round* of the workgroup:
//   (1) the requester writes d = q - mu as its row of a [16 chains][L] LDS tile;        -- barrier A --
//   (2) the first L/32 wavefronts multiply the whole tile by their own 32 columns of P (T goes to a second tile: no barrier B since round 3):
//       T[16][32w..32w+31] = Dm[16][L] * P[L][32w..], v_mfma_f64_16x16x4_f64, k ascending (the
//       engine's summation order, bit-identical to the per-wave GEMV above), P read from L2 in
//       the B-operand layout and prefetched kPrefetch k-blocks ahead;
//   (3) they write their columns of T into the second tile;                                 -- barrier C --
//   (4) the requester reads its row back in its own lane layout: grad = -t, l = -1/2 sum t.d.
// P streams through L1 once per 16 gradients instead of once per gradient.  Wavefronts whose chain has
// finished (or that have none) keep serving rounds -- serve() -- until no chain of the group is alive; the
// alive count changes only between a wavefront's rounds and is read between barriers A and B, so every
// wavefront sees the same value and all leave together.  Rows are private to their wavefront outside (2)-(3),
// so a fast wavefront may write its next d while a slow one still reads its t.
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef IDHMC_COOP_PD
#define IDHMC_COOP_PD 4
#endif
template <int NCH>
struct DenseMvnCoop {
    static constexpr bool kHasParams = true;
    static constexpr bool kSeparable = false;
    static constexpr bool kCooperative = true;
    static constexpr int kWaves = 16, L = 128 * NCH, DS = L + 2, KB = L / 4, kPairs = L / 32, kPrefetch = IDHMC_COOP_PD;
    static constexpr int kTileDoubles = 16 * DS;
    typedef v2d Prefetch[kPrefetch];
    const double *prec;      // [L][L] row-major, device
    const double2 *mu2;      // lane-offset, device
    double *tile;            // [16][DS] in LDS, shared by the workgroup
    int *alive;              // LDS: chains of the current group that may still request a gradient
    int lane, wv;
    template <class State>
    IDHMC_DEV void init(const State &s, double *tile_, int *alive_, int lane_, int wv_)
    {
        prec = s.prec;
        mu2 = reinterpret_cast<const double2 *>(s.mu) + lane_;
        tile = tile_;
        alive = alive_;
        lane = lane_;
        wv = wv_;
    }
    // The first kPairs wavefronts (two per SIMD at L = 256) each own a 32-column block of P: lane (kk, jj) reads
    // columns 2jj, 2jj+1 of row 4kb + kk with one 16-byte load, so the block's two MFMA tiles are its even and its
    // odd columns (L2 -> CU bandwidth bounds this phase: 16-byte requests move it ~30 % faster than 8-byte ones).
    // The first k-blocks do not depend on the tile: requested before barrier A, their L2 latency runs under the
    // wait for the slowest wavefront.
#ifndef IDHMC_COOP_SPLIT16
#define IDHMC_COOP_SPLIT16 0
#endif
    // IDHMC_COOP_SPLIT16 (L = 256 only): all 16 wavefronts multiply, 16 columns (one MFMA tile) each, instead of 8 wavefronts with 32
    // columns while the other 8 wait at the barrier -- same k order per output element, so the same bits
    static constexpr bool kSplit16 = IDHMC_COOP_SPLIT16 != 0 && L == 256;
    IDHMC_DEV void prefetch(Prefetch &bq) const
    {
        static_assert(kPairs <= kWaves, "one 32-column block per wavefront: L <= 512");
        if constexpr (kSplit16) {
            const __amdgpu_buffer_rsrc_t rP = buf_rsrc(prec);
            const int vo = ((lane >> 4) * L + 16 * wv + (lane & 15)) * 8;
#pragma unroll
            for (int u = 0; u < kPrefetch; ++u)
                bq[u].x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rP, vo, 4 * u * L * 8, 0));
            return;
        }
        static_assert((KB & (KB - 1)) == 0 && KB % kPrefetch == 0, "k-block count: power of two, multiple of the prefetch depth");
        if (wv < kPairs) {
            const __amdgpu_buffer_rsrc_t rP = buf_rsrc(prec);
            const int vo = ((lane >> 4) * L + 32 * wv + 2 * (lane & 15)) * 8;
#pragma unroll
            for (int u = 0; u < kPrefetch; ++u)
                bq[u] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rP, vo, 4 * u * L * 8, 0));
        }
    }
    // steps (2)-(3) of a round; entered after barrier A by all 16 wavefronts
    IDHMC_DEV void multiply(Prefetch &bq) const
    {
        // lane-derived addresses are recomputed here (the empty asm hides the lane id from loop-invariant code
        // motion): hoisted out of the transition they get spilled, and a spill reload pending at the loop head
        // makes the compiler wait for ALL loads at every trip of the k loop
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int kk = ln >> 4, jj = ln & 15;
        v4d acc0 = v4d{0.0, 0.0, 0.0, 0.0}, acc1 = v4d{0.0, 0.0, 0.0, 0.0};
        if constexpr (kSplit16) {
            const __amdgpu_buffer_rsrc_t rP = buf_rsrc(prec);
            const int vo = (kk * L + 16 * wv + jj) * 8;
            const double *ap = tile + jj * DS + kk;
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll 1
            for (int kb0 = 0; kb0 < KB; kb0 += kPrefetch) {
#pragma unroll
                for (int u = 0; u < kPrefetch; ++u) {
                    const int kb = kb0 + u;
                    const double a = ap[4 * kb];
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[u].x, acc0, 0, 0, 0);
                    bq[u].x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rP, vo, 4 * ((kb + kPrefetch) & (KB - 1)) * L * 8, 0));
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) tile[kTileDoubles + (kk + 4 * reg) * DS + 16 * wv + jj] = acc0[reg];
            __syncthreads();                               // barrier C
            return;
        }
        if (wv < kPairs) {
            const __amdgpu_buffer_rsrc_t rP = buf_rsrc(prec);
            const int vo = (kk * L + 32 * wv + 2 * jj) * 8;
            const double *ap = tile + jj * DS + kk;
            // drain the vector-memory counter once here: with a clean slate the loop waits for exactly the block it
            // needs (vmcnt = kPrefetch - 1) instead of for everything outstanding
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll 1
            for (int kb0 = 0; kb0 < KB; kb0 += kPrefetch) {
#pragma unroll
                for (int u = 0; u < kPrefetch; ++u) {
                    const int kb = kb0 + u;
                    const double a = ap[4 * kb];
#ifdef IDHMC_CX2      // (cost attribution, results wrong on purpose) no matrix-core work
                    acc0[0] = dfma(a, bq[u].x, acc0[0]);
                    acc1[0] = dfma(a, bq[u].y, acc1[0]);
#else
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[u].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[u].y, acc1, 0, 0, 0);
#endif
                    // unconditional (the last trips wrap around and are discarded): a branch here makes the
                    // compiler drain all outstanding loads at every trip
#ifndef IDHMC_CX1     // (cost attribution) the matrix is fetched once per round, not per k-block
                    bq[u] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(
                        rP, vo, 4 * ((kb + kPrefetch) & (KB - 1)) * L * 8, 0));
#endif
                }
            }
        }
        // T is written to its own tile (round 3): the d tile may still be read by slower wavefronts, so writing into it needed a
        // barrier of its own between the k loop and the write-back (three barriers per round, now two)
        if (wv < kPairs) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                v2d t;
                t.x = acc0[reg];
                t.y = acc1[reg];
                *reinterpret_cast<v2d *>(tile + kTileDoubles + (kk + 4 * reg) * DS + 32 * wv + 2 * jj) = t;
            }
        }
        __syncthreads();                                   // barrier C: T is complete
    }
    IDHMC_DEV double grad(const Vec<NCH> &q, Vec<NCH> &g) const
    {
        double l0, l1;
        grad_partial(q, g, l0, l1);
        const double lq = -0.5 * wave_sum(l0, l1);
        return dfinite(lq) ? lq : -kInf;
    }
    // the same without the final reduction: l = -1/2 wave_sum(l0, l1) -- the leapfrog reduces it together with the kinetic energy
    IDHMC_DEV void grad_partial(const Vec<NCH> &q, Vec<NCH> &g, double &l0, double &l1) const
    {
        Prefetch bq;
        prefetch(bq);
        Vec<NCH> d;
        double2 *row = reinterpret_cast<double2 *>(tile + wv * DS) + lane;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const double2 m = mu2[j * 64];
            d.c[j] = make_double2(q.c[j].x - m.x, q.c[j].y - m.y);
            row[j * 64] = d.c[j];
        }
        __syncthreads();                                   // barrier A
        multiply(bq);
        l0 = 0.0; l1 = 0.0;
        const double2 *trow = reinterpret_cast<const double2 *>(tile + kTileDoubles + wv * DS) + lane;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const double2 t = trow[j * 64];
            g.c[j] = make_double2(-t.x, -t.y);
            l0 = dfma(t.x, d.c[j].x, l0);
            l1 = dfma(t.y, d.c[j].y, l1);
        }
    }
    // this wavefront's chain makes no further request
    IDHMC_DEV void retire() const
    {
        if (lane == 0) atomicSub(alive, 1);
    }
    // one round served without a request of its own (the wavefront has a chain coming, so the group is alive)
    IDHMC_DEV void serve_round() const
    {
        Prefetch bq;
        prefetch(bq);
        __syncthreads();                                   // barrier A
        multiply(bq);
    }
    // serve the other chains' rounds until the whole group has retired
    IDHMC_DEV void serve() const
    {
        for (;;) {
            Prefetch bq;
            prefetch(bq);
            __syncthreads();                               // barrier A
            if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int *>(alive)) == 0) break;
            multiply(bq);
        }
    }
};

// A user-supplied density (IDHMC_MODEL_CUSTOM, include/idhmc.h): the context its
// logdensity_and_gradient<NCH>(q, grad, ctx) receives, and the adapter that makes it a general density.
struct UserCtx {
    const double *params;   // the user's parameter blob (device memory)
    long long nparams;
    double *lds;            // one L-double scratch vector in LDS, private to this wavefront
    int D, lane;
};
#ifdef IDHMC_JIT_USER_DENSITY
template <int NCH>
__device__ double logdensity_and_gradient(const Vec<NCH> &q, Vec<NCH> &grad, const UserCtx &ctx);   // user-defined
template <int NCH>
struct JitModel {
    static constexpr bool kHasParams = true;
    static constexpr bool kSeparable = false;
    static constexpr bool kCooperative = false;
    UserCtx ctx;
    template <class State>
    IDHMC_DEV void init(const State &s, double *lds_vec, int lane)
    {
        ctx.params = s.user_params;
        ctx.nparams = s.user_nparams;
        ctx.lds = lds_vec;
        ctx.D = s.D;
        ctx.lane = lane;
    }
    IDHMC_DEV double grad(const Vec<NCH> &q, Vec<NCH> &g) const
    {
        const double lq = logdensity_and_gradient<NCH>(q, g, ctx);
        return dfinite(lq) ? lq : -kInf;                      // evaluate_l!, src/kinetic_energy.jl:80-84
    }
};
#endif

// leapfrog for a general density (src/kinetic_energy.jl:144-161 as written there): loop A, evaluate_l!,
// loop B, then K(p')
template <int NCH, class Model, class Metric>
IDHMC_DEV void leapfrog_step_general(const Model &mdl, const Metric &minv, double eps, Vec<NCH> &q,
                                     Vec<NCH> &p, Vec<NCH> &g, double &lq, double &K)
{
    const double eh = 0.5 * eps;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mv = minv.get(j);
        p.c[j].x = dfma(eh, g.c[j].x, p.c[j].x);
        p.c[j].y = dfma(eh, g.c[j].y, p.c[j].y);
        q.c[j].x = dfma(eps * mv.x, p.c[j].x, q.c[j].x);
        q.c[j].y = dfma(eps * mv.y, p.c[j].y, q.c[j].y);
    }
    double l0 = 0.0, l1 = 0.0;
    if constexpr (Model::kCooperative) mdl.grad_partial(q, g, l0, l1);
    else lq = mdl.grad(q, g);
    double k0 = 0.0, k1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mv = minv.get(j);
        p.c[j].x = dfma(eh, g.c[j].x, p.c[j].x);
        p.c[j].y = dfma(eh, g.c[j].y, p.c[j].y);
        k0 = dfma(p.c[j].x * mv.x, p.c[j].x, k0);
        k1 = dfma(p.c[j].y * mv.y, p.c[j].y, k1);
    }
    if constexpr (Model::kCooperative) {
        // one pass for both reductions (each the canonical tree: the same bits as two wave_sums)
        double sl, sk;
        wave_sum2(l0, l1, k0, k1, sl, sk);
        lq = -0.5 * sl;
        lq = dfinite(lq) ? lq : -kInf;
        K = 0.5 * sk;
    } else {
        K = 0.5 * wave_sum(k0, k1);
    }
}

// l(q), grad l(q) for a separable density; evaluate_l! semantics (src/kinetic_energy.jl:72-85):
// a non-finite l(q) becomes -Inf.
template <int NCH, class Model>
IDHMC_DEV double eval_density(const Model &mdl, const Vec<NCH> &q, Vec<NCH> &g)
{
    double l0 = 0.0, l1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mu = mdl.mu(j), tau = mdl.tau(j);
        const double dx = q.c[j].x - mu.x, dy = q.c[j].y - mu.y;
        const double tx = tau.x * dx, ty = tau.y * dy;
        g.c[j].x = -tx;
        g.c[j].y = -ty;
        l0 = dfma(tx, dx, l0);
        l1 = dfma(ty, dy, l1);
    }
    const double lq = -0.5 * wave_sum(l0, l1);
    return dfinite(lq) ? lq : -kInf;
}

// kinetic_energy (src/kinetic_energy.jl:14-24): K = 1/2 sum p * M^-1 * p
template <int NCH, class Metric>
IDHMC_DEV double kinetic_energy(const Metric &minv, const Vec<NCH> &p)
{
    double k0 = 0.0, k1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mv = minv.get(j);
        k0 = dfma(p.c[j].x * mv.x, p.c[j].x, k0);
        k1 = dfma(p.c[j].y * mv.y, p.c[j].y, k1);
    }
    return 0.5 * wave_sum(k0, k1);
}

// logdensity(H, z) (src/kinetic_energy.jl:107-112)
IDHMC_DEV double phase_logdensity(double lq, double K)
{
    if (!dfinite(lq)) return -kInf;
    return lq - (dfinite(K) ? K : kInf);
}

// One leapfrog step in registers: loop A, gradient, loop B (src/kinetic_energy.jl:144-161) fused with
// the two reductions the caller needs next (l(q'), K(p')).  Six separate memory passes in the
// reference; zero here.
template <int NCH, class Model, class Metric>
IDHMC_DEV void leapfrog_step(const Model &mdl, const Metric &minv, double eps, Vec<NCH> &q,
                             Vec<NCH> &p, Vec<NCH> &g, double &lq, double &K)
{
    const double eh = 0.5 * eps;
    double l0 = 0.0, l1 = 0.0, k0 = 0.0, k1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mu = mdl.mu(j), tau = mdl.tau(j), mv = minv.get(j);
        // loop A: p_m = p + eps/2 grad;  q' = q + eps M^-1 p_m
        const double pmx = dfma(eh, g.c[j].x, p.c[j].x), pmy = dfma(eh, g.c[j].y, p.c[j].y);
        const double qx = dfma(eps * mv.x, pmx, q.c[j].x);
        const double qy = dfma(eps * mv.y, pmy, q.c[j].y);
        // gradient at q'
        const double dx = qx - mu.x, dy = qy - mu.y;
        const double tx = tau.x * dx, ty = tau.y * dy;
        const double gx = -tx, gy = -ty;
        l0 = dfma(tx, dx, l0);
        l1 = dfma(ty, dy, l1);
        // loop B: p' = p_m + eps/2 grad'
        const double px = dfma(eh, gx, pmx), py = dfma(eh, gy, pmy);
        k0 = dfma(px * mv.x, px, k0);
        k1 = dfma(py * mv.y, py, k1);
        q.c[j] = make_double2(qx, qy);
        p.c[j] = make_double2(px, py);
        g.c[j] = make_double2(gx, gy);
    }
    double sl, sk;
    wave_sum2(l0, l1, k0, k1, sl, sk);
    lq = -0.5 * sl;
    lq = dfinite(lq) ? lq : -kInf;
    K = 0.5 * sk;
}

// The same step for densities whose gradient is cheaper to recompute than to carry (the separable
// Gaussians: 2 flops per element): grad l(q) is re-derived from q at the start of the step, which gives
// the very bits the previous step computed, and is not returned.  Saves a third of the phase point's
// registers in the NUTS kernel.
// FENCE: software pipeline over the 128-element chunks for parameters that live in LDS (see below); with the
// parameters in registers (register-rich NUTS form) the scheduler is left free to interleave the 2 NCH element chains.
template <int NCH, bool FENCE = true, class Model, class Metric>
IDHMC_DEV void leapfrog_step_regrad(const Model &mdl, const Metric &minv, double eps, Vec<NCH> &q,
                                    Vec<NCH> &p, double &lq, double &K)
{
    const double eh = 0.5 * eps;
    double l0 = 0.0, l1 = 0.0, k0 = 0.0, k1 = 0.0;
    // software pipeline over the 128-element chunks: the (LDS) parameter reads of chunk j+1 are issued
    // before the arithmetic of chunk j; the scheduling fences keep the compiler from hoisting all
    // 3*NCH reads to the top (96 VGPRs at D = 1024), which is what spills this kernel otherwise.
    if (FENCE) sched_fence();
    double2 mu_n = mdl.mu(0), tau_n = mdl.tau(0), mv_n = minv.get(0);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mu = mu_n, tau = tau_n, mv = mv_n;
        if (j + 1 < NCH) { mu_n = mdl.mu(j + 1); tau_n = mdl.tau(j + 1); mv_n = minv.get(j + 1); }
        const double g0x = -(tau.x * (q.c[j].x - mu.x)), g0y = -(tau.y * (q.c[j].y - mu.y));
        const double pmx = dfma(eh, g0x, p.c[j].x), pmy = dfma(eh, g0y, p.c[j].y);
        const double qx = dfma(eps * mv.x, pmx, q.c[j].x);
        const double qy = dfma(eps * mv.y, pmy, q.c[j].y);
        const double dx = qx - mu.x, dy = qy - mu.y;
        const double tx = tau.x * dx, ty = tau.y * dy;
        l0 = dfma(tx, dx, l0);
        l1 = dfma(ty, dy, l1);
        const double px = dfma(eh, -tx, pmx), py = dfma(eh, -ty, pmy);
        k0 = dfma(px * mv.x, px, k0);
        k1 = dfma(py * mv.y, py, k1);
        q.c[j] = make_double2(qx, qy);
        p.c[j] = make_double2(px, py);
        if (FENCE) sched_fence();
    }
    double sl, sk;
    wave_sum2(l0, l1, k0, k1, sl, sk);
    lq = -0.5 * sl;
    lq = dfinite(lq) ? lq : -kInf;
    K = 0.5 * sk;
    if (FENCE) sched_fence();
}

// rand_p! (src/kinetic_energy.jl:63): p = W .* randn, pads stay zero
template <int NCH>
IDHMC_DEV Vec<NCH> rand_momentum(const RngKey &key, uint32_t iter, const Vec<NCH> &w, int lane, int D)
{
    Vec<NCH> p;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int pair = j * 64 + lane;
        double n0, n1;
        randn_pair(key, iter, (uint32_t)pair, n0, n1);
        p.c[j].x = (2 * pair < D) ? w.c[j].x * n0 : 0.0;
        p.c[j].y = (2 * pair + 1 < D) ? w.c[j].y * n1 : 0.0;
        sched_fence();
    }
    return p;
}

}  // namespace idhmc
