// idhmc_nuts_kernel.hpp -- one NUTS transition per chain, one chain per wavefront, and the initial-stepsize
// search (kernel templates; instantiated ahead of time for the built-in densities in idhmc_nuts.hip and at
// run time, through hipRTC, for a user-supplied density).  Replaces reference sample_tree / sample_trajectory / adjacent_tree / leaf / is_turning /
// combine_* (src/NUTS.jl:18-264, src/tree.jl:131-444) and find_initial_stepsize (src/stepsize.jl:51-164).
//
// The reference builds each doubling by recursion (adjacent_tree calls itself for the left and right
// half, src/tree.jl:335-346) with a bitmask arena for the live vectors (src/tree.jl:16-121).  Here each
// doubling is a flat loop over its 2^depth leaves.  After leaf n the sub-trees that are complete (one per
// trailing 1 bit of n) are merged bottom-up, exactly the post-order of the recursion, so turn checks,
// early exits, log-sum-exp association and RNG consumption are the same as the reference's.
//
// Where the state lives (D = 1024: one vector = 8 KiB):
//   VGPRs   q, p of the trajectory's moving end (64 regs) for the whole transition -- the separable
//           Gaussian gradients are recomputed from q (2 flops per element) instead of being carried;
//           rho of the sub-tree being merged (32 regs) during a merge cascade; in the one-wavefront-per-SIMD
//           form also the whole-tree rho and the level-2 p#_first (the compiler parks them in AGPRs).
//   LDS     mu, tau (and a shared M^-1) once per workgroup; per wavefront: the level-0 summary (momentum of
//           the previous leaf: rho and, times M^-1, p#), the level-1 summary as far as it fits (nuts_l1_lds),
//           level-2 rho in the one-wavefront form, a per-chain M^-1, and the per-level scalars.
//   HBM/L2  per-wavefront arena: deeper summaries (rho, p#_first), the far edge once it has left the starting
//           point (until then the state arrays s.q, s.p, s.g ARE the far edge), the whole-tree rho where it is
//           not on chip, regeneration checkpoints; proposal candidates only for general densities.
// Workgroups are persistent (one per CU) and pull chains from a device-wide queue; 512 < L <= 1024 runs two wavefronts per
// SIMD by default and has a one-per-SIMD form as well (DESIGN.md 3.3 has the counters behind every choice here).
#pragma once
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"
#include "idhmc_xchg.hpp"

namespace idhmc {

constexpr int kMaxDepth = 16;
// Proposal candidates are not stored: a candidate is (position on the trajectory, l(q), pi), and the winner's q is
// REGENERATED at the end by |position| leapfrog steps from the starting point (bit-identical: the trajectory is one
// deterministic leapfrog chain in each direction).  Storing every candidate's q was 2/3 of the kernel's HBM writes
// (rocprofv3 PMC at D = 1024, depth 7: 9 KB written per leaf, 3.2 TB/s); the regeneration costs ~25 % more leapfrogs.
// Separable densities only: a general density's leapfrog is expensive (the cooperative dense gradient would also
// wait for the longest regeneration of its 16-chain group: measured 2.5e8 -> 1.7e8), so it stores its candidates.
#ifndef IDHMC_ZETA_REGENERATE
#define IDHMC_ZETA_REGENERATE 1
#endif
__host__ __device__ constexpr bool nuts_regenerate(bool separable) { return IDHMC_ZETA_REGENERATE != 0 && separable; }
// Round 3: with regeneration a proposal candidate is a position on the trajectory, so NOTHING the tree's scalar bookkeeping
// computes (the log-sum-exps of the weights and of the acceptance statistic, the multinomial picks and their exponential draws)
// feeds back into the trajectory: tree shape, turn tests and divergences depend on the vectors alone.  The kernel therefore
// only LOGS every leaf's Delta (8 bytes) while it builds the tree and evaluates the bookkeeping afterwards, in the order the
// reference prescribes but 64 leaves per pass -- one lane per leaf (nuts_replay below).  Sequentially it was ~176 vector
// instructions per merge, i.e. per leaf, executed identically by all 64 lanes: a quarter of the kernel's instructions.
#ifndef IDHMC_NUTS_DEFER
#define IDHMC_NUTS_DEFER 1
#endif
__host__ __device__ constexpr bool nuts_defer(bool separable) { return IDHMC_NUTS_DEFER != 0 && nuts_regenerate(separable); }
// vectors of L doubles that hold one double per possible leaf of a transition (2^max_depth - 1 of them)
__host__ __device__ constexpr int nuts_dl_vectors(int max_depth, int L) { return ((1 << max_depth) + L - 1) / L; }
// Wavefronts per workgroup (one workgroup per CU): the phase point of a chain lives in VGPRs, so the register
// budget decides.  This function gives the BASE form; 512 < L <= 1024 also has the two-per-SIMD form of nuts_wide_waves,
// which the host prefers since round 2.  L > 512: 4 wavefronts, one per SIMD with 256 VGPRs + 256 AGPRs (beyond L = 1024
// two per SIMD spill 176-576 B and lose 25-75 %).  L <= 512: 8 wavefronts, two per
// SIMD -- they fit in 256 registers (8 spilled dwords at L = 512), and a single wavefront can only issue an fp64
// instruction every ~7 cycles.  Measured, separable, 8 vs 4 wavefronts: D = 256 1.0e9 vs 0.6e9 leapfrog/s; D = 512
// 4.6e8 vs 3.8e8 at depth 4, 6.9e8 vs 5.0e8 at depth 7.  L <= 256: 16 wavefronts, four per SIMD (121 registers at
// L = 256): D = 128 1.05e9 / 1.53e9 (depth 4 / 7) vs 0.89e9 / 1.09e9 with 8; D = 256 0.81e9 / 1.22e9 vs 0.74e9 / 1.00e9;
// L = 384 (round 2): 7.0e8 / 1.05e9 vs 6.4e8 / 9.1e8 with 8 (128 registers, 16 B of scratch); L = 512 does not fit four per
// SIMD (112 B of scratch: 4.6e8 / 6.6e8 vs 5.6e8 / 8.4e8 with two).
// A general density keeps 4:
// the dense MVN streams its 512 KiB matrix through L1 per gradient, and 8 concurrent streams per CU thrash it
// (63 M/s with 4 wavefronts, 36 M/s with 8).  A cooperative density (DenseMvnCoop, idhmc_device.hpp) runs 16: one per
// chain of its 16-row matrix-core tile.  IDHMC_NUTS_WAVES forces one value for the others (experiments).
// L = 2048 with a per-chain metric: 3 (mu, tau and three wavefronts' p_prev and M^-1 are 128 KB of LDS; four are 160).
__host__ __device__ constexpr int nuts_waves(int nch, bool separable, bool cooperative = false, bool shared_metric = true)
{
    if (cooperative) return 16;
    if (nch > 8 && !shared_metric && 10 * nch > 152) return 3;     // mu, tau + 4 x (p_prev, M^-1) must fit 152 KB
#ifdef IDHMC_NUTS_WAVES
    return IDHMC_NUTS_WAVES;
#else
    return separable ? (nch <= 3 ? 16 : (nch <= 4 ? 12 : 4)) : 4;      // L = 512: three per SIMD since round 3 (167 registers): +6 % over two
#endif
}

// Separable densities keep the level-1 sub-tree summary (rho and p#_first) in LDS next to the level-0 one: at
// L = 1024 the kernel is bound by the arena's traffic to the Infinity Cache (the live arena of the 128 wavefronts
// of an XCD is ~28 MB, its L2 4 MB; ~21 KB per leaf at 3.5e8 leaves/s), and half of the level >= 1 merges are
// level-1 merges.
// (when the workgroup's LDS allows: 2 more vectors per wavefront).
// Returns how many of the two level-1 vectors fit per wavefront: 2 = rho and p#_first, 1 = rho only, 0 = none.
// Budget: the CU's 160 KB less the per-level scalars (LevelScalars, 848 B per wavefront) and the kernel's other statics.
// Round 3: the level-0 summary (the previous leaf's momentum) is no longer staged in LDS by the separable forms: the momentum a
// leapfrog starts from IS the previous leaf's, so the odd leaves simply keep their input momentum in registers until the level-0
// merge (32 VGPRs at L = 1024, live only while rho / p#_first of a merge are not).  The vector this frees per wavefront holds
// the level-1 p#_first, which was the largest single source of arena traffic (one 8 KB store and one 8 KB load per four leaves:
// profiles/r03_nuts_bytes_by_source.json).  The register-rich form keeps its own layout.
#ifndef IDHMC_NUTS_PREV_REGS
#define IDHMC_NUTS_PREV_REGS 1
#endif
// The cooperative dense density takes both over (round 3): its merges of level >= 1 fetched rho and p#_first from the arena one level after
// the other -- the slowest of a workgroup's 16 wavefronts sets the pace of every gradient round -- and a CU has the LDS for the level-1 pair.
#ifndef IDHMC_COOP_PREV_REGS
#define IDHMC_COOP_PREV_REGS 1
#endif
__host__ __device__ constexpr bool nuts_prev_regs(bool separable, bool cooperative, bool rich)
{
    return IDHMC_NUTS_PREV_REGS != 0 && !rich && ((separable && !cooperative) || (cooperative && IDHMC_COOP_PREV_REGS != 0));
}
__host__ __device__ constexpr int nuts_l1_lds(int nch, bool separable, int waves, bool lds_params = true, bool shared_metric = true,
                                              bool prev_regs = false, bool cooperative = false)
{
    if (cooperative) {      // no parameters in LDS, a shared metric is read from L2; the d and T tiles of the gradient rounds
        if (!prev_regs) return 0;
        const int base = (shared_metric ? 1 : 0) + waves * (shared_metric ? 0 : 1);
        const int budget = (163840 - 912 * waves - 256 - 2 * 16 * (128 * nch + 2) * 8) / (1024 * nch);
        return base + 2 * waves <= budget ? 2 : (base + waves <= budget ? 1 : 0);
    }
    if (!separable) return 0;
    const int base = (lds_params ? 2 : 0) + (shared_metric ? 1 : 0) + waves * ((prev_regs ? 0 : 1) + (shared_metric ? 0 : 1));
    const int budget = (163840 - 848 * waves - 256) / (1024 * nch);     // vectors of L doubles
    return base + 2 * waves <= budget ? 2 : (base + waves <= budget ? 1 : 0);
}
// L = 1024, separable: the kernel exists in two forms and the host picks one per launch (launch_nuts): the default
// one wavefront per SIMD (level-1 summary in LDS, inlined merge scalars; best for adapted chains, depth ~4) and a
// WIDE one with two per SIMD (256 registers, level-1 summary in the arena; 11-18 % faster on deep trees, 3-12 %
// slower on shallow ones).
#ifndef IDHMC_WIDE_MAX_NCH
#define IDHMC_WIDE_MAX_NCH 8
#endif
__host__ __device__ constexpr int nuts_wide_waves(int nch, bool separable, bool cooperative = false)
{
    return (separable && !cooperative && nch > 4 && nch <= IDHMC_WIDE_MAX_NCH) ? 8 : 0;     // 0: no wide form
}

// "Register-rich" form: separable density, one wavefront per SIMD (4 per workgroup), L <= 1024.  Each wavefront owns
// the SIMD's whole 512-register file, so everything that is constant over a transition -- mu, tau and M^-1 (shared
// or per chain) -- lives in VGPRs (96 at L = 1024) instead of LDS: the leapfrog reads no memory at all and needs no
// scheduling fences, and the LDS they occupied holds deeper sub-tree summaries instead.
#ifndef IDHMC_NUTS_RICH
#define IDHMC_NUTS_RICH 1
#endif
__host__ __device__ constexpr bool nuts_rich(int nch, bool separable, bool cooperative, int waves)
{
    return IDHMC_NUTS_RICH != 0 && separable && !cooperative && waves == 4 && nch <= 8;
}
// within the register-rich form: mu, tau, M^-1 in VGPRs (1) or staged in LDS like the other forms (0)
#ifndef IDHMC_NUTS_CONST_REGS
#define IDHMC_NUTS_CONST_REGS 0
#endif
__host__ __device__ constexpr bool nuts_const_regs(int nch, bool separable, bool cooperative, int waves)
{
    return IDHMC_NUTS_CONST_REGS != 0 && nuts_rich(nch, separable, cooperative, waves);
}
// register-rich form: does rho of the level-2 summary fit in LDS next to the rest (else it stays in registers like the
// level-2 p#_first)?  Budget: 152 KB of the CU's 160 (the per-level scalars and the compiler's own use take the rest).
__host__ __device__ constexpr bool nuts_l2_lds(int nch, bool lds_params, bool shared_metric, bool const_regs)
{
    return ((const_regs ? 0 : (lds_params ? 2 : 0) + (shared_metric ? 1 : 0)) + 4 * (4 + ((const_regs || shared_metric) ? 0 : 1))) * nch <= 152;
}
template <bool B> struct BoolC { static constexpr bool value = B; };
template <bool C, class A, class B> struct CondT { typedef A type; };
template <class A, class B> struct CondT<false, A, B> { typedef B type; };

// arena vector indices (each vector = L doubles); MD = max_depth
struct ArenaMap {
    int md;
    bool regen;   // no candidate vectors
    int dlv = 0;  // vectors at the end that hold the log of the leaves' Delta (deferred tree bookkeeping, nuts_defer)
    // the trajectory edge that is not in registers (p, q and, for general densities, grad); while that edge is
    // still the starting point it is read from the state arrays (s.p, s.q, s.g) instead and these stay unwritten
    __host__ __device__ int edge_p() const { return 0; }
    __host__ __device__ int edge_q() const { return 1; }
    __host__ __device__ int edge_g() const { return 2; }
    __host__ __device__ int top_rho() const { return 3; }                         // forms that do not keep it in registers
    __host__ __device__ int stk_rho(int k) const { return 4 + k; }                // 1 <= k < md
    __host__ __device__ int pf(int s) const { return 4 + md + s; }                // s < md + 1
    __host__ __device__ int zq(int s) const { return 4 + 2 * md + 1 + (s - 1); }  // s in [1, md + 2]; !regen only
    // regen only: the phase point at which the doubling of depth d >= kCheckpointDepth started (regeneration of the
    // proposal walks from the nearest of these instead of from the starting point)
    __host__ __device__ int ck_q(int d) const { return 4 + 2 * md + 1 + 2 * d; }
    __host__ __device__ int ck_p(int d) const { return 4 + 2 * md + 2 + 2 * d; }
    __host__ __device__ int dl_at() const { return 4 + 2 * md + 1 + (regen ? 2 * md : md + 2); }
    __host__ __device__ int count() const { return dl_at() + dlv; }
};

// Doublings of at least 2^kCheckpointDepth leaves (16: measured 2..5, within 2 % of each other) leave their starting phase point in the arena (2 vector stores): the
// winner of the multinomial sampling lies in the last doubling with probability >= 1/2, and regenerating it from there
// takes ~2^(d-1) leapfrogs instead of ~1.5 * 2^d from the starting point (measured: see DESIGN 3.3).
#ifndef IDHMC_NT_STATE
#define IDHMC_NT_STATE 0
#endif
constexpr int kNt = IDHMC_NT_STATE ? kAuxNt : 0;
#ifndef IDHMC_CHECKPOINT_DEPTH
#define IDHMC_CHECKPOINT_DEPTH 4
#endif
constexpr int kCheckpointDepth = IDHMC_CHECKPOINT_DEPTH;

struct AccStat {  // reference AcceptanceStatistic, src/NUTS.jl:58-66
    double lsa;
    int steps;
};
// The scalar bookkeeping (log-sum-exp, exponential draws) is called from several places of the tree
// loop.  Inlined, the copies make the kernel ~60 KB of code and the wavefronts of a CU pair thrash the
// shared instruction cache (measured: every phase 5-10x over its instruction count).  Out of line there
// is one copy of each; the toolchain's interprocedural register allocation keeps the calls cheap.
__device__ __noinline__ double nuts_logaddexp(double x, double y) { return dlogaddexp(x, y); }
// The same goes for the remaining transcendental code of a transition (Box-Muller of the momentum refresh, exp / log
// of the acceptance rate and of dual averaging), for a second reason: inlined, their ~60 polynomial coefficients are
// materialised as VGPR pairs, hoisted out of the kernel's loops as invariants and then SPILLED (a 64-bit literal is
// two moves, which the register allocator does not rematerialise) -- half of the scratch traffic at two wavefronts
// per SIMD.  Out of line the coefficients live only inside the callee.
struct NormalPair { double a, b; };
__device__ __noinline__ NormalPair nuts_randn_pair(uint32_t k0, uint32_t k1, uint32_t chain, uint32_t iter, uint32_t pair)
{
    NormalPair r;
    randn_pair(RngKey{k0, k1, chain}, iter, pair, r.a, r.b);
    return r;
}
__device__ __noinline__ double nuts_dexp(double x) { return dexp(x); }
__device__ __noinline__ double nuts_dlog(double x) { return dlog(x); }
IDHMC_DEV AccStat combine_acc(AccStat a, AccStat b)  // src/NUTS.jl:68-70
{
    return AccStat{nuts_logaddexp(a.lsa, b.lsa), a.steps + b.steps};
}

// The scalar work of one merge -- logaddexp of the two acceptance sums (src/NUTS.jl:68-70), logaddexp of the
// two tree weights (src/tree.jl:241) and the exponential draw (src/NUTS.jl:33) -- done ONCE across lanes
// instead of three times in sequence: lane parity 0 carries the acceptance pair, parity 1 the weight pair,
// and every lane also carries the draw.  Each lane executes exactly the operation sequence of dlogaddexp /
// randexp for its own operands (one shared dexp, one shared dlog, one shared division), so the three results
// are bit-identical to the sequential form; they are read back with v_readlane.  The draw is speculative
// (a pure function of its address): the caller consumes it only if the reference would have drawn.
struct MergeScalars { double lsa, omega; };
__device__ __forceinline__ MergeScalars nuts_merge_scalars_body(double lsa_a, double lsa_b, double om_a, double om_b)
{
#ifdef IDHMC_X1
    { MergeScalars o; o.lsa = lsa_a < lsa_b ? lsa_b : lsa_a; o.omega = (om_a < om_b ? om_b : om_a) + 0.5; return o; }
#endif
    const bool odd = (threadIdx.x & 1) != 0;
    const double x = odd ? om_a : lsa_a, y = odd ? om_b : lsa_b;
    // dlogaddexp(x, y), opened up (idhmc_math.hpp)
    const bool fin = dfinite(x) && dfinite(y);
    const bool xg = x > y;
    const double hi = xg ? x : y;
    const double t = dexp(xg ? y - x : x - y);          // in (0, 1] when fin
    const double u = 1.0 + t;
    const double lg = dlog(u);
    // dlog1p(t) = (u == 1) ? t : (u == inf ? u : dlog(u) * (t / (u - 1)))
    const double l1p = (u == 1.0) ? t : ((u == kInf) ? u : lg * (t / (u - 1.0)));
    const double lae = fin ? hi + l1p : hi;
    MergeScalars o;
    o.lsa = read_lane(lae, 0);
    o.omega = read_lane(lae, 1);
    return o;
}
__device__ __noinline__ MergeScalars nuts_merge_scalars(double lsa_a, double lsa_b, double om_a, double om_b)
{
    return nuts_merge_scalars_body(lsa_a, lsa_b, om_a, om_b);
}
// With one wavefront per SIMD (512 registers) the merge cascade inlines it: the scheduler then interleaves this
// dependent chain (exp, log, divide) with the independent one of the turn test (two fma chains and their DPP
// reductions), which a call boundary forbids: +3 % at depth 4, +6.5 % at depth 7 at L = 1024.  At two or four
// wavefronts per SIMD the inlined coefficients cost registers the kernel does not have.
template <bool INLINE>
IDHMC_DEV MergeScalars merge_scalars(double lsa_a, double lsa_b, double om_a, double om_b)
{
    if constexpr (INLINE) return nuts_merge_scalars_body(lsa_a, lsa_b, om_a, om_b);
    else return nuts_merge_scalars(lsa_a, lsa_b, om_a, om_b);
}
// The exponential draws of a transition are addressed (seed, chain, transition, draw index), so 64
// consecutive draws are produced by ONE Philox + log pass, lane l holding draw base + l; a merge reads its
// draw with v_readlane.  Same values as randexp(key, iter, draw) one at a time.
__device__ __noinline__ double nuts_randexp_batch(uint32_t k0, uint32_t k1, uint32_t chain, uint32_t iter, uint32_t base)
{
    return randexp(RngKey{k0, k1, chain}, iter, base + (threadIdx.x & 63));
}

// is_turning, src/NUTS.jl:148-170: both dot products in one pass.  p#_a is given, p#_b = M^-1 .* pb.
template <int NCH, class Metric>
IDHMC_DEV void turn_dots(const Vec<NCH> &rho, const Vec<NCH> &psa, const Vec<NCH> &pb, const Metric &minv,
                         double &da, double &db)
{
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mv = minv.get(j);
        a0 = dfma(rho.c[j].x, psa.c[j].x, a0);
        a1 = dfma(rho.c[j].y, psa.c[j].y, a1);
        b0 = dfma(rho.c[j].x, mv.x * pb.c[j].x, b0);
        b1 = dfma(rho.c[j].y, mv.y * pb.c[j].y, b1);
    }
    wave_sum2(a0, a1, b0, b1, da, db);
}

// the same with p#_a = M^-1 .* pa formed on the fly as well (the product is rounded exactly as psharp() rounds it, so
// the bits equal turn_dots(rho, psharp(pa), pb)): the whole-tree test reads the far edge's momentum, no stored p#
template <int NCH, class Metric>
IDHMC_DEV void turn_dots_pp(const Vec<NCH> &rho, const Vec<NCH> &pa, const Vec<NCH> &pb, const Metric &minv,
                            double &da, double &db)
{
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mv = minv.get(j);
        a0 = dfma(rho.c[j].x, mv.x * pa.c[j].x, a0);
        a1 = dfma(rho.c[j].y, mv.y * pa.c[j].y, a1);
        b0 = dfma(rho.c[j].x, mv.x * pb.c[j].x, b0);
        b1 = dfma(rho.c[j].y, mv.y * pb.c[j].y, b1);
    }
    wave_sum2(a0, a1, b0, b1, da, db);
}

// calculate_p# (src/kinetic_energy.jl:39-46): M^-1 .* p
template <int NCH, class Metric>
IDHMC_DEV Vec<NCH> psharp(const Metric &minv, const Vec<NCH> &p)
{
    Vec<NCH> r;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const double2 mv = minv.get(j);
        r.c[j] = make_double2(mv.x * p.c[j].x, mv.y * p.c[j].y);
    }
    return r;
}
template <int NCH>
IDHMC_DEV Vec<NCH> vadd(const Vec<NCH> &a, const Vec<NCH> &b)
{
    Vec<NCH> r;
#pragma unroll
    for (int j = 0; j < NCH; ++j) r.c[j] = make_double2(a.c[j].x + b.c[j].x, a.c[j].y + b.c[j].y);
    return r;
}
template <int NCH>
IDHMC_DEV Vec<NCH> lds_load(const double2 *p)   // p is lane-offset
{
    Vec<NCH> v;
#pragma unroll
    for (int j = 0; j < NCH; ++j) v.c[j] = p[j * 64];
    return v;
}
template <int NCH>
IDHMC_DEV void lds_store(double2 *p, const Vec<NCH> &v)
{
#pragma unroll
    for (int j = 0; j < NCH; ++j) p[j * 64] = v.c[j];
}

// All control flow in the transition is wave-uniform (one chain per wavefront), but values that pass
// through the vector ALU or LDS look divergent to the compiler.  uni() / usi() hand it the proof, so
// branches become scalar and loop state stays in SGPRs.
IDHMC_DEV bool uni(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
IDHMC_DEV int usi(int x) { return __builtin_amdgcn_readfirstlane(x); }

// per-wavefront scalars of the live sub-tree summaries; every lane reads/writes the same address with
// the same value
struct LevelScalars {
    double omega[kMaxDepth];
    double lsa[kMaxDepth];
    int steps[kMaxDepth];
    int zeta[kMaxDepth];
    int pf[kMaxDepth];
    double z_lq[kMaxDepth + 4];
    double z_pi[kMaxDepth + 4];
    int z_idx[kMaxDepth + 4];     // signed position of the candidate on the trajectory (kRegenerate)
    int ck_pos[kMaxDepth];        // signed position of the checkpoint of the doubling of depth d (kRegenerate)
};

// dynamic LDS layout (doubles): [mu L][tau L] if the density has parameters, [M^-1 L] if the metric is
// shared, then per wavefront [p_prev L] and, for a per-chain metric, [M^-1 L].
// A general (non-separable) density adds one staging vector per wavefront and keeps its parameters in L2;
// a cooperative one has the workgroup's [16][L + 2] tile instead.
__host__ __device__ inline size_t nuts_lds_doubles(int L, bool lds_params, bool shared_metric, bool separable,
                                                   bool cooperative = false, int waves = 0)
{
    if (waves == 0) waves = nuts_waves(L / 128, separable, cooperative, shared_metric);
    if (nuts_rich(L / 128, separable, cooperative, waves)) {    // per wavefront: p_prev, the level-1 summary, rho of level 2
        const bool cr = nuts_const_regs(L / 128, separable, cooperative, waves);
        const int l2 = nuts_l2_lds(L / 128, lds_params, shared_metric, cr) ? 1 : 0;
        return (size_t)L * (waves * (3 + l2 + ((cr || shared_metric) ? 0 : 1)) + (cr ? 0 : (lds_params ? 2 : 0) + (shared_metric ? 1 : 0)));
    }
    // per wavefront: [p_prev, or one scratch vector when nothing else is there] [per-chain M^-1] [general: staging] [level-1 rho, p#]
    const bool pr = nuts_prev_regs(separable, cooperative, false);
    const int l1n = nuts_l1_lds(L / 128, separable, waves, lds_params, shared_metric, pr, cooperative);
    const int first = pr ? (l1n == 0 ? 1 : 0) : 1;
    return (size_t)L * ((lds_params ? 2 : 0) + (shared_metric ? 1 : 0) +
                        waves * (first + (shared_metric ? 0 : 1) + ((separable || cooperative) ? 0 : 1) + l1n)) +
           (cooperative ? (size_t)2 * 16 * (L + 2) : 0);      // the d tile and the T tile
}

}  // namespace idhmc
#include "idhmc_nuts_replay.hpp"     // nuts_replay: the deferred bookkeeping (separable densities)
namespace idhmc {

enum : int { kPfLeaf = -1, kPfLevel0 = -2, kPfLevel1 = -3, kPfLevel2 = -4 };

// diagnostic build only: per-phase shader-cycle sums (never in the shipped library)
// diagnostic build only (-DIDHMC_BYTES, tools/nuts_bytes.sh): vectors moved between the wavefront and memory, by source, summed over the
// launch into the debug counters [2..9] (the slots the cycle stamps use: one or the other).  Sources: 0 prologue (q, p0, per-chain M^-1),
// 1 edge swap at a change of direction, 2 regeneration checkpoints (stores) and the regeneration's start point (loads), 3 level-1 summary
// in the arena, 4 level >= 2 summaries, 5 whole-tree statistic (far edge's momentum, whole-tree rho), 6 epilogue (q, grad l), 7 stored candidates
#ifdef IDHMC_BYTES
#define BYTES_DECL unsigned int by_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BYTES(i, nv) (by_acc[(i)] += (unsigned int)(nv))
#define BYTES_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(s.total_steps + 2 + i_, (unsigned long long)by_acc[i_] * (unsigned long long)(L * 8)); } while (0)
#else
#define BYTES_DECL
#define BYTES(i, nv)
#define BYTES_FLUSH
#endif
#ifdef IDHMC_STAMPS
#define STAMP_DECL long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long st_t = clock64()
#define STAMP(i) do { const long long t_ = clock64(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#define STAMP_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(s.total_steps + 2 + i_, (unsigned long long)st_acc[i_]); } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// ---- several transitions per launch (DevState::n_iter > 1) ---------------------------------------------------------------------------
// The queue then hands out (transition, chain) pairs, transition-major.  A chain's transitions are a whole sweep of tickets apart, so the
// previous one has almost always finished when its successor is handed out; where it has not (few chains, one very long tree) the taker
// waits for the chain's count in DevState::iters_done.  What this buys: a launch ends with every wavefront finishing its last tree while
// the queue is empty -- ~1 ms of a 3.7 ms transition at configs[3] (16 384 dense chains are 4 per resident wavefront;
// profiles/r03_dense_nuts_two_chains_per_wavefront.log) -- and that tail is now paid once per launch, not once per transition.
// A chain's state passes between workgroups through memory inside ONE launch.  The chains are therefore cut into as many contiguous
// ranges as there are XCDs (8) with a queue each, and workgroup b serves range b mod 8: workgroups b and b + 8 share an XCD, hence its L2,
// which is the point of coherence of everything they read and write.  The hand-over then needs no L2 write-back (an agent-scope release
// per transition wrote the XCD's dirty tree arena back every time: measured, the whole gain was gone): the producer waits for its
// stores to reach L2 (vmcnt) and publishes the count with a relaxed agent-scope atomic; the consumer reads the count the same way and
// reads the chain's state with agent-scope loads (ld_fresh, kAuxFresh below), which its CU's L1 does not serve.  That workgroups of equal b mod 8 share an XCD is an observation, not a contract, so the kernel CHECKS it:
// every workgroup registers its XCD's id (HW_REG_XCC_ID) for its range, and a second id for one range raises the abort code
// (IDHMC_ERR_HIP) instead of letting a chain be read through the wrong L2.
// Results do not depend on who runs a transition: every random number is addressed by (seed, chain, transition), so fused and single
// launches are bit-identical (tests/test_gpu_fused.py).
// The wait is one block of assembly: as a loop of the program it would be the innermost loop of the kernel to the register allocator,
// which then spills what is live across it.  Bounded (~4 s): returns false when the count never came.
IDHMC_DEV bool nuts_wait_iter(const uint32_t *word, uint32_t need)
{
    int ok, cnt, tv;
    asm volatile(
        "s_mov_b32 %[cnt], 0x1000000\n"
        ".Lnwi_top%=:\n"
        "global_load_dword %[tv], %[addr], off sc1\n"
        "s_waitcnt vmcnt(0)\n"
        "v_readfirstlane_b32 %[ok], %[tv]\n"
        "s_cmp_ge_u32 %[ok], %[need]\n"
        "s_cbranch_scc1 .Lnwi_yes%=\n"
        "s_sleep 8\n"
        "s_sub_u32 %[cnt], %[cnt], 1\n"
        "s_cmp_lg_u32 %[cnt], 0\n"
        "s_cbranch_scc1 .Lnwi_top%=\n"
        "s_mov_b32 %[ok], 0\n"
        "s_branch .Lnwi_end%=\n"
        ".Lnwi_yes%=:\n"
        "s_mov_b32 %[ok], 1\n"
        ".Lnwi_end%=:\n"
        : [ok] "=&s"(ok), [cnt] "=&s"(cnt), [tv] "=&v"(tv)
        : [addr] "v"(word), [need] "s"(need)
        : "scc", "memory");
    return ok != 0;
}
// Loads of a chain's state (what an earlier transition of the chain wrote, possibly on another CU of the XCD within this launch) are
// agent-scope: served by L2, never by this CU's L1, which nothing refreshes (an L1 invalidate per hand-over instead cost the cooperative
// dense kernel ~10 %: the taker holds up its workgroup's round for the 2-7 us it takes).  Own stores are safe in any case (write-through).
constexpr int kAuxFresh = 16;     // sc1 on gfx950's buffer loads
constexpr uint32_t kTestXccFlag = 1u << 30;     // transition flag of the test suite only: see the XCD check in k_nuts
template <class T>
IDHMC_DEV T ld_fresh(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IDHMC_DEV uint32_t nuts_peek_iter(const uint32_t *word)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

#ifndef IDHMC_COOP_REFILL
#define IDHMC_COOP_REFILL 1
#endif
#ifndef IDHMC_COOP_PREFETCH
#define IDHMC_COOP_PREFETCH 0
#endif
template <int NCH, class Model, bool SHARED_METRIC,
          int WAVES = nuts_waves(NCH, Model::kSeparable, Model::kCooperative, SHARED_METRIC)>
#ifdef IDHMC_NUTS_VGPR_CAP     // experiments: how many registers does the kernel really need?
__attribute__((amdgpu_num_vgpr(IDHMC_NUTS_VGPR_CAP)))
#endif
__global__ __launch_bounds__(WAVES * 64, (WAVES + 3) / 4)
void k_nuts(DevState s, uint32_t iter0, uint32_t flags)
{
    constexpr int kNutsWaves = WAVES;
    constexpr bool kCoop = Model::kCooperative;
    constexpr bool kCoopRefill = IDHMC_COOP_REFILL != 0;
    constexpr int kCoopPrefetch = IDHMC_COOP_PREFETCH;      // levels of arena summaries (2, 3) requested ahead of the gradient round
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ LevelScalars Sall[kNutsWaves];
    __shared__ int coop_ctl[2];           // cooperative density: {chain group, chains of it still alive}
    // Counters every chain adds to (leapfrog steps; the 39 scalar diagnostics counters) are summed per workgroup here and reach the global
    // words once, when the workgroup leaves: 65 536 atomics per launch on ONE address cost ~0.2 ms each hot address (3.5 -> 2.5 ms per
    // transition with IDHMC_T_ACCUM_DIAG at configs[2], tools/bench_accum.py).  Integers: the order of the additions is immaterial.
    constexpr int kWgAcc = 40;            // [0..38] diagnostics counters, [39] leapfrog steps
    __shared__ unsigned long long wg_acc[kWgAcc];
    constexpr int L = 128 * NCH;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: branches on it stay wave-uniform
    LevelScalars &S = Sall[wv];
    constexpr bool kRegenerate = nuts_regenerate(Model::kSeparable);
    constexpr bool kDefer = nuts_defer(Model::kSeparable);       // the tree's scalar bookkeeping is evaluated after the tree (nuts_replay)
    const ArenaMap am{s.max_depth, kRegenerate, kDefer ? nuts_dl_vectors(s.max_depth, 128 * NCH) : 0};
    double *const arena = s.arena + ((int64_t)blockIdx.x * kNutsWaves + wv) * s.arena_stride;

    // ---- stage the shared read-only vectors in LDS, once per workgroup (register-rich form: in VGPRs) ----------
    double *cursor = lds;
    Model mdl;
    constexpr bool kRich = nuts_rich(NCH, Model::kSeparable, kCoop, kNutsWaves);
    constexpr bool kConstRegs = nuts_const_regs(NCH, Model::kSeparable, kCoop, kNutsWaves);
    constexpr bool kPrevRegs = nuts_prev_regs(Model::kSeparable, kCoop, kRich);     // level-0 summary in registers, not LDS
    constexpr int kL1N = kRich ? 2 : nuts_l1_lds(NCH, Model::kSeparable, kNutsWaves, Model::kHasParams && Model::kSeparable, SHARED_METRIC, kPrevRegs, kCoop);
#ifdef IDHMC_X3
    constexpr bool kL1Rho = kL1N >= 1, kL1Pf = kL1N >= 1;
#else
    constexpr bool kL1Rho = kL1N >= 1, kL1Pf = kL1N >= 2;     // level-1 summary in LDS: rho / p#_first
#endif
    constexpr bool kL2 = kRich;    // level-2 summary on chip as well: rho in LDS, p#_first in registers
    // LDS vectors per wavefront: [p_prev | scratch], [per-chain M^-1], [general density: staging], [level-1 rho, p#], [level-2 rho];
    // kPrevRegs: no p_prev vector; the momentum refresh's scratch is then the level-1 rho slot (or one vector of its own)
    constexpr int kMetricVec = (SHARED_METRIC || kConstRegs) ? 0 : 1;
    constexpr bool kL2Lds = kL2 && nuts_l2_lds(NCH, Model::kHasParams, SHARED_METRIC, kConstRegs);
    constexpr int kFirstVec = kPrevRegs ? (kL1N == 0 ? 1 : 0) : 1;
    constexpr int kPerWave = kFirstVec + kMetricVec + ((Model::kSeparable || kCoop) ? 0 : 1) + kL1N + (kL2Lds ? 1 : 0);
    if constexpr (Model::kHasParams && Model::kSeparable) {
        if constexpr (kConstRegs) {
            mdl.load(s.mu, s.tau, lane);
        } else {
            double *lmu = cursor, *ltau = cursor + L;
            cursor += 2 * L;
            for (int i = threadIdx.x; i < L; i += kNutsWaves * 64) { lmu[i] = s.mu[i]; ltau[i] = s.tau[i]; }
            mdl.m = reinterpret_cast<const double2 *>(lmu) + lane;
            mdl.t = reinterpret_cast<const double2 *>(ltau) + lane;
        }
    }
    typename CondT<kConstRegs, Vec<NCH>, LdsVec>::type minv;
    if constexpr (kConstRegs) {
        if constexpr (SHARED_METRIC) minv = bload<NCH>(s.minv, lane);
    } else if constexpr (SHARED_METRIC) {
        double *lm = cursor;
        cursor += L;
        for (int i = threadIdx.x; i < L; i += kNutsWaves * 64) lm[i] = s.minv[i];
        minv.p = reinterpret_cast<const double2 *>(lm) + lane;
    }
    double *my = cursor + (size_t)wv * (kPerWave * L);
    if constexpr (kMetricVec) minv.p = reinterpret_cast<const double2 *>(my + kFirstVec * L) + lane;
    // level-1 summary (kL1Rho, kL1Pf): rho and p#_first of the parked two-leaf sub-tree; level-2 (kL2): rho
    constexpr int kL1At = kFirstVec + kMetricVec + ((Model::kSeparable || kCoop) ? 0 : 1);
    // level-0 summary (previous leaf's momentum) where it is staged in LDS; always the momentum refresh's scratch vector
    double2 *const pprev = reinterpret_cast<double2 *>(my + (kFirstVec ? 0 : kL1At) * L) + lane;
    double2 *const l1rho = reinterpret_cast<double2 *>(my + kL1At * L) + lane;
#ifdef IDHMC_X3
    double2 *const l1pf = reinterpret_cast<double2 *>(my + (kL1At + (kL1N >= 2 ? 1 : 0)) * L) + lane;
#else
    double2 *const l1pf = reinterpret_cast<double2 *>(my + (kL1At + 1) * L) + lane;
#endif
    double2 *const l2rho = reinterpret_cast<double2 *>(my + (kL1At + 2) * L) + lane;
    if constexpr (kCoop) mdl.init(s, cursor + (size_t)kNutsWaves * (kPerWave * L), &coop_ctl[1], lane, wv);
    else if constexpr (!Model::kSeparable) mdl.init(s, my + (kFirstVec + kMetricVec) * L, lane);   // general density: one LDS vector
    if constexpr (kCoop && kCoopRefill) { if (threadIdx.x == 0) coop_ctl[1] = kNutsWaves; }     // wavefronts that may still request a gradient
    if (threadIdx.x < kWgAcc) wg_acc[threadIdx.x] = 0ull;
    __syncthreads();

    const uint32_t n_iter = s.n_iter < 1u ? 1u : s.n_iter;      // transitions per chain in this launch
    // this workgroup's range of chains and its queue (see above); fewer than 8 workgroups: one range each
    const uint32_t nparts = gridDim.x < 8u ? gridDim.x : 8u, part = blockIdx.x % nparts;
    const uint32_t part_lo = (uint32_t)((uint64_t)s.C * part / nparts);
    const uint32_t part_n = (uint32_t)((uint64_t)s.C * (part + 1u) / nparts) - part_lo;
    if (n_iter > 1u && nparts == 8u && threadIdx.x == 0) {
        uint32_t me = ((uint32_t)__builtin_amdgcn_s_getreg(6164) & 15u) + 1u;            // hwreg(HW_REG_XCC_ID, 0, 4)
        if (flags & kTestXccFlag) me = ((blockIdx.x >> 3) & 1u) + 1u;                    // (tests: workgroups of one range disagree)
        const uint32_t was = atomicCAS(s.queue + 8 + part, 0u, me);
        if (was != 0u && was != me) {       // the abort word stops the launch, the status makes the drivers report it
            atomicMax(s.total_steps + 1, (unsigned long long)IDHMC_ERR_HIP);
            if (part_n > 0u) s.status[part_lo] = IDHMC_ERR_HIP;
        }
    }
    for (;;) {
        uint32_t cu = 0, it = 0;
        // ticket -> (transition, chain) of this workgroup's range; false: nothing left (or a chain raised the abort code: no further
        // transitions are started)
        auto ticket = [&]() -> bool {
            if (lane == 0) cu = atomicAdd(s.queue + part, 1u);
            cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)cu);
            if (cu >= part_n * n_iter) return false;
            if (n_iter > 1u) {
                if (nuts_peek_iter(reinterpret_cast<const uint32_t *>(s.total_steps + 1)) != 0u) return false;
                it = cu / part_n;
                cu -= it * part_n;
            }
            cu += part_lo;
            return true;
        };
        if constexpr (kCoop && kCoopRefill) {
            // Round 3: every wavefront takes its next chain by itself, as the other forms do.  In groups of 16 (below) a wavefront whose
            // chain finished early only served the others' rounds until the slowest of the group was done -- 15 % of its cycles at
            // configs[3] (stamps); now it goes on with a new chain at once, and the others wait only for its epilogue and
            // prologue (no gradient request there), once per transition.  `alive` counts the wavefronts that may still request: a
            // wavefront leaves it when the queue is empty, and serves rounds until everybody has (all leave together).
            if (!ticket()) {
                mdl.retire();
                mdl.serve();
                break;
            }
            if (n_iter > 1u) {
                // the chain's previous transition may still run elsewhere: this wavefront keeps serving the group's rounds meanwhile
                // (waiting outside them would stop the very workgroup that may hold it)
                while (nuts_peek_iter(s.iters_done + cu) < it) mdl.serve_round();
            }
        } else if constexpr (kCoop) {
            // the workgroup takes chains in groups of 16 (one matrix-core tile); the queue counts groups
            __syncthreads();              // every wavefront is done with the previous group's control words
            if (threadIdx.x == 0) {
                const uint32_t grp = atomicAdd(s.queue, 1u);
                const int64_t left = s.C - (int64_t)grp * 16;
                coop_ctl[0] = (int)grp;
                coop_ctl[1] = left <= 0 ? 0 : (left > 16 ? 16 : (int)left);
            }
            __syncthreads();
            const uint32_t grp = (uint32_t)usi(*reinterpret_cast<volatile int *>(&coop_ctl[0]));
            if ((int64_t)grp * 16 >= s.C) break;
            cu = grp * 16u + (uint32_t)wv;
            if ((int64_t)cu >= s.C) {     // ragged last group: no chain for this wavefront, it only serves
                mdl.serve();
                continue;
            }
        } else {
            if (!ticket()) break;
            if (n_iter > 1u) {
                if (it > 0u && !nuts_wait_iter(s.iters_done + cu, it)) {
                    if (lane == 0) { atomicMax(s.total_steps + 1, (unsigned long long)IDHMC_ERR_HIP); s.status[cu] = IDHMC_ERR_HIP; }   // never came: abort the launch
                    break;
                }
            }
        }
        const uint32_t iter = iter0 + it;
        const int64_t c = (int64_t)cu;
        const RngKey key{s.k0, s.k1, s.first_chain + cu};
        const int64_t off = c * L;
        STAMP_DECL;
        BYTES_DECL;

        // ---- sample_tree prologue (src/NUTS.jl:251-260) -----------------------------------------
        Vec<NCH> q = bload<NCH, kNt | kAuxFresh>(s.q + off, lane);  BYTES(0, 1);
        Vec<NCH> g;                     // carried only for general densities (separable ones recompute it)
        if constexpr (!Model::kSeparable) { g = bload<NCH, kNt | kAuxFresh>(s.g + off, lane); BYTES(0, 1); }
        if constexpr (!SHARED_METRIC) {
            if constexpr (kConstRegs) { minv = bload<NCH>(s.minv + off, lane); BYTES(0, 1); }
            else { lds_store<NCH>(reinterpret_cast<double2 *>(my + kFirstVec * L) + lane, bload<NCH>(s.minv + off, lane)); BYTES(0, 1); }
        }
        Vec<NCH> p;
        if (flags & IDHMC_T_KEEP_P) {
            p = bload<NCH, kAuxFresh>(s.p + off, lane);  BYTES(0, 1);
        } else {
            // rand_p! (:254), one 128-element chunk per trip through a ROLLED loop staged in this
            // wavefront's LDS scratch vector: unrolled, the eight Box-Muller bodies are 20 KB of
            // straight-line code that every transition streams through the instruction cache once.
            // W is fetched in one burst into the same LDS vector first (a global load inside the rolled
            // loop would expose one full memory latency per chunk).
            lds_store<NCH>(pprev, bload<NCH>(s.w + c * s.minv_stride, lane));
#pragma unroll 1
            for (int j = 0; j < NCH; ++j) {
                const int pair = j * 64 + lane;
                const NormalPair nn = nuts_randn_pair(key.k0, key.k1, key.chain, iter, (uint32_t)pair);
                const double2 wj = pprev[j * 64];
                pprev[j * 64] = make_double2((2 * pair < s.D) ? wj.x * nn.a : 0.0, (2 * pair + 1 < s.D) ? wj.y * nn.b : 0.0);
            }
            p = lds_load<NCH>(pprev);
        }
        // p0 stays in the state array: the starting point (s.q, s.p, s.g) doubles as the far edge of the trajectory
        // until that side is extended, and the regeneration of the proposal starts from it
        if (!(flags & IDHMC_T_KEEP_P)) { bstore<NCH>(s.p + off, lane, p); BYTES(0, 1); }
        STAMP(6);                       // momentum refresh
        uint32_t dirs = (flags & IDHMC_T_USE_DIRECTIONS) ? s.directions[c] : rand_directions(key, iter);  // :252
        dirs = (uint32_t)usi((int)dirs);
        const double eps = ld_fresh(s.eps + c);
        const double lq0 = ld_fresh(s.lq + c);
        const double pi0 = phase_logdensity(lq0, kinetic_energy<NCH>(minv, p));  // :260
        // randexp draws of this transition, 64 per batch (src/NUTS.jl:33; RNG address = draw index)
        uint32_t draw = 0, ebase = 0;
        double ebatch = 0.0;
        if constexpr (!kDefer) ebatch = nuts_randexp_batch(key.k0, key.k1, key.chain, iter, 0u);
        auto take_draw = [&]() -> double {
            if (draw >= ebase + 64u) {
                ebase += 64u;
                ebatch = nuts_randexp_batch(key.k0, key.k1, key.chain, iter, ebase);
            }
            const double e = read_lane(ebatch, usi((int)(draw - ebase)));
            ++draw;
            return e;
        };

        // ---- sample_trajectory initial leaf (src/tree.jl:388-393) ---------------------------------
        // whole-tree turn statistic: rho in registers (register-rich form) or in the arena; the p# of the two ends
        // are not kept -- the tests form them from the momenta of the edges (turn_dots_pp)
        Vec<NCH> top_rho_r;             // kRich
        Vec<NCH> l2pf_r;                // kL2: p#_first of the parked level-2 sub-tree
        Vec<NCH> l2rho_r;               // kL2 && !kL2Lds: its rho
        // Forms with the level-1 rho in LDS park the whole-tree rho in that slot between doublings (nothing is parked
        // there then): a doubling of one or two leaves never needs the slot, a longer one moves the vector to the
        // arena when its first two-leaf sub-tree is parked.
        constexpr bool kTopLds = !kRich && kL1Rho;
        bool top_in_lds = kTopLds;
        if constexpr (kRich) top_rho_r = p;
        else if constexpr (kTopLds) lds_store<NCH>(l1rho, p);
        else { bstore<NCH>(arena + (int64_t)am.top_rho() * L, lane, p); BYTES(5, 1); }
        STAMP(0);                       // prologue
        int top_zeta = 0;               // slot 0 = the starting point itself (lives in s.q / s.g)
        double top_omega = 0.0;
        AccStat v{-kInf, 0};
        S.z_lq[0] = lq0;
        S.z_pi[0] = pi0;
        uint32_t zfree = ((1u << (s.max_depth + 2)) - 1u) << 1;   // zeta slots 1..md+2 free
        uint32_t pffree = (1u << (s.max_depth + 1)) - 1u;         // p#_first slots 0..md free
        uint32_t ckmask = 0;            // depths whose doubling left a checkpoint
        int regs_edge = 1;              // registers hold the '+' edge; the arena holds the '-' edge
        int i_minus = 0, i_plus = 0, depth = 0;
        int term_left = 1, term_right = 0;                        // REACHED_MAX_DEPTH, src/tree.jl:300
        // kDefer: the record nuts_replay works from -- Delta of every leaf, where each doubling started, where the tree stopped
        double *const dlog = arena + (int64_t)am.dl_at() * L;
        int stop_kind = 0, stop_n = 0, stop_k = 0;
        uint32_t fwdmask = 0;

        while (depth < s.max_depth) {                             // src/tree.jl:395
            const int fwd = (int)(dirs & 1u);                     // next_direction :152-155
            dirs >>= 1;
            if (fwd != regs_edge) {                               // continue from the other edge (:398-404)
                // the edge in registers goes to the arena unless it is still the starting point (which the state
                // arrays hold); the other edge comes from the arena, or from the state arrays while it is the start
                const int i_regs = regs_edge ? i_plus : i_minus, i_other = regs_edge ? i_minus : i_plus;
                if (i_regs != 0 || i_other != 0) {
                    const double *src_p = i_other ? arena + (int64_t)am.edge_p() * L : s.p + off;
                    const double *src_q = i_other ? arena + (int64_t)am.edge_q() * L : s.q + off;
                    const Vec<NCH> op = bload<NCH, kAuxFresh>(src_p, lane);
                    const Vec<NCH> oq = bload<NCH, kAuxFresh>(src_q, lane);  BYTES(1, 2);
                    Vec<NCH> og;
                    if constexpr (!Model::kSeparable) {
                        og = bload<NCH, kAuxFresh>(i_other ? arena + (int64_t)am.edge_g() * L : s.g + off, lane);  BYTES(1, 1);
                    }
                    if (i_regs != 0) {
                        bstore<NCH>(arena + (int64_t)am.edge_p() * L, lane, p);
                        bstore<NCH>(arena + (int64_t)am.edge_q() * L, lane, q);  BYTES(1, 2);
                        if constexpr (!Model::kSeparable) { bstore<NCH>(arena + (int64_t)am.edge_g() * L, lane, g); BYTES(1, 1); }
                    }
                    p = op; q = oq;
                    if constexpr (!Model::kSeparable) g = og;
                }
                regs_edge = fwd;
            }
            const int i_start = fwd ? i_plus : i_minus;
            const int sgn = fwd ? 1 : -1;
            if constexpr (kRegenerate) {
                if (depth >= kCheckpointDepth && i_start != 0) {                  // (position 0 is the state arrays themselves)
                    bstore<NCH, kNt>(arena + (int64_t)am.ck_q(depth) * L, lane, q);
                    bstore<NCH, kNt>(arena + (int64_t)am.ck_p(depth) * L, lane, p);  BYTES(2, 2);
                    S.ck_pos[depth] = i_start;
                    ckmask |= 1u << depth;
                }
            }
            const double eps_dir = fwd ? eps : -eps;              // move, src/NUTS.jl:18-21
            const int nleaves = 1 << depth;
            if constexpr (kDefer) {
                S.z_idx[depth] = i_start;                         // (the candidates' position array is free in this form)
                fwdmask |= (uint32_t)fwd << depth;
            }

            // ---- adjacent_tree(depth), src/tree.jl:321-366, as a flat loop over its leaves --------
            bool invalid = false;
            AccStat vres{-kInf, 0};
            Vec<NCH> rho;                 // rho of the sub-tree being merged
            bool has_rho = false;
            int cur_zeta = -1, cur_pf = kPfLeaf, i_n = i_start;
            double cur_omega = 0.0;
            AccStat cur_v{-kInf, 0};
            for (int n = 0; n < nleaves; ++n) {
                double lq, K;
                // kPrevRegs: the momentum this leapfrog starts from is the previous leaf's -- the level-0 summary an odd leaf merges with
                Vec<NCH> p_in;
                if constexpr (kPrevRegs) p_in = p;
                // cooperative density: the level-2 (and -3) summaries this leaf's cascade will merge with are requested BEFORE the gradient
                // round, whose ~8 us hide the arena's latency (the slowest of the workgroup's 16 cascades sets the pace of every round,
                // and some wavefront has a level >= 2 merge in 88 % of the rounds)
                Vec<NCH> pre_rx2, pre_pf2, pre_rx3, pre_pf3;
                bool pre2 = false, pre3 = false;
                if constexpr (kCoop && kCoopPrefetch >= 1) {
                    if ((n & 7) == 7 && !(kL2)) {
                        pre_rx2 = bload<NCH>(arena + (int64_t)am.stk_rho(2) * L, lane);
                        pre_pf2 = bload<NCH>(arena + (int64_t)am.pf(usi(S.pf[2])) * L, lane);
                        pre2 = true;
                    }
                    if constexpr (kCoopPrefetch >= 2) {
                        if ((n & 15) == 15) {
                            pre_rx3 = bload<NCH>(arena + (int64_t)am.stk_rho(3) * L, lane);
                            pre_pf3 = bload<NCH>(arena + (int64_t)am.pf(usi(S.pf[3])) * L, lane);
                            pre3 = true;
                        }
                    }
                }
                if constexpr (Model::kSeparable)                                 // leapfrog, kinetic_energy.jl:126-163
                    leapfrog_step_regrad<NCH, !kConstRegs>(mdl, minv, eps_dir, q, p, lq, K);
                else
                    leapfrog_step_general<NCH>(mdl, minv, eps_dir, q, p, g, lq, K);
                const double pi = phase_logdensity(lq, K);
                STAMP(1);                                                        // leapfrog + reductions
                const double delta = pi - pi0;                                   // leaf, src/NUTS.jl:179
                i_n = i_start + sgn * (n + 1);
                cur_v = AccStat{delta < 0.0 ? delta : 0.0, 1};                   // :76-78
                if constexpr (kDefer) { if (lane == 0) dlog[nleaves - 1 + n] = delta; }
                if (uni(delta < s.min_delta)) {                                  // divergence :180
                    invalid = true;
                    term_left = i_n; term_right = i_n;                           // InvalidTree(i'), tree.jl:332
                    if constexpr (kDefer) {
                        stop_kind = 1; stop_n = n;
                    } else {
                        vres = cur_v;
                        for (int k = 0; k < depth; ++k)
                            if ((n >> k) & 1) vres = combine_acc(AccStat{S.lsa[k], usi(S.steps[k])}, vres);   // :347
                    }
                    break;
                }
                cur_omega = delta;
                cur_zeta = -1;            // the leaf in registers
                cur_pf = kPfLeaf;
                has_rho = false;
                int k = 0;
                // One merge at level k.  The level-0 merge is a separate instantiation (IS0) and is called outside the loop over
                // the higher levels: p_in is then dead before that loop starts -- as a `k == 0` case inside one loop it stayed live
                // through every level and the two-wavefront form spilled 160 B (3.0e8 instead of 3.6e8 leapfrog/s at depth 4).
                // Returns false when the sub-tree turned (the caller leaves the leaf loop).
                auto merge_level = [&](auto IS0, const int k) -> bool {
                    constexpr bool kIs0 = decltype(IS0)::value;
                    // The left sibling's vectors (level >= 1: from the L2-resident arena) are requested before
                    // the scalar bookkeeping so that the ~1k-cycle log-sum-exp runs under their latency.  With
                    // one wavefront per SIMD the 64 registers this holds across the call are free; at two per
                    // SIMD they spilled and the order cost 9 % (measured), hence the switch.
                    Vec<NCH> rx, pfx;     // rho and p#_first of the left sub-tree
                    auto fetch_left = [&]() {
                        if constexpr (kIs0) {
                            if constexpr (kPrevRegs) rx = p_in;
                            else rx = lds_load<NCH>(pprev);
                        } else {
                            if (kL1Rho && k == 1) {
                                rx = lds_load<NCH>(l1rho);
                                if constexpr (kL1Pf) pfx = lds_load<NCH>(l1pf);
                                else { pfx = bload<NCH>(arena + (int64_t)am.pf(usi(S.pf[1])) * L, lane); BYTES(3, 1); }
                            } else if (kL2 && k == 2) {
                                if constexpr (kL2Lds) rx = lds_load<NCH>(l2rho);
                                else rx = l2rho_r;
                                pfx = l2pf_r;
                            } else if (kCoop && kCoopPrefetch >= 1 && k == 2 && pre2) {
                                rx = pre_rx2; pfx = pre_pf2;
                            } else if (kCoop && kCoopPrefetch >= 2 && k == 3 && pre3) {
                                rx = pre_rx3; pfx = pre_pf3;
                            } else {
                                rx = bload<NCH>(arena + (int64_t)am.stk_rho(k) * L, lane);
                                pfx = bload<NCH>(arena + (int64_t)am.pf(usi(S.pf[k])) * L, lane);  BYTES(4, 2);
                            }
                        }
                    };
                    if constexpr (kNutsWaves == 4 || kCoop) fetch_left();
                    MergeScalars ms{0.0, 0.0};
                    AccStat vk{0.0, 0};
                    if constexpr (!kDefer) {
                        ms = merge_scalars<kNutsWaves == 4 && Model::kSeparable>(S.lsa[k], cur_v.lsa, S.omega[k], cur_omega);
                        vk = AccStat{ms.lsa, usi(S.steps[k]) + cur_v.steps};                         // tree.jl:347
                    }
                    if constexpr (kNutsWaves != 4 && !kCoop) fetch_left();
                    if constexpr (kIs0) {
                        rho = vadd<NCH>(rx, p);                                  // combine_turn_statistics, NUTS.jl:139-141
                        pfx = psharp<NCH>(minv, rx);
                    } else {
                        rho = vadd<NCH>(rx, rho);
                    }
                    has_rho = true;
                    double d_first, d_last;
#ifdef IDHMC_X2
                    d_first = 1.0 + rho.c[0].x * 1e-300; d_last = 1.0 + pfx.c[0].x * 1e-300;
#else
                    turn_dots<NCH>(rho, pfx, p, minv, d_first, d_last);          // is_turning, NUTS.jl:148-170
#endif
                    if (uni((d_first < 0.0) | (d_last < 0.0))) {                 // tree.jl:358
                        invalid = true;
                        term_left = i_start + sgn * (n - (2 << k) + 2);          // first node of this sub-tree
                        term_right = i_n;
                        if constexpr (kDefer) {
                            stop_kind = 2; stop_n = n; stop_k = k;
                        } else {
                            vres = vk;
                            for (int j = k + 1; j < depth; ++j)
                                if ((n >> j) & 1) vres = combine_acc(AccStat{S.lsa[j], usi(S.steps[j])}, vres);
                        }
                        return false;
                    }
                    if constexpr (!kDefer) {
                        // combine_proposals_and_logweights(is_doubling = false), tree.jl:238-245, :361-363
                        const double omega = ms.omega;
                        const double logprob2 = cur_omega - omega;               // biased_progressive_logprob2 :261-263
                        bool pick2 = uni(logprob2 >= 0.0);                       // rand_bool_logprob, NUTS.jl:32-34
                        if (!pick2) pick2 = uni(take_draw() > -logprob2);        // a draw is consumed only here
                        const int zk = usi(S.zeta[k]);
                        if (pick2) {
                            zfree |= 1u << zk;                                   // free_z!, NUTS.jl:43
                        } else {
                            if (cur_zeta >= 0) zfree |= 1u << cur_zeta;
                            cur_zeta = zk;
                        }
                        cur_omega = omega;
                        cur_v = vk;
                    }
                    if (cur_pf >= 0) pffree |= 1u << cur_pf;                     // free_rho#!, NUTS.jl:136-137
                    cur_pf = kIs0 ? (int)kPfLevel0 : usi(S.pf[k]);
                    if constexpr (kPrevRegs && kIs0) {
                        // M^-1 p_in is the p#_first of the two-leaf sub-tree: where that sub-tree parks at level 1 next (n = 1 mod 4)
                        // it goes to its place at once -- the LDS slot, or (forms without one) an arena slot -- instead of staying live
                        // until the park below (kept across the loop over the higher levels it cost the per-chain-metric form 160 B of scratch)
                        if (!((n >> 1) & 1) && n != nleaves - 1) {
                            if constexpr (kL1Pf) {
                                lds_store<NCH>(l1pf, pfx);
                            } else {
                                const int ps = __builtin_ctz(pffree);
                                pffree &= ~(1u << ps);
                                bstore<NCH>(arena + (int64_t)am.pf(ps) * L, lane, pfx);  BYTES(3, 1);
                                cur_pf = ps;
                            }
                        }
                    }
                    return true;
                };
                if (n & 1) {                                                     // complete pairs: level 0, then every further trailing 1 bit of n
                    bool ok = merge_level(BoolC<true>{}, 0);
                    k = 1;
                    while (ok && ((n >> k) & 1)) {
                        ok = merge_level(BoolC<false>{}, k);
                        if (ok) ++k;
                    }
                }
                STAMP(2);                                                        // merge cascade
                if (invalid) break;
                // materialise the leaf as a proposal candidate if it survived its merges (write-only until the end)
                if (!kDefer && cur_zeta < 0) {
                    const int zs = __builtin_ctz(zfree);
                    zfree &= ~(1u << zs);
                    if constexpr (kRegenerate) S.z_idx[zs] = i_n;
                    else { bstore<NCH>(arena + (int64_t)am.zq(zs) * L, lane, q); BYTES(7, 1); }
                    S.z_lq[zs] = lq;
                    S.z_pi[zs] = pi;
                    cur_zeta = zs;
                }
                if (n == nleaves - 1) break;                                     // the whole adjacent tree is in `cur`
                // park the sub-tree summary at level k until its right sibling is complete
                if (k == 0) {
                    // level 0: rho = p, p# = M^-1 p, both from p (kPrevRegs: p is the next leapfrog's input, nothing to store)
                    if constexpr (!kPrevRegs) lds_store<NCH>(pprev, p);
                } else {
                    // rho of the parked sub-tree: LDS for level 1 (a two-leaf sub-tree) where it fits, level 2 on chip in
                    // the register-rich form, else the arena
                    if (kL1Rho && k == 1) {
                        if (kTopLds && top_in_lds) {                             // the slot still holds the whole-tree rho
                            bstore<NCH>(arena + (int64_t)am.top_rho() * L, lane, lds_load<NCH>(l1rho));  BYTES(5, 1);
                            top_in_lds = false;
                        }
                        lds_store<NCH>(l1rho, rho);
                    } else if (kL2 && k == 2) {
                        if constexpr (kL2Lds) lds_store<NCH>(l2rho, rho);
                        else l2rho_r = rho;
                    } else {
                        bstore<NCH>(arena + (int64_t)am.stk_rho(k) * L, lane, rho);  BYTES(4, 1);
                    }
                    // its p#_first
                    if (kL1Pf && k == 1) {
                        // the first leaf is the level-0 summary just merged: M^-1 p_prev stays in LDS
                        if constexpr (!kPrevRegs) lds_store<NCH>(l1pf, psharp<NCH>(minv, lds_load<NCH>(pprev)));     // (kPrevRegs: stored in the merge)
                        S.pf[1] = kPfLevel1;
                    } else if (kL2 && k == 2) {
                        l2pf_r = lds_load<NCH>(l1pf);                            // the level-1 summary's, still in LDS
                        S.pf[2] = kPfLevel2;
                    } else {
                        if (cur_pf < kPfLeaf) {                                  // p#_first moves from LDS / registers to an arena slot
                            const int ps = __builtin_ctz(pffree);
                            pffree &= ~(1u << ps);
                            BYTES(cur_pf == kPfLevel0 ? 3 : 4, 1);
                            if (kL2 && cur_pf == kPfLevel2)
                                bstore<NCH>(arena + (int64_t)am.pf(ps) * L, lane, l2pf_r);
                            else
                                bstore<NCH>(arena + (int64_t)am.pf(ps) * L, lane,
                                            cur_pf == kPfLevel0 ? psharp<NCH>(minv, lds_load<NCH>(pprev)) : lds_load<NCH>(l1pf));
                            cur_pf = ps;
                        }
                        S.pf[k] = cur_pf;
                    }
                }
                if constexpr (!kDefer) {
                    S.omega[k] = cur_omega;
                    S.lsa[k] = cur_v.lsa;
                    S.steps[k] = cur_v.steps;
                    S.zeta[k] = cur_zeta;
                }
                STAMP(3);                                                        // candidate + park
            }

            if (invalid) {
                if constexpr (!kDefer) v = combine_acc(v, vres);                 // tree.jl:414, :417
                break;
            }
            // request the far edge's momentum (and the whole-tree rho where it lives in the arena) now; the scalar work
            // below covers the latency
            const int i_far = fwd ? i_minus : i_plus;
            const Vec<NCH> p_far = bload<NCH, kAuxFresh>(i_far ? arena + (int64_t)am.edge_p() * L : s.p + off, lane);  BYTES(5, 1);
            Vec<NCH> tr;
            if constexpr (kRich) tr = top_rho_r;
            else if (kTopLds && top_in_lds) tr = lds_load<NCH>(l1rho);
            else { tr = bload<NCH>(arena + (int64_t)am.top_rho() * L, lane); BYTES(5, 1); }
            if (fwd) i_plus = i_n; else i_minus = i_n;                           // :424-428
            if (cur_pf >= 0) pffree |= 1u << cur_pf;

            // combine_proposals_and_logweights(is_doubling = true), tree.jl:431-433
            if constexpr (!kDefer) {
                const MergeScalars mt = nuts_merge_scalars(v.lsa, cur_v.lsa, top_omega, cur_omega);
                v = AccStat{mt.lsa, v.steps + cur_v.steps};                      // tree.jl:414
                const double omega = mt.omega;
                const double logprob2 = cur_omega - top_omega;
                bool pick2 = uni(logprob2 >= 0.0);
                if (!pick2) pick2 = uni(take_draw() > -logprob2);
                if (pick2) {
                    if (top_zeta > 0) zfree |= 1u << top_zeta;
                    top_zeta = cur_zeta;
                } else {
                    zfree |= 1u << cur_zeta;
                }
                top_omega = omega;
            }
            depth += 1;                                                          // :434

            // whole-tree turn statistic and U-turn test, tree.jl:437-438
            {
                const Vec<NCH> trho = has_rho ? vadd<NCH>(tr, rho) : vadd<NCH>(tr, p);
                if constexpr (kRich) top_rho_r = trho;
                else if constexpr (kTopLds) { lds_store<NCH>(l1rho, trho); top_in_lds = true; }
                else { bstore<NCH>(arena + (int64_t)am.top_rho() * L, lane, trho); BYTES(5, 1); }
                double d_other, d_new;
                turn_dots_pp<NCH>(trho, p_far, p, minv, d_other, d_new);
                if (uni((d_other < 0.0) | (d_new < 0.0))) {
                    term_left = i_minus; term_right = i_plus;                    // InvalidTree(i-, i+)
                    break;
                }
            }
        }

        STAMP(4);                                                                // doubling bookkeeping
        // ---- epilogue: TreeStatisticsNUTS (src/NUTS.jl:262), next state, adaptation hooks ----------
        double lq_new = lq0, pi_new = pi0;                                       // the state the transition ends in (slot 0: the start)
        int iw_defer = 0;
        if constexpr (kDefer) {
            // the tree is built; now its bookkeeping (nuts_replay): acceptance statistic and the winner of the progressive sampling
            ReplayIn ri;
            ri.dl = dlog;
            ri.ndone = depth;                            // `depth` counts the doublings merged into the tree; a stopped one is not among them
            ri.stop_kind = stop_kind; ri.stop_n = stop_n; ri.stop_k = stop_k;
            ri.k0 = key.k0; ri.k1 = key.k1; ri.chain = key.chain; ri.iter = iter;
            __builtin_amdgcn_s_waitcnt(0);               // the leaves' Delta (stored by lane 0) are read back by every lane
#ifdef IDHMC_X4      // (cost attribution, results wrong on purpose) no bookkeeping at all: the last leaf of the last doubling wins
            ReplayOut ro; ro.lsa = 0.0; ro.steps = (1 << depth) - 1 + (stop_kind ? stop_n + 1 : 0); ro.win_d = depth > 0 ? depth - 1 : -1; ro.win_n = 0;
#else
            const ReplayOut ro = nuts_replay(ri, S);
#endif
            v = AccStat{ro.lsa, ro.steps};
            if (ro.win_d >= 0) {
                const int st = usi(S.z_idx[ro.win_d]);
                iw_defer = ((fwdmask >> ro.win_d) & 1u) ? st + (ro.win_n + 1) : st - (ro.win_n + 1);
                top_zeta = 1;
            }
        } else {
            lq_new = S.z_lq[top_zeta];
            pi_new = S.z_pi[top_zeta];
        }
        const double a_raw = nuts_dexp(v.lsa) / (double)v.steps;                      // acceptance_rate, NUTS.jl:84
        const double a = a_raw < 1.0 ? a_raw : 1.0;
        if (top_zeta > 0) {
            if constexpr (kRegenerate) {
                // walk to the winner along the same leapfrog chain the tree took: from the nearest checkpoint before it
                // on its side of the trajectory, else from the starting point
                const int iw = kDefer ? iw_defer : usi(S.z_idx[top_zeta]);
                int i_from = 0, d_from = -1;
                for (int d = kCheckpointDepth; d < depth + 1 && d < s.max_depth; ++d) {
                    if (!((ckmask >> d) & 1u)) continue;
                    const int cp = usi(S.ck_pos[d]);
                    const bool before = iw > 0 ? (cp > 0 && cp < iw) : (cp < 0 && cp > iw);
                    if (before && (cp > 0 ? cp : -cp) > (i_from > 0 ? i_from : -i_from)) { i_from = cp; d_from = d; }
                }
                const double eps_w = iw > 0 ? eps : -eps;
                const int nw = (iw > 0 ? iw : -iw) - (i_from > 0 ? i_from : -i_from);
                q = bload<NCH, kAuxFresh>(d_from >= 0 ? arena + (int64_t)am.ck_q(d_from) * L : s.q + off, lane);
                p = bload<NCH, kAuxFresh>(d_from >= 0 ? arena + (int64_t)am.ck_p(d_from) * L : s.p + off, lane);  BYTES(2, 2);
                if constexpr (!Model::kSeparable) { g = bload<NCH, kAuxFresh>(s.g + off, lane); BYTES(2, 1); }
                double lqw, Kw;
                for (int t = 0; t < nw - (kDefer ? 1 : 0); ++t) {
                    if constexpr (Model::kSeparable) leapfrog_step_regrad<NCH, !kConstRegs>(mdl, minv, eps_w, q, p, lqw, Kw);
                    else leapfrog_step_general<NCH>(mdl, minv, eps_w, q, p, g, lqw, Kw);
                }
                if constexpr (kDefer) {
                    // the last step is the winning leaf itself: its l(q) and pi are the leaf's, bit for bit (the steps before it
                    // drop their reductions: nothing reads them)
                    leapfrog_step_regrad<NCH, !kConstRegs>(mdl, minv, eps_w, q, p, lqw, Kw);
                    lq_new = lqw;
                    pi_new = phase_logdensity(lqw, Kw);
                }
                // (separable densities do not write grad l back: the gradient of a separable density is re-derived from q wherever it
                // is needed -- this kernel never reads the array -- and the host marks it stale, idhmc_api.hip ensure_grad)
            } else {
                q = bload<NCH>(arena + (int64_t)am.zq(top_zeta) * L, lane);  BYTES(7, 1);
                // the proposal's gradient, same bits as when it was a leaf
                if constexpr (Model::kSeparable) (void)eval_density<NCH>(mdl, q, g);
                else (void)mdl.grad(q, g);
            }
            bstore<NCH, kNt>(s.q + off, lane, q);  BYTES(6, 1);
            if constexpr (!(kRegenerate && Model::kSeparable)) { bstore<NCH, kNt>(s.g + off, lane, g); BYTES(6, 1); }
        } else if ((flags & (IDHMC_T_ACCUM_METRIC | IDHMC_T_ACCUM_MOMENTS)) || s.fz_q) {
            q = bload<NCH, kAuxFresh>(s.q + off, lane);  BYTES(6, 1);
        }
        if (s.fz_q) {
            // the draw leaves for the host from here: row (transition, chain) of the launch's staging block, D doubles, unpadded
            double *row = s.fz_q + ((int64_t)it * s.C + c) * s.D;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int e = 128 * j + 2 * lane;
                if (e < s.D) row[e] = q.c[j].x;
                if (e + 1 < s.D) row[e + 1] = q.c[j].y;
            }
        }
        if (lane == 0) {
            if (top_zeta > 0) s.lq[c] = lq_new;
            s.pi[c] = pi_new;
            idhmc_tree_stats st;
            st.pi = pi_new;
            st.acceptance_rate = a;
            st.term_left = term_left; st.term_right = term_right;
            st.depth = depth; st.steps = v.steps;
            s.stats[c] = st;
            if (s.fz_st) s.fz_st[(int64_t)it * s.C + c] = st;
            atomicAdd(&wg_acc[39], (unsigned long long)v.steps);
        }
        if ((flags & IDHMC_T_ADAPT_EPS) && s.eps_mode == IDHMC_EPS_PER_CHAIN) {
            // adapt_stepsize, src/stepsize.jl:220-229, then current_eps (:235) for the next transition
            const double mu = s.da.mu[c];
            const double m = (double)(ld_fresh(s.da.m + c) + 1);
            double Hbar = ld_fresh(s.da.Hbar + c), lb = ld_fresh(s.da.logeps_bar + c);
            Hbar += (s.da_delta - a - Hbar) / (m + (double)s.da_t0);
            const double le = mu - __builtin_sqrt(m) / s.da_gamma * Hbar;
            lb += nuts_dexp(-s.da_kappa * nuts_dlog(m)) * (le - lb);
            const double e = nuts_dexp(le);
            if (lane == 0) {
                s.da.m[c] = (int64_t)m;
                s.da.Hbar[c] = Hbar;
                s.da.logeps[c] = le;
                s.da.logeps_bar[c] = lb;
                s.eps[c] = e;
                if (e < 1e-10) {                                                // src/warmup.jl:291-296
                    s.status[c] = IDHMC_ERR_EPS_UNDERFLOW;
                    atomicMax(s.total_steps + 1, (unsigned long long)IDHMC_ERR_EPS_UNDERFLOW);   // the host's pulse
                }
            }
        }
        if (flags & IDHMC_T_ACCUM_METRIC) {
            // running form of the block body of GaussianKineticEnergy!, src/hamiltonian.jl:86-93
            const int nwin = ld_fresh(s.mw_n + c);
            if (nwin == 0) {
                bstore<NCH>(s.mw_x1 + off, lane, q);
                bstore<NCH>(s.mw_s1 + off, lane, vfill<NCH>(0.0));
                bstore<NCH>(s.mw_s2 + off, lane, vfill<NCH>(0.0));
            } else {
                const Vec<NCH> x1 = bload<NCH, kAuxFresh>(s.mw_x1 + off, lane);
                Vec<NCH> s1 = bload<NCH, kAuxFresh>(s.mw_s1 + off, lane);
                Vec<NCH> s2 = bload<NCH, kAuxFresh>(s.mw_s2 + off, lane);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const double dx = q.c[j].x - x1.c[j].x, dy = q.c[j].y - x1.c[j].y;
                    s1.c[j].x = dx + s1.c[j].x; s1.c[j].y = dy + s1.c[j].y;
                    s2.c[j].x = dfma(dx, dx, s2.c[j].x); s2.c[j].y = dfma(dy, dy, s2.c[j].y);
                }
                bstore<NCH>(s.mw_s1 + off, lane, s1);
                bstore<NCH>(s.mw_s2 + off, lane, s2);
            }
            if (lane == 0) s.mw_n[c] = nwin + 1;
        }
        if (flags & IDHMC_T_ACCUM_MOMENTS) {
            const int64_t nm = ld_fresh(s.mom_n + c) + 1;
            const double inv = 1.0 / (double)nm;
            Vec<NCH> mean = bload<NCH, kAuxFresh>(s.mom_mean + off, lane);
            Vec<NCH> m2 = bload<NCH, kAuxFresh>(s.mom_m2 + off, lane);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const double dx = q.c[j].x - mean.c[j].x, dy = q.c[j].y - mean.c[j].y;
                mean.c[j].x = dfma(dx, inv, mean.c[j].x); mean.c[j].y = dfma(dy, inv, mean.c[j].y);
                m2.c[j].x = dfma(dx, q.c[j].x - mean.c[j].x, m2.c[j].x);
                m2.c[j].y = dfma(dy, q.c[j].y - mean.c[j].y, m2.c[j].y);
            }
            bstore<NCH>(s.mom_mean + off, lane, mean);
            bstore<NCH>(s.mom_m2 + off, lane, m2);
            if (lane == 0) s.mom_n[c] = nm;
        }
        if ((flags & IDHMC_T_ACCUM_DIAG) && lane == 0) {
            // reference diagnostics reduced as the records are produced (src/diagnostics.jl:28-32, 61-101)
            const int nd = ld_fresh(s.diag.n + c);
            if (nd == 0) {
                s.diag.pi1[c] = pi_new; s.diag.s1[c] = 0.0; s.diag.s2[c] = 0.0; s.diag.d2[c] = 0.0;
            } else {
                const double dl = pi_new - ld_fresh(s.diag.pi1 + c), dp = pi_new - ld_fresh(s.diag.prev + c);
                s.diag.s1[c] = ld_fresh(s.diag.s1 + c) + dl;
                s.diag.s2[c] = dfma(dl, dl, ld_fresh(s.diag.s2 + c));
                s.diag.d2[c] = dfma(dp, dp, ld_fresh(s.diag.d2 + c));
            }
            s.diag.prev[c] = pi_new;
            s.diag.n[c] = nd + 1;
            unsigned long long *cn = s.diag.counters;
            long long hi, lo;
            xchg_limbs(IDHMC_XCHG_ACCEPT, a, hi, lo);
            atomicAdd(&wg_acc[0], 1ull);
            atomicAdd(&wg_acc[1], (unsigned long long)hi);
            atomicAdd(&wg_acc[2], (unsigned long long)lo);
            const int cls = (term_left == 1 && term_right == 0) ? 0 : (term_left == term_right ? 1 : 2);   // src/tree.jl:285,300
            atomicAdd(&wg_acc[3 + cls], 1ull);
            atomicAdd(&wg_acc[6 + (depth < 32 ? depth : 32)], 1ull);
            int bin = (int)(a * (double)IDHMC_DIAG_ACC_BINS);
            bin = bin < 0 ? 0 : (bin > IDHMC_DIAG_ACC_BINS - 1 ? IDHMC_DIAG_ACC_BINS - 1 : bin);
            atomicAdd(cn + 39 + bin, 1ull);          // (1024 bins: spread addresses, straight to memory)
        }
        if (n_iter > 1u) {
            // this transition of the chain is in the XCD's L2 for whoever runs the next: all stores acknowledged, then the count
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (lane == 0) __hip_atomic_store(s.iters_done + cu, it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        STAMP(5);                                                                // epilogue
        STAMP_FLUSH;
        BYTES_FLUSH;
        if constexpr (kCoop && !kCoopRefill) {
            mdl.retire();
            mdl.serve();
        }
    }
    // the workgroup's sums -> the global counters (every wavefront gets here: the queue is empty for all of them in the end)
    __syncthreads();
    if (threadIdx.x < kWgAcc) {
        const unsigned long long x = wg_acc[threadIdx.x];
        if (x) {
            if (threadIdx.x == 39) atomicAdd(s.total_steps, x);
            else if (s.diag.counters) atomicAdd(s.diag.counters + threadIdx.x, x);
        }
    }
}

// ---- find_initial_stepsize (src/stepsize.jl:111-126,150-164) per chain ------------------------------
// A(eps) = exp(logdensity(H, leapfrog(z, eps)) - logdensity(H, z)); only scalars leave the registers.
template <int NCH, class Model>
IDHMC_DEV double local_ratio(const Model &mdl, const Vec<NCH> &minv, const Vec<NCH> &q, const Vec<NCH> &p,
                             const Vec<NCH> &g, double eps, double target)
{
    Vec<NCH> q1 = q, p1 = p, g1 = g;
    double lq, K;
    leapfrog_step<NCH>(mdl, minv, eps, q1, p1, g1, lq, K);
    return dexp(phase_logdensity(lq, K) - target);
}

template <int NCH, class Model>
__global__ __launch_bounds__(256) void k_stepsize_search(DevState s)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Model mdl;
    mdl.load(s.mu, s.tau, lane);
    for (int64_t c = wave; c < s.C; c += nw) {
        const int64_t off = c * s.L;
        const Vec<NCH> q = vload<NCH>(s.q + off, lane);
        const Vec<NCH> p = vload<NCH>(s.p + off, lane);
        const Vec<NCH> g = vload<NCH>(s.g + off, lane);
        const Vec<NCH> minv = vload<NCH>(s.minv + c * s.minv_stride, lane);
        const double target = phase_logdensity(s.lq[c], kinetic_energy<NCH>(minv, p));   // :151
        int rc = 0;
        double e0 = s.ss_eps0, result = s.ss_eps0;
        if (!dfinite(target)) {
            rc = IDHMC_ERR_NONFINITE_START;                                              // :152-153
        } else {
            double A0 = local_ratio<NCH>(mdl, minv, q, p, g, e0, target);                // :113
            if (!(s.ss_a_min <= A0 && A0 <= s.ss_a_max)) {                               // :114
                // find_crossing_stepsize :51-72
                const double sg = A0 > s.ss_a_max ? 1.0 : -1.0;
                const double a = A0 > s.ss_a_max ? s.ss_a_max : s.ss_a_min;
                const double Cf = sg < 0.0 ? 1.0 / s.ss_C : s.ss_C;
                double e1 = e0, A1 = A0;
                bool found = false;
                for (int it = 0; it < s.ss_maxiter_crossing; ++it) {
                    const double e = e0 * Cf;
                    const double Ae = local_ratio<NCH>(mdl, minv, q, p, g, e, target);
                    if (sg * (Ae - a) <= 0.0) { e1 = e; A1 = Ae; found = true; break; }
                    e0 = e; A0 = Ae;
                }
                if (!found) {
                    rc = IDHMC_ERR_STEPSIZE_SEARCH;                                      // :71
                } else if (s.ss_a_min <= A1 && A1 <= s.ss_a_max) {
                    result = e1;                                                         // :118
                } else {
                    double lo = e0, hi = e1;                                             // :120-124
                    if (!(e0 < e1)) { lo = e1; hi = e0; }
                    found = false;
                    for (int it = 0; it < s.ss_maxiter_bisect; ++it) {                   // bisect_stepsize :83-102
                        const double em = 0.5 * (lo + hi);
                        const double Am = local_ratio<NCH>(mdl, minv, q, p, g, em, target);
                        if (s.ss_a_min <= Am && Am <= s.ss_a_max) { result = em; found = true; break; }
                        else if (Am < s.ss_a_min) hi = em;
                        else lo = em;
                    }
                    if (!found) rc = IDHMC_ERR_STEPSIZE_SEARCH;                          // :101
                }
            }
        }
        if (lane == 0) {
            s.eps[c] = result;
            if (rc) s.status[c] = rc;
        }
    }
}

}  // namespace idhmc
