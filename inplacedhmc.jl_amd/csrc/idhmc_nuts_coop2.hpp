// idhmc_nuts_coop2.hpp -- k_nuts_coop2: the NUTS transition for the dense multivariate normal at L <= 256 with a shared metric,
// TWO chains per wavefront, so that one chain's tree work runs while the matrix cores work on the other's gradient.
//
// Why.  In k_nuts<DenseMvnCoop> (idhmc_nuts_kernel.hpp, idhmc_device.hpp) a CU holds one workgroup of 16 wavefronts = 16 chains, and a
// gradient is a service round of that workgroup: ~8.4 us on the matrix cores in which the chains wait, then ~6.9 us of per-chain tree
// work in which the matrix cores idle.  Nothing overlaps because every resident chain is in the same phase.  Here a
// wavefront owns two chains ("contexts" 0 and 1) and a workgroup two 16-row tiles; time is cut into PHASES by one workgroup barrier each:
//
//   phase n:    every wavefront runs the tree work of its context X from "T_X is there" (second half of the leapfrog, the leaf, its
//               merges, possibly the end of the transition and the start of the next chain) to "q'_X is in its tile row" (first half
//               of the next leapfrog), and the workgroup multiplies tile Y:  T_Y = (q'_Y - mu) P, 16 columns per wavefront.
//   phase n+1:  the same with X and Y exchanged.
//
// Half of the wavefronts of a SIMD do their share of the multiply at the start of a phase and half at its end, so the matrix cores
// and the vector ALU of a SIMD are both busy throughout.  There is ONE place in the program where a wavefront changes from one chain
// to the other (the "yield": between the two halves of a leapfrog), and both contexts are always parked exactly there, so changing
// context is exchanging the values of the variables that are live at that point -- struct Ctx below -- and going on: the loop nest of
// the transition stays a loop nest.  A context without a chain (queue empty: `dead`; not started yet: `fresh`) takes the same path
// with its work switched off, so every wavefront passes the yield -- its only barrier -- once per phase until the workgroup's last
// chain is done.
//
// What a context keeps where: q and grad l = -T live in the context's rows of the q tile and the T tile (LDS) -- the tiles are the
// operands of the multiply anyway --, the momentum and the previous leaf's momentum in registers (16 VGPRs per context at L = 256),
// the per-level scalars in LDS, sub-tree summaries, trajectory edges and proposal candidates in the context's arena slot (L2).
// Arithmetic, order of operations and random numbers are those of k_nuts: results are bit-identical (tests/test_gpu_dense.py).
// Reference: src/tree.jl:321-444, src/NUTS.jl:32-191,251-264 as cited line by line in idhmc_nuts_kernel.hpp.
#pragma once
#include "idhmc_nuts_kernel.hpp"

namespace idhmc {

#ifndef IDHMC_COOP2_PD
#define IDHMC_COOP2_PD 8            // k-blocks of P requested ahead in the multiply (two per 16-byte load)
#endif

struct Coop2Scalars {               // per-context scalars of the live sub-tree summaries and of the proposal candidates (LDS)
    double omega[kMaxDepth];
    double lsa[kMaxDepth];
    int steps[kMaxDepth];
    int zeta[kMaxDepth];
    int pf[kMaxDepth];
    double z_lq[kMaxDepth + 4];
    double z_pi[kMaxDepth + 4];
};

template <int NCH>
struct Coop2Shape {
    static constexpr int L = 128 * NCH, DS = L + 2, KB = L / 4, kTile = 16 * DS, kColTiles = L / 16, kWaves = 16;
    // dynamic LDS (doubles): [mu L][M^-1 L][q tile 0][q tile 1][T tile 0][T tile 1]
    __host__ __device__ static constexpr size_t lds_doubles() { return (size_t)2 * L + (size_t)4 * kTile; }
};
// arena vectors of one context: those of the candidate-storing form of k_nuts plus the candidates' gradients (the winner's gradient
// is stored with it: re-deriving it would be a second place to wait for the matrix cores)
__host__ __device__ constexpr int coop2_arena_vectors(int max_depth) { return ArenaMap{max_depth, false, 0}.count() + max_depth + 2; }

template <int NCH>
struct Coop2Ctx {
    Vec<NCH> p;          // momentum; between the two halves of a leapfrog: the half-kicked momentum
    Vec<NCH> pin;        // the momentum the current leapfrog started from = the level-0 summary an odd leaf merges with
    double ebatch;       // 64 exponential draws, one per lane
    double eps, lq0, pi0, top_omega, v_lsa;
    uint32_t cu, dirs, draw, ebase, zfree, pffree;
    int v_steps, top_zeta, regs_edge, i_minus, i_plus, depth, term_left, term_right, fwd, i_start, n;
    int id;              // 0 / 1: which tiles, scalars and arena slot are this context's
    bool dead, fresh;
};

template <int NCH>
__global__ __launch_bounds__(1024, 1)
void k_nuts_coop2(DevState s, uint32_t iter, uint32_t flags)
{
    typedef Coop2Shape<NCH> Sh;
    constexpr int L = Sh::L, DS = Sh::DS, KB = Sh::KB, kTile = Sh::kTile, kPD = IDHMC_COOP2_PD;
    static_assert((KB & (KB - 1)) == 0 && KB % kPD == 0, "k-block count: power of two, multiple of the prefetch depth");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ Coop2Scalars Sall[16][2];
    __shared__ int alive[2];              // contexts of the workgroup that still have or may take a chain, one copy per phase parity
    constexpr int kWgAcc = 40;            // [0..38] diagnostics counters, [39] leapfrog steps (as in k_nuts)
    __shared__ unsigned long long wg_acc[kWgAcc];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const ArenaMap am{s.max_depth, false, 0};
    const int zg0 = am.count();           // first vector of the candidates' gradients
    double *const lmu = lds, *const lminv = lds + L, *const qtile = lds + 2 * L, *const ttile = qtile + 2 * kTile;
    for (int i = threadIdx.x; i < L; i += 1024) { lmu[i] = s.mu[i]; lminv[i] = s.minv[i]; }
    LdsVec minv, muv;
    minv.p = reinterpret_cast<const double2 *>(lminv) + lane;
    muv.p = reinterpret_cast<const double2 *>(lmu) + lane;
    if (threadIdx.x < 2) alive[threadIdx.x] = 32;
    if (threadIdx.x < kWgAcc) wg_acc[threadIdx.x] = 0ull;
    __syncthreads();

    // Which quarter of a phase a wavefront does its share of the multiply in: the four wavefronts of a SIMD (w, w + 4, w + 8, w + 12) take
    // different ones, so that a SIMD's matrix pipe has one customer at a time while the other three run tree code on its vector ALU.
#ifdef IDHMC_STAMPS      // diagnostic build: cycles in [2] the multiply, [3] the barrier, [4] everything else, [5] phases
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();   // [4..7] inside the multiply: drain, first operands, k loop, T
#define C2STAMP(i) do { const long long t_ = clock64(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define C2STAMP(i)
#endif
#ifndef IDHMC_COOP2_SLOTS
#define IDHMC_COOP2_SLOTS 4
#endif
    const int slot = IDHMC_COOP2_SLOTS == 4 ? ((wv >> 2) & 3) : (((wv >> 2) & 1) ? 3 : 0);
    bool mult_done = false;               // this phase's share is done
    // T_t = (Q_t - mu) P for tile t: this wavefront's 16 columns, v_mfma_f64_16x16x4_f64, k ascending (the engine's summation order);
    // lane (kk, jj) supplies A[row jj][k = 4 kb + kk] and B[k = 4 kb + kk][column 16 wv + jj].  B comes from the PACKED copy of P
    // (DevState::prec_pack, made by idhmc_create): the lane's elements of k-blocks 2 m and 2 m + 1 are the two doubles of one 16-byte
    // load, 1 KiB contiguous per wave-level load.  (Read from P itself a lane's two k-blocks are 8 bytes each, 32 bytes apart: every
    // wave-level load occupies the CU's address path for 16 cycles whatever its width, and 16 wavefronts x 64 8-byte loads are 7.8 us
    // per phase -- measured 8.9 -- for which the matrix cores wait.)
    auto multiply = [&](const int t) {
        mult_done = true;
        if (wv >= Sh::kColTiles) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));      // (addresses recomputed here: hoisted out of the transition they would be spilled)
        const int kk = ln >> 4, jj = ln & 15;
        const __amdgpu_buffer_rsrc_t rP = buf_rsrc(s.prec_pack + (size_t)wv * (L / 8) * 128);
        const int vo = ln * 16;
        constexpr int kPairs = KB / 2, kPD2 = kPD / 2;
        __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the loop then waits for exactly the block it needs
        C2STAMP(4);
        v2d bq[kPD2];
#pragma unroll
        for (int u = 0; u < kPD2; ++u) bq[u] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rP, vo, 1024 * u, 0));
        const double *ap = qtile + t * kTile + jj * DS + kk;
        const double *mp = lmu + kk;
        v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
#ifdef IDHMC_STAMPS
        __builtin_amdgcn_s_waitcnt(0x0F70);
        C2STAMP(5);
#endif
        // The A operands of pair m + 1 are read from LDS BEFORE the two matrix instructions of pair m are issued (sched_barrier pins
        // that): the accumulator chain is serial (64 dependent instructions of 64 cycles), and with the reads issued after them their
        // latency sat on that chain -- 127 ticks per instruction instead of 64 (stamps).
        double a0 = ap[0] - mp[0], a1 = ap[4] - mp[4];
#pragma unroll 1
        for (int m0 = 0; m0 < kPairs; m0 += kPD2) {
#pragma unroll
            for (int u = 0; u < kPD2; ++u) {
                const int m = m0 + u, mn = (m + 1) & (kPairs - 1);
                const double qn0 = ap[8 * mn], qn1 = ap[8 * mn + 4], mn0 = mp[8 * mn], mn1 = mp[8 * mn + 4];
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bq[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bq[u].y, acc, 0, 0, 0);
                // unconditional (the last trips wrap around and are discarded): a branch here makes the compiler drain all loads
                bq[u] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rP, vo, 1024 * ((m + kPD2) & (kPairs - 1)), 0));
                a0 = qn0 - mn0;
                a1 = qn1 - mn1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef IDHMC_STAMPS
        asm volatile("" :: "v"(acc));
        C2STAMP(6);
#endif
        double *tp = ttile + t * kTile + kk * DS + 16 * wv + jj;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tp[4 * reg * DS] = acc[reg];
        C2STAMP(7);
    };

    Coop2Ctx<NCH> cx, ox;
    cx.id = 0; ox.id = 1;
    cx.dead = false; ox.dead = false;
    cx.fresh = false; ox.fresh = true;
    cx.depth = 0; ox.depth = 0; cx.n = 0; ox.n = 0; cx.fwd = 1; ox.fwd = 1; cx.eps = 0.0; ox.eps = 0.0;
    cx.p = vfill<NCH>(0.0); ox.p = cx.p; cx.pin = cx.p; ox.pin = cx.p;
    int ph = 0, owed = 0;                 // phase parity; decrements this wavefront still owes the other parity's alive count
    bool all_done = false;

    auto Sc = [&]() -> Coop2Scalars & { return Sall[wv][cx.id]; };
    auto arena = [&]() -> double * { return s.arena + (((int64_t)blockIdx.x * 16 + wv) * 2 + cx.id) * s.arena_stride; };
    auto qrow = [&]() -> double2 * { return reinterpret_cast<double2 *>(qtile + cx.id * kTile + wv * DS) + lane; };
    auto trow = [&]() -> double2 * { return reinterpret_cast<double2 *>(ttile + cx.id * kTile + wv * DS) + lane; };
    auto vneg = [&](const Vec<NCH> &a) { Vec<NCH> r;
#pragma unroll
        for (int j = 0; j < NCH; ++j) r.c[j] = make_double2(-a.c[j].x, -a.c[j].y);
        return r; };

    for (;;) {
        // ---- take a chain (src/mcmc.jl:150-157: chains are independent; the queue hands them out) ----------------------------
        bool have = false;
        if (!cx.dead) {
            uint32_t cu = 0;
            if (lane == 0) cu = atomicAdd(s.queue, 1u);
            cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)cu);
            if ((int64_t)cu >= s.C) {
                cx.dead = true;           // this context takes no further part; the wavefront keeps multiplying until all are
                if (lane == 0) atomicSub(&alive[ph], 1);
                ++owed;
            } else {
                cx.cu = cu;
                have = true;
            }
        }
        if (have) {
            // ---- sample_tree prologue (src/NUTS.jl:251-260) ------------------------------------------------------------------
            const int64_t c = (int64_t)cx.cu, off = c * L;
            const RngKey key{s.k0, s.k1, s.first_chain + cx.cu};
            double2 *const scratch = trow();          // (the T row is this context's to use: its tile is not being multiplied now)
            if (flags & IDHMC_T_KEEP_P) {
                cx.p = bload<NCH>(s.p + off, lane);
            } else {
                // rand_p! (:254), as in k_nuts: one 128-element chunk per trip through a rolled loop, W staged in LDS first
                lds_store<NCH>(scratch, bload<NCH>(s.w, lane));
#pragma unroll 1
                for (int j = 0; j < NCH; ++j) {
                    const int pair = j * 64 + lane;
                    const NormalPair nn = nuts_randn_pair(key.k0, key.k1, key.chain, iter, (uint32_t)pair);
                    const double2 wj = scratch[j * 64];
                    scratch[j * 64] = make_double2((2 * pair < s.D) ? wj.x * nn.a : 0.0, (2 * pair + 1 < s.D) ? wj.y * nn.b : 0.0);
                }
                cx.p = lds_load<NCH>(scratch);
                bstore<NCH>(s.p + off, lane, cx.p);   // p0 stays in the state array: the starting point doubles as the far edge
            }
            lds_store<NCH>(qrow(), bload<NCH, kNt>(s.q + off, lane));
            lds_store<NCH>(trow(), vneg(bload<NCH, kNt>(s.g + off, lane)));
            cx.dirs = (flags & IDHMC_T_USE_DIRECTIONS) ? s.directions[c] : rand_directions(key, iter);   // :252
            cx.dirs = (uint32_t)usi((int)cx.dirs);
            cx.eps = s.eps[c];
            cx.lq0 = s.lq[c];
            cx.pi0 = phase_logdensity(cx.lq0, kinetic_energy<NCH>(minv, cx.p));                       // :260
            cx.draw = 0; cx.ebase = 0;
            cx.ebatch = nuts_randexp_batch(key.k0, key.k1, key.chain, iter, 0u);
            // ---- sample_trajectory initial leaf (src/tree.jl:388-393) ---------------------------------------------------------
            bstore<NCH>(arena() + (int64_t)am.top_rho() * L, lane, cx.p);
            cx.top_zeta = 0; cx.top_omega = 0.0;
            cx.v_lsa = -kInf; cx.v_steps = 0;
            Sc().z_lq[0] = cx.lq0;
            Sc().z_pi[0] = cx.pi0;
            cx.zfree = ((1u << (s.max_depth + 2)) - 1u) << 1;
            cx.pffree = (1u << (s.max_depth + 1)) - 1u;
            cx.regs_edge = 1;
            cx.i_minus = 0; cx.i_plus = 0; cx.depth = 0;
            cx.term_left = 1; cx.term_right = 0;                                                     // REACHED_MAX_DEPTH, src/tree.jl:300
        }
        auto take_draw = [&]() -> double {
            if (cx.draw >= cx.ebase + 64u) {
                cx.ebase += 64u;
                cx.ebatch = nuts_randexp_batch(s.k0, s.k1, s.first_chain + cx.cu, iter, cx.ebase);
            }
            const double e = read_lane(cx.ebatch, usi((int)(cx.draw - cx.ebase)));
            ++cx.draw;
            return e;
        };

        bool leave = false;               // a context that had no chain yet goes to take one
        while (cx.dead || cx.depth < s.max_depth) {                                                   // src/tree.jl:395
            if (!cx.dead) {
                const int64_t off = (int64_t)cx.cu * L;
                const int fwd = (int)(cx.dirs & 1u);                                                  // next_direction :152-155
                cx.dirs >>= 1;
                if (fwd != cx.regs_edge) {                                                            // continue from the other edge (:398-404)
                    const int i_regs = cx.regs_edge ? cx.i_plus : cx.i_minus, i_other = cx.regs_edge ? cx.i_minus : cx.i_plus;
                    if (i_regs != 0 || i_other != 0) {
                        double *const ar = arena();
                        const Vec<NCH> op = bload<NCH>(i_other ? ar + (int64_t)am.edge_p() * L : s.p + off, lane);
                        const Vec<NCH> oq = bload<NCH>(i_other ? ar + (int64_t)am.edge_q() * L : s.q + off, lane);
                        const Vec<NCH> og = bload<NCH>(i_other ? ar + (int64_t)am.edge_g() * L : s.g + off, lane);
                        if (i_regs != 0) {
                            bstore<NCH>(ar + (int64_t)am.edge_p() * L, lane, cx.p);
                            bstore<NCH>(ar + (int64_t)am.edge_q() * L, lane, lds_load<NCH>(qrow()));
                            bstore<NCH>(ar + (int64_t)am.edge_g() * L, lane, vneg(lds_load<NCH>(trow())));
                        }
                        cx.p = op;
                        lds_store<NCH>(qrow(), oq);
                        lds_store<NCH>(trow(), vneg(og));
                    }
                    cx.regs_edge = fwd;
                }
                cx.fwd = fwd;
                cx.i_start = fwd ? cx.i_plus : cx.i_minus;
                cx.n = 0;
            }

            // ---- adjacent_tree(depth), src/tree.jl:321-366, as a flat loop over its leaves -----------------------------------
            bool invalid = false;
            AccStat vres{-kInf, 0};
            Vec<NCH> rho;
            bool has_rho = false;
            int cur_zeta = -1, cur_pf = kPfLeaf, i_n = 0;
            double cur_omega = 0.0;
            AccStat cur_v{-kInf, 0};
            for (;;) {
                if (!cx.dead) {
                    // leapfrog loop A (src/kinetic_energy.jl:144-150): p_m = p + eps/2 grad, q' = q + eps M^-1 p_m; q' goes to the tile
                    cx.pin = cx.p;
                    const double eps_dir = cx.fwd ? cx.eps : -cx.eps;                                  // move, src/NUTS.jl:18-21
                    const double eh = 0.5 * eps_dir;
                    double2 *const qr = qrow();
                    const double2 *const tr = trow();
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const double2 mv = minv.get(j), t = tr[j * 64], q = qr[j * 64];
                        cx.p.c[j].x = dfma(eh, -t.x, cx.p.c[j].x);
                        cx.p.c[j].y = dfma(eh, -t.y, cx.p.c[j].y);
                        qr[j * 64] = make_double2(dfma(eps_dir * mv.x, cx.p.c[j].x, q.x), dfma(eps_dir * mv.y, cx.p.c[j].y, q.y));
                    }
                }
                // ================= yield: the only place where a wavefront changes chains, and its only barrier ==================
                C2STAMP(2);
                if (!mult_done) multiply(ox.id);      // this phase's tile: the last quarter, and whoever did not pass its own place
                C2STAMP(0);
                { const Coop2Ctx<NCH> t = cx; cx = ox; ox = t; }
                __syncthreads();                      // q' rows of the old context and T of the new one are complete
                C2STAMP(1);
                ph ^= 1;
                if (usi(*reinterpret_cast<volatile int *>(&alive[ph ^ 1])) == 0) { all_done = true; break; }
                if (owed) { if (lane == 0) atomicSub(&alive[ph], owed); owed = 0; }
                mult_done = false;
                if (slot == 0) multiply(ox.id);       // the new phase's tile: first quarter
                C2STAMP(0);
#ifdef IDHMC_STAMPS
                st_acc[3] += 1;
#endif
                if (cx.dead) continue;
                if (cx.fresh) { cx.fresh = false; leave = true; break; }
                // ================= resume: T = (q' - mu) P of this context is in its T row ========================================
                const int n = cx.n, depth = cx.depth, nleaves = 1 << depth, sgn = cx.fwd ? 1 : -1, i_start = cx.i_start;
                Coop2Scalars &S = Sc();
                double *const ar = arena();
                double lq, K;
                {
                    // evaluate_l! and leapfrog loop B (src/kinetic_energy.jl:72-85, 152-161)
                    const double eh = 0.5 * (cx.fwd ? cx.eps : -cx.eps);
                    const double2 *const qr = qrow();
                    const double2 *const tr = trow();
                    double l0 = 0.0, l1 = 0.0, k0 = 0.0, k1 = 0.0;
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const double2 mv = minv.get(j), m = muv.get(j), t = tr[j * 64], q = qr[j * 64];
                        const double dx = q.x - m.x, dy = q.y - m.y;
                        l0 = dfma(t.x, dx, l0);
                        l1 = dfma(t.y, dy, l1);
                        cx.p.c[j].x = dfma(eh, -t.x, cx.p.c[j].x);
                        cx.p.c[j].y = dfma(eh, -t.y, cx.p.c[j].y);
                        k0 = dfma(cx.p.c[j].x * mv.x, cx.p.c[j].x, k0);
                        k1 = dfma(cx.p.c[j].y * mv.y, cx.p.c[j].y, k1);
                    }
                    double sl, sk;
                    wave_sum2(l0, l1, k0, k1, sl, sk);
                    lq = -0.5 * sl;
                    lq = dfinite(lq) ? lq : -kInf;
                    K = 0.5 * sk;
                }
                C2STAMP(2);
                if (slot == 1) multiply(ox.id);       // second quarter
                C2STAMP(0);
                const double pi = phase_logdensity(lq, K);
                const double delta = pi - cx.pi0;                                                     // leaf, src/NUTS.jl:179
                i_n = i_start + sgn * (n + 1);
                cur_v = AccStat{delta < 0.0 ? delta : 0.0, 1};                                        // :76-78
                invalid = false;
                if (uni(delta < s.min_delta)) {                                                       // divergence :180
                    invalid = true;
                    cx.term_left = i_n; cx.term_right = i_n;                                          // InvalidTree(i'), tree.jl:332
                    vres = cur_v;
                    for (int k = 0; k < depth; ++k)
                        if ((n >> k) & 1) vres = combine_acc(AccStat{S.lsa[k], usi(S.steps[k])}, vres);   // :347
                    break;
                }
                cur_omega = delta;
                cur_zeta = -1;
                cur_pf = kPfLeaf;
                has_rho = false;
                int k = 0;
                const Vec<NCH> &p = cx.p;
                // one merge at level k (IS0: the level-0 merge, whose left sibling is the previous leaf = cx.pin)
                auto merge_level = [&](auto IS0, const int k) -> bool {
                    constexpr bool kIs0 = decltype(IS0)::value;
                    Vec<NCH> rx, pfx;
                    if constexpr (kIs0) {
                        rx = cx.pin;
                    } else {
                        rx = bload<NCH>(ar + (int64_t)am.stk_rho(k) * L, lane);
                        pfx = bload<NCH>(ar + (int64_t)am.pf(usi(S.pf[k])) * L, lane);
                    }
                    const MergeScalars ms = nuts_merge_scalars(S.lsa[k], cur_v.lsa, S.omega[k], cur_omega);
                    const AccStat vk{ms.lsa, usi(S.steps[k]) + cur_v.steps};                          // tree.jl:347
                    if constexpr (kIs0) {
                        rho = vadd<NCH>(rx, p);                                                       // combine_turn_statistics, NUTS.jl:139-141
                        pfx = psharp<NCH>(minv, rx);
                    } else {
                        rho = vadd<NCH>(rx, rho);
                    }
                    has_rho = true;
                    double d_first, d_last;
                    turn_dots<NCH>(rho, pfx, p, minv, d_first, d_last);                               // is_turning, NUTS.jl:148-170
                    if (uni((d_first < 0.0) | (d_last < 0.0))) {                                      // tree.jl:358
                        invalid = true;
                        cx.term_left = i_start + sgn * (n - (2 << k) + 2);
                        cx.term_right = i_n;
                        vres = vk;
                        for (int j = k + 1; j < depth; ++j)
                            if ((n >> j) & 1) vres = combine_acc(AccStat{S.lsa[j], usi(S.steps[j])}, vres);
                        return false;
                    }
                    // combine_proposals_and_logweights(is_doubling = false), tree.jl:238-245, :361-363
                    const double omega = ms.omega;
                    const double logprob2 = cur_omega - omega;                                        // biased_progressive_logprob2 :261-263
                    bool pick2 = uni(logprob2 >= 0.0);                                                // rand_bool_logprob, NUTS.jl:32-34
                    if (!pick2) pick2 = uni(take_draw() > -logprob2);
                    const int zk = usi(S.zeta[k]);
                    if (pick2) {
                        cx.zfree |= 1u << zk;                                                         // free_z!, NUTS.jl:43
                    } else {
                        if (cur_zeta >= 0) cx.zfree |= 1u << cur_zeta;
                        cur_zeta = zk;
                    }
                    cur_omega = omega;
                    cur_v = vk;
                    if (cur_pf >= 0) cx.pffree |= 1u << cur_pf;                                       // free_rho#!, NUTS.jl:136-137
                    cur_pf = kIs0 ? (int)kPfLevel0 : usi(S.pf[k]);
                    if constexpr (kIs0) {
                        // M^-1 p_in is the p#_first of the two-leaf sub-tree: where that sub-tree parks at level 1 next it goes to its arena slot at once
                        if (!((n >> 1) & 1) && n != nleaves - 1) {
                            const int ps = __builtin_ctz(cx.pffree);
                            cx.pffree &= ~(1u << ps);
                            bstore<NCH>(ar + (int64_t)am.pf(ps) * L, lane, pfx);
                            cur_pf = ps;
                        }
                    }
                    return true;
                };
                if (n & 1) {
                    bool ok = merge_level(BoolC<true>{}, 0);
                    k = 1;
                    while (ok && ((n >> k) & 1)) {
                        ok = merge_level(BoolC<false>{}, k);
                        if (ok) ++k;
                    }
                }
                if (invalid) break;
                C2STAMP(2);
                if (slot == 2) multiply(ox.id);       // third quarter
                C2STAMP(0);
                // materialise the leaf as a proposal candidate if it survived its merges (write-only until the end): q and grad l
                if (cur_zeta < 0) {
                    const int zs = __builtin_ctz(cx.zfree);
                    cx.zfree &= ~(1u << zs);
                    bstore<NCH>(ar + (int64_t)am.zq(zs) * L, lane, lds_load<NCH>(qrow()));
                    bstore<NCH>(ar + (int64_t)(zg0 + zs - 1) * L, lane, vneg(lds_load<NCH>(trow())));
                    S.z_lq[zs] = lq;
                    S.z_pi[zs] = pi;
                    cur_zeta = zs;
                }
                if (n == nleaves - 1) break;                                                          // the whole adjacent tree is in `cur`
                // park the sub-tree summary at level k until its right sibling is complete (level 0: rho = p, the next leapfrog's input)
                if (k != 0) {
                    bstore<NCH>(ar + (int64_t)am.stk_rho(k) * L, lane, rho);
                    S.pf[k] = cur_pf;                 // (>= 0: the level-0 merge moved p#_first to an arena slot)
                }
                S.omega[k] = cur_omega;
                S.lsa[k] = cur_v.lsa;
                S.steps[k] = cur_v.steps;
                S.zeta[k] = cur_zeta;
                cx.n = n + 1;
            }
            if (all_done || leave) break;

            AccStat v{cx.v_lsa, cx.v_steps};
            if (invalid) {
                v = combine_acc(v, vres);                                                             // tree.jl:414, :417
                cx.v_lsa = v.lsa; cx.v_steps = v.steps;
                break;
            }
            const int64_t off = (int64_t)cx.cu * L;
            double *const ar = arena();
            const int fwd = cx.fwd;
            const int i_far = fwd ? cx.i_minus : cx.i_plus;
            const Vec<NCH> p_far = bload<NCH>(i_far ? ar + (int64_t)am.edge_p() * L : s.p + off, lane);
            const Vec<NCH> tr = bload<NCH>(ar + (int64_t)am.top_rho() * L, lane);
            if (fwd) cx.i_plus = i_n; else cx.i_minus = i_n;                                          // :424-428
            if (cur_pf >= 0) cx.pffree |= 1u << cur_pf;
            // combine_proposals_and_logweights(is_doubling = true), tree.jl:431-433
            {
                const MergeScalars mt = nuts_merge_scalars(v.lsa, cur_v.lsa, cx.top_omega, cur_omega);
                cx.v_lsa = mt.lsa; cx.v_steps = v.steps + cur_v.steps;                                // tree.jl:414
                const double logprob2 = cur_omega - cx.top_omega;
                bool pick2 = uni(logprob2 >= 0.0);
                if (!pick2) pick2 = uni(take_draw() > -logprob2);
                if (pick2) {
                    if (cx.top_zeta > 0) cx.zfree |= 1u << cx.top_zeta;
                    cx.top_zeta = cur_zeta;
                } else {
                    cx.zfree |= 1u << cur_zeta;
                }
                cx.top_omega = mt.omega;
            }
            cx.depth += 1;                                                                            // :434
            // whole-tree turn statistic and U-turn test, tree.jl:437-438
            {
                const Vec<NCH> trho = has_rho ? vadd<NCH>(tr, rho) : vadd<NCH>(tr, cx.p);
                bstore<NCH>(ar + (int64_t)am.top_rho() * L, lane, trho);
                double d_other, d_new;
                turn_dots_pp<NCH>(trho, p_far, cx.p, minv, d_other, d_new);
                if (uni((d_other < 0.0) | (d_new < 0.0))) {
                    cx.term_left = cx.i_minus; cx.term_right = cx.i_plus;                             // InvalidTree(i-, i+)
                    break;
                }
            }
        }
        if (all_done) break;
        if (leave) continue;

        // ---- epilogue: TreeStatisticsNUTS (src/NUTS.jl:262), next state, adaptation hooks (as in k_nuts) -----------------------
        {
            const int64_t c = (int64_t)cx.cu, off = c * L;
            Coop2Scalars &S = Sc();
            double *const ar = arena();
            const int top_zeta = cx.top_zeta, depth = cx.depth, term_left = cx.term_left, term_right = cx.term_right;
            const double lq_new = S.z_lq[top_zeta], pi_new = S.z_pi[top_zeta];
            const double a_raw = nuts_dexp(cx.v_lsa) / (double)cx.v_steps;                            // acceptance_rate, NUTS.jl:84
            const double a = a_raw < 1.0 ? a_raw : 1.0;
            Vec<NCH> q;
            if (top_zeta > 0) {
                q = bload<NCH>(ar + (int64_t)am.zq(top_zeta) * L, lane);
                const Vec<NCH> g = bload<NCH>(ar + (int64_t)(zg0 + top_zeta - 1) * L, lane);       // same bits as when it was a leaf
                bstore<NCH, kNt>(s.q + off, lane, q);
                bstore<NCH, kNt>(s.g + off, lane, g);
            } else if (flags & (IDHMC_T_ACCUM_METRIC | IDHMC_T_ACCUM_MOMENTS)) {
                q = bload<NCH>(s.q + off, lane);
            }
            if (lane == 0) {
                if (top_zeta > 0) s.lq[c] = lq_new;
                s.pi[c] = pi_new;
                idhmc_tree_stats st;
                st.pi = pi_new;
                st.acceptance_rate = a;
                st.term_left = term_left; st.term_right = term_right;
                st.depth = depth; st.steps = cx.v_steps;
                s.stats[c] = st;
                atomicAdd(&wg_acc[39], (unsigned long long)cx.v_steps);
            }
            if ((flags & IDHMC_T_ADAPT_EPS) && s.eps_mode == IDHMC_EPS_PER_CHAIN) {
                // adapt_stepsize, src/stepsize.jl:220-229, then current_eps (:235) for the next transition
                const double mu = s.da.mu[c];
                const double m = (double)(s.da.m[c] + 1);
                double Hbar = s.da.Hbar[c], lb = s.da.logeps_bar[c];
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
                    if (e < 1e-10) {                                                                  // src/warmup.jl:291-296
                        s.status[c] = IDHMC_ERR_EPS_UNDERFLOW;
                        atomicMax(s.total_steps + 1, (unsigned long long)IDHMC_ERR_EPS_UNDERFLOW);
                    }
                }
            }
            if (flags & IDHMC_T_ACCUM_METRIC) {
                // running form of the block body of GaussianKineticEnergy!, src/hamiltonian.jl:86-93
                const int nwin = s.mw_n[c];
                if (nwin == 0) {
                    bstore<NCH>(s.mw_x1 + off, lane, q);
                    bstore<NCH>(s.mw_s1 + off, lane, vfill<NCH>(0.0));
                    bstore<NCH>(s.mw_s2 + off, lane, vfill<NCH>(0.0));
                } else {
                    const Vec<NCH> x1 = bload<NCH>(s.mw_x1 + off, lane);
                    Vec<NCH> s1 = bload<NCH>(s.mw_s1 + off, lane);
                    Vec<NCH> s2 = bload<NCH>(s.mw_s2 + off, lane);
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
                const int64_t nm = s.mom_n[c] + 1;
                const double inv = 1.0 / (double)nm;
                Vec<NCH> mean = bload<NCH>(s.mom_mean + off, lane);
                Vec<NCH> m2 = bload<NCH>(s.mom_m2 + off, lane);
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
                const int nd = s.diag.n[c];
                if (nd == 0) {
                    s.diag.pi1[c] = pi_new; s.diag.s1[c] = 0.0; s.diag.s2[c] = 0.0; s.diag.d2[c] = 0.0;
                } else {
                    const double dl = pi_new - s.diag.pi1[c], dp = pi_new - s.diag.prev[c];
                    s.diag.s1[c] = s.diag.s1[c] + dl;
                    s.diag.s2[c] = dfma(dl, dl, s.diag.s2[c]);
                    s.diag.d2[c] = dfma(dp, dp, s.diag.d2[c]);
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
                atomicAdd(cn + 39 + bin, 1ull);
            }
        }
    }
#ifdef IDHMC_STAMPS
    if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(s.total_steps + 2 + i_, (unsigned long long)st_acc[i_]);
#endif
    // the workgroup's sums -> the global counters (every wavefront left at the same barrier)
    __syncthreads();
    if (threadIdx.x < kWgAcc) {
        const unsigned long long x = wg_acc[threadIdx.x];
        if (x) {
            if (threadIdx.x == 39) atomicAdd(s.total_steps, x);
            else if (s.diag.counters) atomicAdd(s.diag.counters + threadIdx.x, x);
        }
    }
}

}  // namespace idhmc
