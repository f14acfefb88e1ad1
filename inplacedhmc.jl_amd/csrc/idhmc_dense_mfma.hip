// idhmc_dense_mfma.hip -- fused leapfrog (n steps per launch) for the dense multivariate normal with the
// Sigma^-1 (q - mu) gradient on the fp64 matrix cores (BASELINE.json configs[3]: D = 256, 16 384 chains):
// T[16 chains][L] = Dm[16][L] * P[L][L] per step with v_mfma_f64_16x16x4_f64 (A = 16 chains x 4 columns of
// d = q' - mu, B = 4 rows x 16 columns of P).
// Summation order = the engine's defined order for this density (ascending column, one fma chain per
// output element: k-blocks ascending, k = 0..3 inside the MFMA), and the canonical 128-residue tree for the
// two reductions, so the result is bit-identical to the per-wave GEMV kernel and to the CPU oracle.
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"
#include <cstdlib>

namespace idhmc {

IDHMC_DEV double dpp_xor_add(double v, int m) { return v + __shfl_xor(v, m, 64); }

// ---------------------------------------------------------------------------------------------------------
// Column-split tiling for CDNA4: a workgroup is ONE 16-chain tile; its 4 wavefronts split the output
// columns (wave w owns residues 32w..32w+31 of every 128-column chunk).  The B operand (P) is read straight
// from L2 in the MFMA layout with 16-byte loads -- lane (kk, jj) holds columns 2jj, 2jj+1 of the block, i.e.
// the two MFMA tiles of a 32-column block are its even and its odd columns -- so there is no LDS panel and no
// barrier in the k loop, and several workgroups are resident per CU.  The phase point (q, p in VGPRs, the
// gradient as -T in the accumulators) stays on chip for all n_steps of a call: only the first step reads and
// only the last one writes HBM, so a multi-step call runs at the matrix-core rate.
// TR = row tiles (16 chains each) per workgroup.  -DIDHMC_M2_TR=2 gives the single-step form 32-chain tiles, i.e. two MFMAs per
// B operand fetched from L2 (the matrix is re-read once per tile: 512 KiB x 1024 tiles = 2.7x the state's traffic at configs[3]).
// Measured in round 2 and not the default: the matrix traffic is not what the sweep waits for (without it the
// sweep is 3 % shorter), and 232 registers leave two workgroups per CU instead of three.
#ifndef IDHMC_M2_TR
#define IDHMC_M2_TR 1
#endif
#ifndef IDHMC_M2_OCC
#define IDHMC_M2_OCC (NCH <= 2 ? (SINGLE && TR == 1 ? 3 : 2) : 1)
#endif
#ifndef IDHMC_M2_W
#define IDHMC_M2_W 4
#endif
#ifndef IDHMC_M2_PD
#define IDHMC_M2_PD (NCH <= 2 ? 4 : 2)
#endif
#ifdef IDHMC_STAMPS      // diagnostic build (tools/stamps.sh): wall-clock (100 MHz) phase sums of wavefront 0 into total_steps[2..]
#define DSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (tid == 0) atomicAdd(s.total_steps + 2 + (i), t_ - st_t); st_t = t_; } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif
template <int NCH, bool SINGLE = false, int TR = 1> struct MfmaDims {
    static constexpr int L = 128 * NCH, KB = L / 4, DS = L + 2, ROWS = 16 * TR;
    static constexpr size_t lds_doubles = ROWS * DS + 4 * ROWS * 2;
    static constexpr int kWavesPerSimd = IDHMC_M2_OCC;
};

template <int NCH, bool SINGLE, int TR>
__global__ __launch_bounds__(256, (MfmaDims<NCH, SINGLE, TR>::kWavesPerSimd)) void k_leapfrog_dense_mfma(
    DevState s, double eps_arg, int own_eps, int n_steps_arg, int64_t tile_begin, int64_t tile_end)   // tiles of 16 TR chains
{
    // SINGLE: one step per launch; q' leaves in loop A (under the matrix phase) and its registers are free from there on
    const int n_steps = SINGLE ? 1 : n_steps_arg;
    using M = MfmaDims<NCH, SINGLE, TR>;
    constexpr int L = M::L, KB = M::KB, DS = M::DS, ROWS = M::ROWS, PD = IDHMC_M2_PD;
    static_assert(KB % PD == 0 && (KB & (KB - 1)) == 0, "prefetch depth must divide the k-block count (a power of two)");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *dT = lds;                 // [ROWS chains][DS] tile of d = q' - mu
    double *red = lds + ROWS * DS;    // [4 waves][ROWS chains][2] partial sums of l and K
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kk = lane >> 4, jj = lane & 15;
    const int col0 = 32 * wv + 2 * jj;                     // this lane's column pair inside each 128-chunk
    const double *ap = dT + jj * DS + kk;                  // A operand of row tile t: ap[t * 16 * DS + 4 kb]
    // all global accesses go through raw buffer resources (bload/bstore in idhmc_device.hpp explain why): wave-uniform
    // bases in SGPRs, one 32-bit per-lane offset per chain row, chunk and k-block offsets in the immediate / SGPR field
    const __amdgpu_buffer_rsrc_t rP = buf_rsrc(s.prec), rMu = buf_rsrc(s.mu);
    const int pvo = (kk * L + col0) * 8;                   // P: row 4 kb + kk, this lane's column pair
    auto ld2 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) -> v2d {
        return __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto st2 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff, v2d v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v), r, voff, soff, 0);
    };

    for (int64_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
        const int64_t c0 = tile * ROWS;
#ifdef IDHMC_STAMPS
        unsigned long long st_t = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) atomicAdd(s.total_steps + 7, 1ull);
#endif
        const __amdgpu_buffer_rsrc_t rQ = buf_rsrc(s.q + c0 * L), rPm = buf_rsrc(s.p + c0 * L), rG = buf_rsrc(s.g + c0 * L),
                                     rM = buf_rsrc(s.minv + c0 * s.minv_stride);
        // this lane's rows: chain c0 + 16 t + kk + 4 reg (t = row tile, reg = accumulator register).  One per-lane offset
        // (row kk) and a wave-uniform one per (t, reg); rows past the last chain of a ragged tile are read (the state arrays
        // of a dense context are allocated kRowPad rows longer, idhmc_create) and never stored
        const int rowv = (kk * L + col0) * 8, mrowv = (kk * (int)s.minv_stride + col0) * 8;
        const int mrow_step = 4 * (int)s.minv_stride * 8;
        v2d pv[TR][4][NCH], qv[TR][4][NCH];   // accumulator layout: columns 128 j + col0 + {0,1}
        v4d acc[TR][NCH][2];                  // T = Dm P, i.e. minus the gradient; [t][j][e][reg]
        double eps4[TR][4];                   // (not SINGLE) the rows' stepsizes, kept for the steps of the call
        auto eps_at = [&](int t, int reg) -> double {
            const int64_t chain = c0 + 16 * t + kk + 4 * reg;
            return own_eps ? s.eps[chain < s.C ? chain : s.C - 1] : eps_arg;
        };
        if constexpr (SINGLE) {
            // ---- load + loop A, one chain quad (this lane's row `reg` of row tile t) at a time, the loads of W quads in
            // flight: the whole tile at once would hold 3 vectors x 16 TR rows in registers next to the matrix phase's own
            constexpr int NQ = 4 * TR, W = IDHMC_M2_W;
            v2d gb[W][NCH], pb[W][NCH], qb[W][NCH];
#pragma unroll
            for (int i = 0; i < W; ++i)
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    gb[i][j] = ld2(rG, rowv + 1024 * j, (16 * (i / 4) + 4 * (i % 4)) * L * 8);
                    pb[i][j] = ld2(rPm, rowv + 1024 * j, (16 * (i / 4) + 4 * (i % 4)) * L * 8);
                    qb[i][j] = ld2(rQ, rowv + 1024 * j, (16 * (i / 4) + 4 * (i % 4)) * L * 8);
                }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int t = i / 4, reg = i % 4, b = i % W;
                const double eps = eps_at(t, reg), eh = 0.5 * eps;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int r = 128 * j + col0;
                    const v2d m2 = ld2(rM, mrowv + 1024 * j, (4 * (t) + (reg)) * mrow_step), u2 = ld2(rMu, col0 * 8 + 1024 * j, 0);
                    v2d pm, qn, d;
                    pm.x = dfma(eh, gb[b][j].x, pb[b][j].x); pm.y = dfma(eh, gb[b][j].y, pb[b][j].y);
                    qn.x = dfma(eps * m2.x, pm.x, qb[b][j].x); qn.y = dfma(eps * m2.y, pm.y, qb[b][j].y);
                    d.x = qn.x - u2.x; d.y = qn.y - u2.y;
                    pv[t][reg][j] = pm;
                    if (c0 + 16 * t + kk + 4 * reg < s.C) st2(rQ, rowv + 1024 * j, (16 * (t) + 4 * (reg)) * L * 8, qn);
                    *reinterpret_cast<v2d *>(dT + (16 * t + kk + 4 * reg) * DS + r) = d;
                }
                if (i + W < NQ) {
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        gb[b][j] = ld2(rG, rowv + 1024 * j, (16 * ((i + W) / 4) + 4 * ((i + W) % 4)) * L * 8);
                        pb[b][j] = ld2(rPm, rowv + 1024 * j, (16 * ((i + W) / 4) + 4 * ((i + W) % 4)) * L * 8);
                        qb[b][j] = ld2(rQ, rowv + 1024 * j, (16 * ((i + W) / 4) + 4 * ((i + W) % 4)) * L * 8);
                    }
                }
                sched_fence();
            }
        } else {
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    eps4[t][reg] = eps_at(t, reg);
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const v2d g2 = ld2(rG, rowv + 1024 * j, (16 * (t) + 4 * (reg)) * L * 8);
                        pv[t][reg][j] = ld2(rPm, rowv + 1024 * j, (16 * (t) + 4 * (reg)) * L * 8);
                        qv[t][reg][j] = ld2(rQ, rowv + 1024 * j, (16 * (t) + 4 * (reg)) * L * 8);
                        acc[t][j][0][reg] = -g2.x;
                        acc[t][j][1][reg] = -g2.y;
                    }
                }
        }
        for (int step = 0; step < n_steps; ++step) {
            if (step) __syncthreads();   // every wavefront is out of the previous step's k loop (reads of dT)
            // ---- loop A (src/kinetic_energy.jl:146-150) in the accumulator layout --------------------
            if constexpr (!SINGLE) {
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const double eps = eps4[t][reg], eh = 0.5 * eps;
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const int r = 128 * j + col0;
                        const v2d m2 = ld2(rM, mrowv + 1024 * j, (4 * (t) + (reg)) * mrow_step), u2 = ld2(rMu, col0 * 8 + 1024 * j, 0);
                        v2d pm, qn, d;
                        pm.x = dfma(eh, -acc[t][j][0][reg], pv[t][reg][j].x); pm.y = dfma(eh, -acc[t][j][1][reg], pv[t][reg][j].y);
                        qn.x = dfma(eps * m2.x, pm.x, qv[t][reg][j].x); qn.y = dfma(eps * m2.y, pm.y, qv[t][reg][j].y);
                        d.x = qn.x - u2.x; d.y = qn.y - u2.y;
                        pv[t][reg][j] = pm;
                        qv[t][reg][j] = qn;
                        *reinterpret_cast<v2d *>(dT + (16 * t + kk + 4 * reg) * DS + r) = d;
                    }
                    sched_fence();   // bound the loads in flight to one chain quad (register pressure)
                }
            }
            __syncthreads();
            DSTAMP(0);   // load + loop A + barrier
            // ---- T = Dm * P on the matrix cores: k-blocks ascending, P prefetched PD blocks ahead ------
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int j = 0; j < NCH; ++j) { acc[t][j][0] = v4d{0.0, 0.0, 0.0, 0.0}; acc[t][j][1] = v4d{0.0, 0.0, 0.0, 0.0}; }
            v2d bq[PD][NCH];
#pragma unroll
            for (int u = 0; u < PD; ++u)
#pragma unroll
                for (int j = 0; j < NCH; ++j) bq[u][j] = ld2(rP, pvo + 1024 * j, 4 * u * L * 8);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) once, so that the loop waits per block (see DenseMvnCoop::multiply)
            // software pipeline, pinned with scheduling fences: block kb's A operand is read from LDS one block ahead and block
            // kb + PD's B operands are requested right behind block kb's MFMAs.  Left alone the compiler gathers the PD blocks'
            // requests at the end of the unrolled body, which leaves the first of them one block (256 cycles) of lead instead of PD.
            double a_nx[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) a_nx[t] = ap[t * 16 * DS];
#pragma unroll 1
            for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int kb = kb0 + u;
                    double a[TR];
#pragma unroll
                    for (int t = 0; t < TR; ++t) { a[t] = a_nx[t]; a_nx[t] = ap[t * 16 * DS + 4 * ((kb + 1) & (KB - 1))]; }
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
#pragma unroll
                        for (int t = 0; t < TR; ++t) {
#ifdef IDHMC_DX2      // (cost attribution) no matrix-core work: one fma per operand instead
                            acc[t][j][0][0] = dfma(a[t], bq[u][j].x, acc[t][j][0][0]);
                            acc[t][j][1][0] = dfma(a[t], bq[u][j].y, acc[t][j][1][0]);
#else
                            acc[t][j][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bq[u][j].x, acc[t][j][0], 0, 0, 0);
                            acc[t][j][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bq[u][j].y, acc[t][j][1], 0, 0, 0);
#endif
                        }
                    // unconditional (the last trips wrap around and are discarded): a branch here makes the
                    // compiler drain all outstanding loads at every trip
#ifndef IDHMC_DX1      // (cost attribution, results wrong on purpose) IDHMC_DX1: the matrix is fetched once, not per k-block
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
                        bq[u][j] = ld2(rP, pvo + 1024 * j, 4 * ((kb + PD) & (KB - 1)) * L * 8);
#endif
                    sched_fence();
                }
            }
            DSTAMP(1);   // k loop
            // ---- loop B (src/kinetic_energy.jl:159-161) ------------------------------------------------
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const double eh = 0.5 * (SINGLE ? eps_at(t, reg) : eps4[t][reg]);
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        pv[t][reg][j].x = dfma(eh, -acc[t][j][0][reg], pv[t][reg][j].x);
                        pv[t][reg][j].y = dfma(eh, -acc[t][j][1][reg], pv[t][reg][j].y);
                    }
                }
        }
        // ---- write the phase point back; K(p'), l(q') -------------------------------------------------
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * t + kk + 4 * reg;
                const bool valid = c0 + row < s.C;
                double la0 = 0.0, la1 = 0.0, ka0 = 0.0, ka1 = 0.0;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int r = 128 * j + col0;
                    const double t0 = acc[t][j][0][reg], t1 = acc[t][j][1][reg];
                    const v2d d = *reinterpret_cast<const v2d *>(dT + row * DS + r);   // this lane wrote it
                    const v2d m2 = ld2(rM, mrowv + 1024 * j, (4 * (t) + (reg)) * mrow_step);
                    const v2d pn = pv[t][reg][j];
                    v2d gn;
                    gn.x = -t0; gn.y = -t1;
                    la0 = dfma(t0, d.x, la0); la1 = dfma(t1, d.y, la1);
                    ka0 = dfma(pn.x * m2.x, pn.x, ka0); ka1 = dfma(pn.y * m2.y, pn.y, ka1);
                    if (valid) {
                        if constexpr (!SINGLE) st2(rQ, rowv + 1024 * j, (16 * (t) + 4 * (reg)) * L * 8, qv[t][reg][j]);
                        st2(rPm, rowv + 1024 * j, (16 * (t) + 4 * (reg)) * L * 8, pn);
                        st2(rG, rowv + 1024 * j, (16 * (t) + 4 * (reg)) * L * 8, gn);
                    }
                }
                // canonical tree over the 128 residues rho = 32 w + 2 jj + e: bit 0 in the lane, bits 1..4 across
                // the 16 lanes of the row group, bits 5..6 across the wavefronts (through LDS below)
                double la = la0 + la1, ka = ka0 + ka1;
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { la = dpp_xor_add(la, m); ka = dpp_xor_add(ka, m); }
                if (jj == 0) {
                    red[(wv * ROWS + row) * 2 + 0] = la;
                    red[(wv * ROWS + row) * 2 + 1] = ka;
                }
                sched_fence();
            }
        __syncthreads();
        if (tid < ROWS && c0 + tid < s.C) {
            const double l = (red[(0 * ROWS + tid) * 2] + red[(1 * ROWS + tid) * 2]) + (red[(2 * ROWS + tid) * 2] + red[(3 * ROWS + tid) * 2]);
            const double k = (red[(0 * ROWS + tid) * 2 + 1] + red[(1 * ROWS + tid) * 2 + 1]) +
                             (red[(2 * ROWS + tid) * 2 + 1] + red[(3 * ROWS + tid) * 2 + 1]);
            double lq = -0.5 * l;
            lq = dfinite(lq) ? lq : -kInf;
            s.lq[c0 + tid] = lq;
            s.pi[c0 + tid] = phase_logdensity(lq, 0.5 * k);
        }
#ifdef IDHMC_STAMPS
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the stores have left
        DSTAMP(2);   // loop B + stores + reductions
#endif
        // no trailing barrier: `red` is rewritten only after the next tile's loop-A barrier, which wavefront 0 reaches
        // after these reads; dT is rewritten by a lane only at positions that lane alone reads outside the k loop, and
        // every wavefront left the k loop before the barrier above.
    }
}

// tile_begin / tile_end count 16-chain tiles; a range handed to the 32-chain form must begin at an even tile
template <int NCH>
static hipError_t launch_mfma_t(const DevState &s, double eps, int own, int n_steps, int64_t tile_begin, int64_t tile_end,
                                int64_t max_grid, hipStream_t st)
{
    constexpr int TR1 = (NCH <= 2 && IDHMC_M2_TR == 2) ? 2 : 1;     // row tiles per workgroup of the single-step form
    using M = MfmaDims<NCH, false, 1>;
    using M1 = MfmaDims<NCH, true, TR1>;
    static bool attr_done[64] = {};  // per instantiation and device (the attribute is per device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leapfrog_dense_mfma<NCH, false, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(M::lds_doubles * sizeof(double)));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leapfrog_dense_mfma<NCH, true, TR1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(M1::lds_doubles * sizeof(double)));
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    if (n_steps == 1) {
        if (tile_begin % TR1) return hipErrorInvalidValue;
        const int64_t tb = tile_begin / TR1, te = (tile_end + TR1 - 1) / TR1;
        int64_t grid = te - tb;
        const int64_t resident = 256 * M1::kWavesPerSimd;
        if (grid > resident) grid = resident;
        if (max_grid > 0 && grid > max_grid) grid = max_grid;
        if (grid < 1) return hipSuccess;
        hipLaunchKernelGGL((k_leapfrog_dense_mfma<NCH, true, TR1>), dim3((unsigned)grid), dim3(256), M1::lds_doubles * sizeof(double),
                           st, s, eps, own, n_steps, tb, te);
    } else {
        int64_t grid = tile_end - tile_begin;
        const int64_t resident = 256 * M::kWavesPerSimd;
        if (grid > resident) grid = resident;
        if (max_grid > 0 && grid > max_grid) grid = max_grid;
        if (grid < 1) return hipSuccess;
        hipLaunchKernelGGL((k_leapfrog_dense_mfma<NCH, false, 1>), dim3((unsigned)grid), dim3(256), M::lds_doubles * sizeof(double),
                           st, s, eps, own, n_steps, tile_begin, tile_end);
    }
    return hipGetLastError();
}

// returns hipErrorNotSupported when the shape is outside the MFMA kernel's range (L > 512)
hipError_t launch_leapfrog_dense_mfma_tiles(const DevState &s, double eps, int own, int n_steps, int64_t tile_begin, int64_t tile_end,
                                            int64_t max_grid, hipStream_t st)
{
    if (s.nch == 1) return launch_mfma_t<1>(s, eps, own, n_steps, tile_begin, tile_end, max_grid, st);
    if (s.nch == 2) return launch_mfma_t<2>(s, eps, own, n_steps, tile_begin, tile_end, max_grid, st);
    if (s.nch == 4) return launch_mfma_t<4>(s, eps, own, n_steps, tile_begin, tile_end, max_grid, st);
    return hipErrorNotSupported;
}
int dense_mfma_tile_align(const DevState &s) { return (s.nch <= 2 && IDHMC_M2_TR == 2) ? 2 : 1; }
hipError_t launch_leapfrog_dense_mfma(const DevState &s, double eps, int own, int n_steps, hipStream_t st)
{
    return launch_leapfrog_dense_mfma_tiles(s, eps, own, n_steps, 0, (s.C + 15) / 16, 0, st);
}

}  // namespace idhmc
