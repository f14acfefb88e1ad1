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
#ifndef IDHMC_M2_OCC
#define IDHMC_M2_OCC (NCH <= 2 ? 2 : 1)
#endif
#ifndef IDHMC_M2_PD
#define IDHMC_M2_PD (NCH <= 2 ? 4 : 2)
#endif
template <int NCH> struct MfmaDims {
    static constexpr int L = 128 * NCH, KB = L / 4, DS = L + 2;
    static constexpr size_t lds_doubles = 16 * DS + 4 * 16 * 2;
    static constexpr int kWavesPerSimd = IDHMC_M2_OCC;
};

template <int NCH>
__global__ __launch_bounds__(256, MfmaDims<NCH>::kWavesPerSimd) void k_leapfrog_dense_mfma(DevState s, double eps_arg,
                                                                                            int own_eps, int n_steps)
{
    using M = MfmaDims<NCH>;
    constexpr int L = M::L, KB = M::KB, DS = M::DS, PD = IDHMC_M2_PD;
    static_assert(KB % PD == 0 && (KB & (KB - 1)) == 0, "prefetch depth must divide the k-block count (a power of two)");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *dT = lds;                 // [16 chains][DS] tile of d = q' - mu
    double *red = lds + 16 * DS;      // [4 waves][16 chains][2] partial sums of l and K
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kk = lane >> 4, jj = lane & 15;
    const int col0 = 32 * wv + 2 * jj;                     // this lane's column pair inside each 128-chunk
    const int64_t ntiles = (s.C + 15) / 16;
    const double *ap = dT + jj * DS + kk;
    // all global accesses go through raw buffer resources (bload/bstore in idhmc_device.hpp explain why): wave-uniform
    // bases in SGPRs, one 32-bit per-lane offset per chain row, chunk and k-block offsets in the immediate / SGPR field
    const __amdgpu_buffer_rsrc_t rP = buf_rsrc(s.prec), rMu = buf_rsrc(s.mu);
    const int pvo = (kk * L + col0) * 8;                   // P: row 4 kb + kk, this lane's column pair
    auto ld2 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) -> v2d {
        return __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto st2 = [](__amdgpu_buffer_rsrc_t r, int voff, v2d v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v), r, voff, 0, 0);
    };

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t c0 = tile * 16;
        const __amdgpu_buffer_rsrc_t rQ = buf_rsrc(s.q + c0 * L), rPm = buf_rsrc(s.p + c0 * L), rG = buf_rsrc(s.g + c0 * L),
                                     rM = buf_rsrc(s.minv + c0 * s.minv_stride);
        int rowo[4], mrowo[4];        // byte offsets of this lane's four chain rows inside the tile (clamped at the ragged end)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t chain = c0 + kk + 4 * reg;
            const int rel = (int)((chain < s.C ? chain : s.C - 1) - c0);
            rowo[reg] = (rel * L + col0) * 8;
            mrowo[reg] = (rel * (int)s.minv_stride + col0) * 8;
        }
        v2d pv[4][NCH], qv[4][NCH];   // accumulator layout: [reg] = chain c0 + kk + 4 reg, columns 128 j + col0 + {0,1}
        v4d acc[NCH][2];              // T = Dm P, i.e. minus the gradient; [j][e][reg]
        double eps4[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t chain = c0 + kk + 4 * reg;
            const int64_t ch = chain < s.C ? chain : s.C - 1;
            eps4[reg] = own_eps ? s.eps[ch] : eps_arg;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const v2d g2 = ld2(rG, rowo[reg] + 1024 * j, 0);
                pv[reg][j] = ld2(rPm, rowo[reg] + 1024 * j, 0);
                qv[reg][j] = ld2(rQ, rowo[reg] + 1024 * j, 0);
                acc[j][0][reg] = -g2.x;
                acc[j][1][reg] = -g2.y;
            }
        }
        for (int step = 0; step < n_steps; ++step) {
            if (step) __syncthreads();   // every wavefront is out of the previous step's k loop (reads of dT)
            // ---- loop A (src/kinetic_energy.jl:146-150) in the accumulator layout --------------------
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double eps = eps4[reg], eh = 0.5 * eps;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int r = 128 * j + col0;
                    const v2d m2 = ld2(rM, mrowo[reg] + 1024 * j, 0), u2 = ld2(rMu, col0 * 8 + 1024 * j, 0);
                    v2d pm, qn, d;
                    pm.x = dfma(eh, -acc[j][0][reg], pv[reg][j].x); pm.y = dfma(eh, -acc[j][1][reg], pv[reg][j].y);
                    qn.x = dfma(eps * m2.x, pm.x, qv[reg][j].x); qn.y = dfma(eps * m2.y, pm.y, qv[reg][j].y);
                    d.x = qn.x - u2.x; d.y = qn.y - u2.y;
                    pv[reg][j] = pm;
                    qv[reg][j] = qn;
                    *reinterpret_cast<v2d *>(dT + (kk + 4 * reg) * DS + r) = d;
                }
                sched_fence();   // bound the loads in flight to one chain quad (register pressure)
            }
            __syncthreads();
            // ---- T = Dm * P on the matrix cores: k-blocks ascending, P prefetched PD blocks ahead ------
#pragma unroll
            for (int j = 0; j < NCH; ++j) { acc[j][0] = v4d{0.0, 0.0, 0.0, 0.0}; acc[j][1] = v4d{0.0, 0.0, 0.0, 0.0}; }
            v2d bq[PD][NCH];
#pragma unroll
            for (int u = 0; u < PD; ++u)
#pragma unroll
                for (int j = 0; j < NCH; ++j) bq[u][j] = ld2(rP, pvo + 1024 * j, 4 * u * L * 8);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) once, so that the loop waits per block (see DenseMvnCoop::multiply)
#pragma unroll 1
            for (int kb0 = 0; kb0 < KB; kb0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int kb = kb0 + u;
                    const double a = ap[4 * kb];
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        acc[j][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[u][j].x, acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[u][j].y, acc[j][1], 0, 0, 0);
                    }
                    // unconditional (the last trips wrap around and are discarded): a branch here makes the
                    // compiler drain all outstanding loads at every trip
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
                        bq[u][j] = ld2(rP, pvo + 1024 * j, 4 * ((kb + PD) & (KB - 1)) * L * 8);
                }
            }
            // ---- loop B (src/kinetic_energy.jl:159-161) ------------------------------------------------
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double eh = 0.5 * eps4[reg];
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    pv[reg][j].x = dfma(eh, -acc[j][0][reg], pv[reg][j].x);
                    pv[reg][j].y = dfma(eh, -acc[j][1][reg], pv[reg][j].y);
                }
            }
        }
        // ---- write the phase point back; K(p'), l(q') -------------------------------------------------
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t chain = c0 + kk + 4 * reg;
            const bool valid = chain < s.C;
            double la0 = 0.0, la1 = 0.0, ka0 = 0.0, ka1 = 0.0;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int r = 128 * j + col0;
                const double t0 = acc[j][0][reg], t1 = acc[j][1][reg];
                const v2d d = *reinterpret_cast<const v2d *>(dT + (kk + 4 * reg) * DS + r);   // this lane wrote it
                const v2d m2 = ld2(rM, mrowo[reg] + 1024 * j, 0);
                const v2d pn = pv[reg][j];
                v2d gn;
                gn.x = -t0; gn.y = -t1;
                la0 = dfma(t0, d.x, la0); la1 = dfma(t1, d.y, la1);
                ka0 = dfma(pn.x * m2.x, pn.x, ka0); ka1 = dfma(pn.y * m2.y, pn.y, ka1);
                if (valid) {
                    st2(rQ, rowo[reg] + 1024 * j, qv[reg][j]);
                    st2(rPm, rowo[reg] + 1024 * j, pn);
                    st2(rG, rowo[reg] + 1024 * j, gn);
                }
            }
            // canonical tree over the 128 residues rho = 32 w + 2 jj + e: bit 0 in the lane, bits 1..4 across
            // the 16 lanes of the row group, bits 5..6 across the wavefronts (through LDS below)
            double la = la0 + la1, ka = ka0 + ka1;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { la = dpp_xor_add(la, m); ka = dpp_xor_add(ka, m); }
            if (jj == 0) {
                red[(wv * 16 + kk + 4 * reg) * 2 + 0] = la;
                red[(wv * 16 + kk + 4 * reg) * 2 + 1] = ka;
            }
            sched_fence();
        }
        __syncthreads();
        if (tid < 16 && c0 + tid < s.C) {
            const double l = (red[(0 * 16 + tid) * 2] + red[(1 * 16 + tid) * 2]) + (red[(2 * 16 + tid) * 2] + red[(3 * 16 + tid) * 2]);
            const double k = (red[(0 * 16 + tid) * 2 + 1] + red[(1 * 16 + tid) * 2 + 1]) +
                             (red[(2 * 16 + tid) * 2 + 1] + red[(3 * 16 + tid) * 2 + 1]);
            double lq = -0.5 * l;
            lq = dfinite(lq) ? lq : -kInf;
            s.lq[c0 + tid] = lq;
            s.pi[c0 + tid] = phase_logdensity(lq, 0.5 * k);
        }
        // no trailing barrier: `red` is rewritten only after the next tile's loop-A barrier, which wavefront 0 reaches
        // after these reads; dT is rewritten by a lane only at positions that lane alone reads outside the k loop, and
        // every wavefront left the k loop before the barrier above.
    }
}

template <int NCH>
static hipError_t launch_mfma_t(const DevState &s, double eps, int own, int n_steps, hipStream_t st)
{
    using M = MfmaDims<NCH>;
    const size_t bytes = M::lds_doubles * sizeof(double);
    static bool attr_done[64] = {};  // per instantiation and device (the attribute is per device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leapfrog_dense_mfma<NCH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    int64_t grid = (s.C + 15) / 16;
    const int64_t resident = 256 * M::kWavesPerSimd;
    if (grid > resident) grid = resident;
    hipLaunchKernelGGL((k_leapfrog_dense_mfma<NCH>), dim3((unsigned)grid), dim3(256), bytes, st, s, eps, own, n_steps);
    return hipGetLastError();
}

// returns hipErrorNotSupported when the shape is outside the MFMA kernel's range (L > 512)
hipError_t launch_leapfrog_dense_mfma(const DevState &s, double eps, int own, int n_steps, hipStream_t st)
{
    if (s.nch == 1) return launch_mfma_t<1>(s, eps, own, n_steps, st);
    if (s.nch == 2) return launch_mfma_t<2>(s, eps, own, n_steps, st);
    if (s.nch == 4) return launch_mfma_t<4>(s, eps, own, n_steps, st);
    return hipErrorNotSupported;
}

}  // namespace idhmc
