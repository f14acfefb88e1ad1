// idhmc_dense_mfma.hip -- fused single-step leapfrog for the dense multivariate normal with the
// Sigma^-1 (q - mu) gradient on the fp64 matrix cores (BASELINE.json configs[3]: D = 256, 16 384 chains).
//
// Tiling for CDNA4: a workgroup is 4 wavefronts (one per SIMD, 512-register budget each); each wavefront
// owns a tile of 16 chains, so the workgroup computes T[64 chains][L] = Dm[64][L] * P[L][L] per step with
// v_mfma_f64_16x16x4_f64 (A = 16 chains x 4 columns of d = q' - mu, B = 4 rows x 16 columns of P).
//   * P streams from L2 through a double-buffered 4-row LDS panel loaded cooperatively (8 KiB contiguous
//     per k-block, 32 bytes per thread), shared by the 4 wavefronts;
//   * d is produced by loop A in the MFMA *output* layout (lane = (chain quad, column mod 16)), written once
//     to a padded LDS tile and read back as the A operand (lane = (chain, k)) -- the one transpose;
//   * grad' = -T comes out of the accumulators in the same lane layout loop A used, so loop B, K(p') and
//     l(q') need no further data movement.
// Summation order = the engine's defined order for this density (ascending column, one fma chain per
// output element: k-blocks ascending, k = 0..3 inside the MFMA), and the canonical 128-residue tree for the
// two reductions, so the result is bit-identical to the per-wave GEMV kernel and to the CPU oracle.
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"

namespace idhmc {

typedef double v4d __attribute__((ext_vector_type(4)));

template <int L> struct MfmaDims {
    static constexpr int KB = L / 4, RT = L / 16, DS = L + 2, PS = L + 16, PT = L / 64;
    static constexpr size_t lds_doubles = 4 * 16 * DS + 2 * 4 * PS;
};

IDHMC_DEV double dpp_xor_add(double v, int m) { return v + __shfl_xor(v, m, 64); }

template <int NCH>
__global__ __launch_bounds__(256, 1) void k_leapfrog_dense_mfma(DevState s, double eps_arg, int own_eps)
{
    constexpr int L = 128 * NCH;
    using M = MfmaDims<L>;
    constexpr int KB = M::KB, RT = M::RT, DS = M::DS, PS = M::PS, PT = M::PT, NA = (RT < 8 ? RT : 8);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, kk = lane >> 4, jj = lane & 15;
    double *dT = lds + (size_t)wv * 16 * DS;          // this wavefront's [16 chains][DS] tile of d
    double *panel = lds + (size_t)4 * 16 * DS;        // [2][4 rows][PS] of P
    const int prow = tid >> 6, pcol = (tid & 63) * PT; // this thread's share of a 4-row panel

    for (int64_t base = (int64_t)blockIdx.x * 64; base < s.C; base += (int64_t)gridDim.x * 64) {
        const int64_t c0 = base + wv * 16;
        double pmr[4][RT];
        double eps4[4];
        // ---- loop A (src/kinetic_energy.jl:146-150) in the accumulator layout ------------------------
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t chain = c0 + kk + 4 * reg;
            const bool valid = chain < s.C;
            const int64_t ch = valid ? chain : s.C - 1;
            const double eps = own_eps ? s.eps[ch] : eps_arg;
            eps4[reg] = eps;
            const double eh = 0.5 * eps;
            const double *qp = s.q + ch * L, *pp = s.p + ch * L, *gp = s.g + ch * L, *mp = s.minv + ch * s.minv_stride;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int r = 16 * rt + jj;
                const double pm = dfma(eh, gp[r], pp[r]);
                const double qn = dfma(eps * mp[r], pm, qp[r]);
                const double d = qn - s.mu[r];
                if (valid) s.q[ch * L + r] = qn;
                pmr[reg][rt] = pm;
                dT[(kk + 4 * reg) * DS + r] = d;
            }
            sched_fence();   // bound the loads in flight to one chain quad (register pressure)
        }
        // ---- T = Dm * P on the matrix cores ----------------------------------------------------------
        v4d acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = v4d{0.0, 0.0, 0.0, 0.0};
        double pre[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) pre[i] = s.prec[(size_t)prow * L + pcol + i];
        for (int kb = 0; kb < KB; ++kb) {
            double *buf = panel + (size_t)(kb & 1) * 4 * PS;
#pragma unroll
            for (int i = 0; i < PT; ++i) buf[prow * PS + pcol + i] = pre[i];
            __syncthreads();
            if (kb + 1 < KB) {
#pragma unroll
                for (int i = 0; i < PT; ++i) pre[i] = s.prec[(size_t)(4 * (kb + 1) + prow) * L + pcol + i];
            }
            const double a = dT[jj * DS + 4 * kb + kk];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const double b = buf[kk * PS + 16 * rt + jj];
                acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[rt], 0, 0, 0);
            }
        }
        // ---- loop B (src/kinetic_energy.jl:159-161), K(p'), l(q') -------------------------------------
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t chain = c0 + kk + 4 * reg;
            const bool valid = chain < s.C;
            const int64_t ch = valid ? chain : s.C - 1;
            const double eh = 0.5 * eps4[reg];
            const double *mp = s.minv + ch * s.minv_stride;
            double la[NA], ka[NA];
#pragma unroll
            for (int t = 0; t < NA; ++t) { la[t] = 0.0; ka[t] = 0.0; }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int r = 16 * rt + jj;
                const double t = acc[rt][reg];
                const double pn = dfma(eh, -t, pmr[reg][rt]);
                la[rt & 7] = dfma(t, dT[(kk + 4 * reg) * DS + r], la[rt & 7]);   // d re-read from this wave's LDS tile
                ka[rt & 7] = dfma(pn * mp[r], pn, ka[rt & 7]);
                if (valid) {
                    s.p[ch * L + r] = pn;
                    s.g[ch * L + r] = -t;
                }
            }
            // canonical tree over the 128 residues r mod 128 = 16 (rt mod 8) + jj: bits 0..3 across lanes,
            // bits 4..6 inside the lane
#pragma unroll
            for (int t = 0; t < NA; ++t) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { la[t] = dpp_xor_add(la[t], m); ka[t] = dpp_xor_add(ka[t], m); }
            }
#pragma unroll
            for (int w = 1; w < NA; w <<= 1)
#pragma unroll
                for (int t = 0; t < NA; t += 2 * w) { la[t] = la[t] + la[t + w]; ka[t] = ka[t] + ka[t + w]; }
            double lq = -0.5 * la[0];
            lq = dfinite(lq) ? lq : -kInf;
            if (valid && jj == 0) {
                s.lq[ch] = lq;
                s.pi[ch] = phase_logdensity(lq, 0.5 * ka[0]);
            }
            sched_fence();
        }
        __syncthreads();   // the panel buffers are reused by the next tile
    }
}

template <int NCH>
static hipError_t launch_mfma_t(const DevState &s, double eps, int own, hipStream_t st)
{
    using M = MfmaDims<128 * NCH>;
    const size_t bytes = M::lds_doubles * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_leapfrog_dense_mfma<NCH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    int64_t grid = (s.C + 63) / 64;
    if (grid > 256) grid = 256;      // one workgroup per CU, grid-stride over 64-chain tiles
    hipLaunchKernelGGL((k_leapfrog_dense_mfma<NCH>), dim3((unsigned)grid), dim3(256), bytes, st, s, eps, own);
    return hipGetLastError();
}

// returns hipErrorNotSupported when the shape is outside the MFMA kernel's range (L > 256)
hipError_t launch_leapfrog_dense_mfma(const DevState &s, double eps, int own, hipStream_t st)
{
    if (s.nch == 1) return launch_mfma_t<1>(s, eps, own, st);
    if (s.nch == 2) return launch_mfma_t<2>(s, eps, own, st);
    return hipErrorNotSupported;
}

}  // namespace idhmc
