// idhmc_nuts.hip -- ahead-of-time instantiation and launch of the NUTS transition kernel and the separable
// initial-stepsize search (templates in idhmc_nuts_kernel.hpp) for the built-in densities.
#include "idhmc_nuts_kernel.hpp"
#include "idhmc_optimum.hpp"
#include <cstdlib>

namespace idhmc {

// the tree arena also serves as the L-BFGS history of the FindLocalOptimum stage (2 * kLbfgsR vectors)
int arena_vectors(int max_depth, int model, int L)
{
    const bool separable = model == IDHMC_MODEL_ISO_GAUSSIAN || model == IDHMC_MODEL_DIAG_GAUSSIAN;
    const int n = ArenaMap{max_depth, nuts_regenerate(separable), nuts_defer(separable) ? nuts_dl_vectors(max_depth, L) : 0}.count();
    return n > 2 * kLbfgsR ? n : 2 * kLbfgsR;
}
// the dense MVN runs the workgroup-cooperative matrix-core gradient (DenseMvnCoop) when one 16-column tile per
// wavefront covers the matrix (L <= 256); IDHMC_DENSE_COOP=0 selects the per-wave GEMV (experiments)
static bool dense_coop(int nch)
{
    static const bool off = [] { const char *e = getenv("IDHMC_DENSE_COOP"); return e && e[0] == '0'; }();
    return nch <= 2 && !off;
}
int nuts_waves_per_block(int nch, int model, int shared_metric)
{
    return nuts_waves(nch, model == IDHMC_MODEL_ISO_GAUSSIAN || model == IDHMC_MODEL_DIAG_GAUSSIAN,
                      model == IDHMC_MODEL_DENSE_MVN && dense_coop(nch), shared_metric != 0);
}
// wavefronts per workgroup of the wide form of the kernel (0: the model/shape has none); the arena is sized for it
int nuts_wide_waves_per_block(int nch, int model)
{
    return nuts_wide_waves(nch, model == IDHMC_MODEL_ISO_GAUSSIAN || model == IDHMC_MODEL_DIAG_GAUSSIAN, false);
}
size_t nuts_lds_bytes(int L, bool lds_params, bool shared_metric, bool separable)
{
    return sizeof(double) * nuts_lds_doubles(L, lds_params, shared_metric, separable) ;
}

hipError_t launch_nuts_sep_from1(const DevState &s, uint32_t iter, uint32_t flags, int wide, int grid, hipStream_t st);
hipError_t launch_nuts_sep_from5(const DevState &s, uint32_t iter, uint32_t flags, int wide, int grid, hipStream_t st);
hipError_t launch_nuts_sep_from9(const DevState &s, uint32_t iter, uint32_t flags, int wide, int grid, hipStream_t st);
hipError_t launch_nuts_sep_from13(const DevState &s, uint32_t iter, uint32_t flags, int wide, int grid, hipStream_t st);
hipError_t launch_stepsize_search_dense(const DevState &s, hipStream_t st);
hipError_t launch_nuts_jit(const DevState &s, uint32_t iter, uint32_t flags, int grid, hipStream_t st);
hipError_t launch_stepsize_search_jit(const DevState &s, hipStream_t st);

template <int NCH, class Model, bool SHARED, int WAVES = nuts_waves(NCH, Model::kSeparable, Model::kCooperative, SHARED)>
static hipError_t launch_nuts_t(const DevState &s, uint32_t iter, uint32_t flags, int grid, hipStream_t st)
{
    const size_t bytes = sizeof(double) * nuts_lds_doubles(128 * NCH, Model::kHasParams && Model::kSeparable, SHARED,
                                                           Model::kSeparable, Model::kCooperative, WAVES);
    static bool attr_done[64] = {};  // per instantiation and device (the attribute is per device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts<NCH, Model, SHARED, WAVES>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    hipLaunchKernelGGL((k_nuts<NCH, Model, SHARED, WAVES>), dim3(grid), dim3(WAVES * 64),
                       bytes, st,
                       s, iter, flags);
    return hipGetLastError();
}

// wide != 0 selects the wide form of the kernel where one exists (same arithmetic, same results)
hipError_t launch_nuts(const DevState &s0, uint32_t iter, uint32_t flags, int wide, hipStream_t st, uint32_t n_iter, double *fz_q, idhmc_tree_stats *fz_st)
{
    if (s0.max_depth < 1 || s0.max_depth > kMaxDepth - 1) return hipErrorInvalidValue;
    if (n_iter < 1 || (uint64_t)s0.C * n_iter >= (1ull << 31)) return hipErrorInvalidValue;
    if (n_iter > 1 && (!s0.iters_done || (flags & IDHMC_T_USE_DIRECTIONS))) return hipErrorInvalidValue;
    DevState s = s0;
    s.n_iter = n_iter;
    s.fz_q = fz_q;
    s.fz_st = fz_st;
    hipError_t e = hipMemsetAsync(s.queue, 0, sizeof(uint32_t) * 16, st);       // 8 range queues, 8 XCD ids (idhmc_nuts_kernel.hpp)
    if (e != hipSuccess) return e;
    if (n_iter > 1) {
        e = hipMemsetAsync(s.iters_done, 0, sizeof(uint32_t) * (size_t)s.C, st);
        if (e != hipSuccess) return e;
    }
    const int WW = nuts_wide_waves_per_block(s.nch, s.model);
    wide = wide && WW > 0;
    const int W = wide ? WW : nuts_waves_per_block(s.nch, s.model, s.minv_stride == 0);
    int64_t need = (s.C + W - 1) / W;
    const int64_t have = s.nslots / W;
    const int grid = (int)(need < have ? need : have);
    const bool shared = s.minv_stride == 0;
    if (s.model == IDHMC_MODEL_CUSTOM) return launch_nuts_jit(s, iter, flags, grid, st);
    if (s.model == IDHMC_MODEL_DENSE_MVN) {
        IDHMC_DISPATCH_NCH_POW2(s.nch, {
            if constexpr (NCH <= 2) {
                if (dense_coop(NCH))
                    return shared ? launch_nuts_t<NCH, DenseMvnCoop<NCH>, true>(s, iter, flags, grid, st)
                                  : launch_nuts_t<NCH, DenseMvnCoop<NCH>, false>(s, iter, flags, grid, st);
            }
            return shared ? launch_nuts_t<NCH, DenseMvn<NCH>, true>(s, iter, flags, grid, st)
                          : launch_nuts_t<NCH, DenseMvn<NCH>, false>(s, iter, flags, grid, st);
        });
        return hipErrorInvalidValue;
    }
    // separable densities: one translation unit per four padded lengths (idhmc_nuts_sep.inc)
    if (s.nch <= 4) return launch_nuts_sep_from1(s, iter, flags, wide, grid, st);
    if (s.nch <= 8) return launch_nuts_sep_from5(s, iter, flags, wide, grid, st);
    if (s.nch <= 12) return launch_nuts_sep_from9(s, iter, flags, wide, grid, st);
    return launch_nuts_sep_from13(s, iter, flags, wide, grid, st);
}

hipError_t launch_local_optimum_dense(const DevState &s, double penalty, int iterations, hipStream_t st);
hipError_t launch_local_optimum_jit(const DevState &s, double penalty, int iterations, int grid, hipStream_t st);

// FindLocalOptimum (src/warmup.jl:137-187), idhmc_optimum.hpp
hipError_t launch_local_optimum(const DevState &s, double penalty, int iterations, hipStream_t st)
{
    if (s.model == IDHMC_MODEL_DENSE_MVN) return launch_local_optimum_dense(s, penalty, iterations, st);
    if (s.model == IDHMC_MODEL_CUSTOM) return launch_local_optimum_jit(s, penalty, iterations, optimum_grid(s), st);
    IDHMC_DISPATCH_NCH(s.nch, {
        if (s.model == IDHMC_MODEL_ISO_GAUSSIAN)
            hipLaunchKernelGGL((k_local_optimum<NCH, IsoGaussian<NCH>>), dim3(optimum_grid(s)), dim3(kOptimumWaves * 64),
                               0, st, s, penalty, iterations);
        else
            hipLaunchKernelGGL((k_local_optimum<NCH, DiagGaussian<NCH>>), dim3(optimum_grid(s)), dim3(kOptimumWaves * 64),
                               0, st, s, penalty, iterations);
    });
    return hipGetLastError();
}

hipError_t launch_stepsize_search(const DevState &s, hipStream_t st)
{
    if (s.model == IDHMC_MODEL_DENSE_MVN) return launch_stepsize_search_dense(s, st);
    if (s.model == IDHMC_MODEL_CUSTOM) return launch_stepsize_search_jit(s, st);
    int64_t b = (s.C + 3) / 4;
    if (b > 4096) b = 4096;
    const int grid = (int)b;
    IDHMC_DISPATCH_NCH(s.nch, {
        if (s.model == IDHMC_MODEL_ISO_GAUSSIAN)
            hipLaunchKernelGGL((k_stepsize_search<NCH, IsoGaussian<NCH>>), dim3(grid), dim3(256), 0, st, s);
        else
            hipLaunchKernelGGL((k_stepsize_search<NCH, DiagGaussian<NCH>>), dim3(grid), dim3(256), 0, st, s);
    });
    return hipGetLastError();
}

}  // namespace idhmc
