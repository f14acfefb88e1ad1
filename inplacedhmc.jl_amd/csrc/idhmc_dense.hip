// idhmc_dense.hip -- dense multivariate-normal density (BASELINE.json configs[3]): ahead-of-time
// instantiation of the general-density kernels (idhmc_general.hpp) with DenseMvn (per-wave GEMV streaming the
// symmetric precision matrix from L2, idhmc_device.hpp).  The single-step leapfrog goes to the fp64 matrix-core
// kernel (idhmc_dense_mfma.hip) when the shape allows; the NUTS transition is k_nuts<.., DenseMvn, ..>.
#include "idhmc_general.hpp"
#include "idhmc_optimum.hpp"
#include <cstdlib>

namespace idhmc {

hipError_t launch_eval_dense(const DevState &s, hipStream_t st)
{
    IDHMC_DISPATCH_NCH_POW2(s.nch, hipLaunchKernelGGL((k_eval_general<NCH, DenseMvn<NCH>>), dim3(general_grid(s.C)),
                                                 dim3(kGeneralWaves * 64), 0, st, s, 0));
    return hipGetLastError();
}
hipError_t launch_random_position_dense(const DevState &s, hipStream_t st)
{
    IDHMC_DISPATCH_NCH_POW2(s.nch, hipLaunchKernelGGL((k_eval_general<NCH, DenseMvn<NCH>>), dim3(general_grid(s.C)),
                                                 dim3(kGeneralWaves * 64), 0, st, s, 1));
    return hipGetLastError();
}

hipError_t launch_leapfrog_dense_mfma(const DevState &s, double eps, int own, int n_steps, hipStream_t st);

hipError_t launch_leapfrog_dense(const DevState &s, double eps, int own, int n_steps, hipStream_t st)
{
    // the matrix-core kernel (L <= 512); IDHMC_DENSE_MFMA=0 selects the per-wave GEMV kernel
    const char *e = getenv("IDHMC_DENSE_MFMA");
    if (!(e && e[0] == '0')) {
        const hipError_t r = launch_leapfrog_dense_mfma(s, eps, own, n_steps, st);
        if (r != hipErrorNotSupported) return r;
    }
    IDHMC_DISPATCH_NCH_POW2(s.nch, hipLaunchKernelGGL((k_leapfrog_general<NCH, DenseMvn<NCH>>), dim3(general_grid(s.C)),
                                                 dim3(kGeneralWaves * 64), 0, st, s, eps, own, n_steps));
    return hipGetLastError();
}
hipError_t launch_local_optimum_dense(const DevState &s, double penalty, int iterations, hipStream_t st)
{
    IDHMC_DISPATCH_NCH_POW2(s.nch, hipLaunchKernelGGL((k_local_optimum_general<NCH, DenseMvn<NCH>>), dim3(optimum_grid(s)),
                                                 dim3(kOptimumWaves * 64), 0, st, s, penalty, iterations));
    return hipGetLastError();
}
hipError_t launch_stepsize_search_dense(const DevState &s, hipStream_t st)
{
    IDHMC_DISPATCH_NCH_POW2(s.nch, hipLaunchKernelGGL((k_stepsize_general<NCH, DenseMvn<NCH>>), dim3(general_grid(s.C)),
                                                 dim3(kGeneralWaves * 64), 0, st, s));
    return hipGetLastError();
}

}  // namespace idhmc
