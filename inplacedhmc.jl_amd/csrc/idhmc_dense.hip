// idhmc_dense.hip -- dense multivariate-normal density (Sigma^-1 x gradient); placeholder launchers.
#include "idhmc_device.hpp"
#include "idhmc_internal.hpp"
namespace idhmc {
hipError_t launch_eval_dense(const DevState &, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_leapfrog_dense(const DevState &, double, int, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_random_position_dense(const DevState &, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_nuts_dense(const DevState &, uint32_t, uint32_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_stepsize_search_dense(const DevState &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace idhmc
