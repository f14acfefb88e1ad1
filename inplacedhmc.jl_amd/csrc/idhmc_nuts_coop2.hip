// idhmc_nuts_coop2.hip -- instantiation and launch of k_nuts_coop2 (idhmc_nuts_coop2.hpp): dense MVN, L <= 256, shared metric,
// two chains per wavefront.
#include "idhmc_nuts_coop2.hpp"
#include <cstdlib>

namespace idhmc {

// work in progress: IDHMC_DENSE_COOP2=1 selects it; the default is the one-chain-per-wavefront form (k_nuts<DenseMvnCoop>)
bool dense_coop2(int nch, int model, int shared_metric)
{
    static const bool off = [] { const char *e = getenv("IDHMC_DENSE_COOP2"); return !(e && e[0] == '1'); }();
    static const bool coop_off = [] { const char *e = getenv("IDHMC_DENSE_COOP"); return e && e[0] == '0'; }();
    return model == IDHMC_MODEL_DENSE_MVN && nch <= 2 && shared_metric != 0 && !off && !coop_off;
}
int coop2_arena_vectors_host(int max_depth) { return coop2_arena_vectors(max_depth); }

template <int NCH>
static hipError_t launch_t(const DevState &s, uint32_t iter, uint32_t flags, int grid, hipStream_t st)
{
    const size_t bytes = sizeof(double) * Coop2Shape<NCH>::lds_doubles();
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_coop2<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    hipLaunchKernelGGL((k_nuts_coop2<NCH>), dim3(grid), dim3(1024), bytes, st, s, iter, flags);
    return hipGetLastError();
}

// one workgroup hosts 32 chains at a time (16 wavefronts x 2 contexts), each with an arena slot of its own
hipError_t launch_nuts_coop2(const DevState &s, uint32_t iter, uint32_t flags, hipStream_t st)
{
    if ((int64_t)coop2_arena_vectors(s.max_depth) * s.L > s.arena_stride) return hipErrorInvalidValue;
    const int64_t need = (s.C + 31) / 32, have = s.nslots / 32;
    const int grid = (int)(need < have ? need : have);
    if (grid < 1) return hipErrorInvalidValue;
    if (s.nch == 1) return launch_t<1>(s, iter, flags, grid, st);
    if (s.nch == 2) return launch_t<2>(s, iter, flags, grid, st);
    return hipErrorInvalidValue;
}

}  // namespace idhmc
