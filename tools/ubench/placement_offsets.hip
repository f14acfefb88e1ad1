// placement_offsets.hip -- inside ONE allocation: does the offset between two 512 MiB arrays decide whether they stream well together?
// (tools/ubench, GPU box)  pair (slab, slab + 512 MiB + delta) for a list of deltas; ratio of the pair's rate to one array alone.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/placement_offsets tools/ubench/placement_offsets.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
__global__ __launch_bounds__(256) void probe(double *a0, double *a1, double *a2, int nvec, long long C, int L)
{
    const int lane = threadIdx.x & 63;
    const long long chain = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (chain >= C) return;
    double *v[3] = {a0, a1, a2};
    for (int j = 0; j < L / 128; ++j) {
        double2 x[3];
        for (int k = 0; k < nvec; ++k) x[k] = reinterpret_cast<const double2 *>(v[k] + chain * L)[j * 64 + lane];
        for (int k = 0; k < nvec; ++k) reinterpret_cast<double2 *>(v[k] + chain * L)[j * 64 + lane] = x[k];
    }
}
static double rate(double *a, double *b, double *c, int nvec, long long C, int L)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(probe, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, a, b, c, nvec, C, L);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, a, b, c, nvec, C, L);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 5.0 * nvec * 2.0 * C * L * 8 / (ms * 1e-3) / 1e9;
}
int main()
{
    const int L = 1024; const long long C = 65536; const size_t A = sizeof(double) * C * L;     // 512 MiB
    const size_t slack = (size_t)600 << 20;
    char *slab = nullptr;
    if (hipMalloc((void **)&slab, 3 * A + 2 * slack) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(slab, 0, 3 * A + 2 * slack); hipDeviceSynchronize();
    double *q = (double *)slab;
    double single = 0; for (int i = 0; i < 3; ++i) single = std::max(single, rate(q, nullptr, nullptr, 1, C, L));
    printf("slab at %p; one array alone: %.0f GB/s\n", (void *)slab, single);
    const size_t KB = 1024, MB = 1024 * 1024;
    const size_t deltas[] = {0, 4 * KB, 8 * KB, 16 * KB, 32 * KB, 64 * KB, 128 * KB, 256 * KB, 512 * KB, 1 * MB, 2 * MB, 3 * MB, 4 * MB, 6 * MB, 8 * MB, 12 * MB, 16 * MB, 24 * MB,
                             32 * MB, 48 * MB, 64 * MB, 96 * MB, 128 * MB, 192 * MB, 256 * MB, 384 * MB, 511 * MB, 1 * MB + 64 * KB, 2 * MB + 4 * KB, 33 * MB, 100 * MB + 36 * KB};
    printf("pair (slab, slab + 512 MiB + delta):\n");
    for (size_t d : deltas) printf("  delta %9.3f MiB: %.3f\n", d / 1048576.0, rate(q, (double *)(slab + A + d), nullptr, 2, C, L) / single);
    printf("triple (slab, +512 MiB + d, +1024 MiB + 2 d + slack-ish):\n");
    for (size_t d : {(size_t)0, 2 * MB, 16 * MB, 64 * MB, 128 * MB, 256 * MB}) printf("  d %9.3f MiB: %.3f\n", d / 1048576.0, rate(q, (double *)(slab + A + d), (double *)(slab + 2 * A + 2 * d), 3, C, L) / single);
    return 0;
}
