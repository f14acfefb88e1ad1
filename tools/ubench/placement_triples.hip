// placement_triples.hip -- which TRIPLES of 512 MiB arrays stream well together?  (tools/ubench, GPU box)
// Allocates M arrays, times the single-step sweep's access pattern (one chain of L doubles per wavefront, 16 bytes per lane, every array
// read and written in place, element i of all of them at the same time) on one array alone and on triples: consecutive allocations
// (what place_state tries) and random ones.  build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/placement_triples tools/ubench/placement_triples.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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
int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 24, L = 1024; const long long C = 65536;
    std::vector<double *> a(M);
    for (int i = 0; i < M; ++i) { if (hipMalloc(&a[i], sizeof(double) * C * L) != hipSuccess) { printf("alloc %d failed\n", i); return 1; } hipMemset(a[i], 0, sizeof(double) * C * L); }
    hipDeviceSynchronize();
    printf("arrays at:"); for (int i = 0; i < M; ++i) printf(" %p", (void *)a[i]); printf("\n");
    double single = 0; for (int i = 0; i < 3; ++i) single = std::max(single, rate(a[i], nullptr, nullptr, 1, C, L));
    printf("one array alone: %.0f GB/s\n", single);
    printf("pairs (i, i+1):"); for (int i = 0; i + 1 < M; i += 2) printf(" %.3f", rate(a[i], a[i + 1], nullptr, 2, C, L) / single); printf("\n");
    printf("consecutive triples (i, i+1, i+2), ratio to one array alone:\n ");
    int good = 0, n = 0;
    for (int i = 0; i + 2 < M; ++i) { const double r = rate(a[i], a[i + 1], a[i + 2], 3, C, L) / single; printf(" %.3f", r); good += r >= 1.10; ++n; }
    printf("\n  good (>= 1.10): %d of %d\n", good, n);
    printf("random triples:\n ");
    srand(12345); good = 0; n = 0;
    for (int t = 0; t < 60; ++t) {
        int i = rand() % M, j = rand() % M, k = rand() % M;
        if (i == j || j == k || i == k) continue;
        const double r = rate(a[i], a[j], a[k], 3, C, L) / single; printf(" %.3f", r); good += r >= 1.10; ++n;
    }
    printf("\n  good (>= 1.10): %d of %d\n", good, n);
    // does goodness decompose into pairs?  all pairs among the first 8 arrays
    printf("pair matrix of arrays 0..7 (ratio of the pair's rate to one array alone):\n");
    for (int i = 0; i < 8 && i < M; ++i) { printf("  "); for (int j = 0; j < 8 && j < M; ++j) printf(" %5.3f", i == j ? 1.0 : rate(a[i], a[j], nullptr, 2, C, L) / single); printf("\n"); }
    return 0;
}
