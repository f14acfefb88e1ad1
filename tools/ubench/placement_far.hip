// placement_far.hip -- how far apart (in allocation order) must two 512 MiB arrays be to stream well together?  (tools/ubench, GPU box)
// array 0, then repeatedly: a spacer of S GiB (kept), another array, pair rate (array 0, new array) / one array alone.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/placement_far tools/ubench/placement_far.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
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
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(probe, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, a, b, c, nvec, C, L);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, a, b, c, nvec, C, L);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 5.0 * nvec * 2.0 * C * L * 8 / (ms * 1e-3) / 1e9;
}
int main(int argc, char **argv)
{
    const int L = 1024; const long long C = 65536; const size_t A = sizeof(double) * C * L;
    const double S = argc > 1 ? atof(argv[1]) : 4.0;     // spacer GiB
    const int N = argc > 2 ? atoi(argv[2]) : 24;
    std::vector<double *> arr;
    std::vector<char> good;
    double *a0 = nullptr; (void)hipMalloc(&a0, A); (void)hipMemset(a0, 0, A); (void)hipDeviceSynchronize();
    double single = 0; for (int i = 0; i < 3; ++i) single = std::max(single, rate(a0, nullptr, nullptr, 1, C, L));
    printf("array 0 at %p; one array alone: %.0f GB/s; spacer %.1f GiB\n", (void *)a0, single, S);
    double held = 0.5;
    for (int i = 0; i < N; ++i) {
        void *sp = nullptr;
        if (S > 0 && hipMalloc(&sp, (size_t)(S * 1073741824.0)) != hipSuccess) { printf("spacer alloc failed at %d\n", i); break; }
        double *x = nullptr;
        if (hipMalloc(&x, A) != hipSuccess) { printf("array alloc failed at %d\n", i); break; }
        (void)hipMemset(x, 0, A); (void)hipDeviceSynchronize();
        held += S + 0.5;
        const double r = rate(a0, x, nullptr, 2, C, L) / single;
        printf("  after %6.1f GiB allocated: array at %p  alone %.0f GB/s  pair with array 0: %.3f%s\n", held, (void *)x, rate(x, nullptr, nullptr, 1, C, L), r, r >= 1.10 ? "  <-- good" : "");
        arr.push_back(x);
        good.push_back(r >= 1.10);
    }
    // are the good ones good with each other?  and with a bad one other than array 0?
    std::vector<int> gi, bi;
    for (size_t i = 0; i < arr.size(); ++i) (good[i] ? gi : bi).push_back((int)i);
    if (gi.size() >= 2) {
        printf("pairs among the good ones:");
        for (size_t a = 0; a + 1 < gi.size() && a < 6; ++a) printf(" (%d,%d) %.3f", gi[a], gi[a + 1], rate(arr[gi[a]], arr[gi[a + 1]], nullptr, 2, C, L) / single);
        printf("\n");
    }
    if (!gi.empty() && !bi.empty()) {
        printf("good one with other bad ones:");
        for (size_t b = 0; b < bi.size() && b < 8; ++b) printf(" (%d,%d) %.3f", gi[0], bi[b], rate(arr[gi[0]], arr[bi[b]], nullptr, 2, C, L) / single);
        printf("\n");
    }
    return 0;
}
