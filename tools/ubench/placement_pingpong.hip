// placement_pingpong.hip -- would the sweep stream faster if it wrote to OTHER arrays than it reads (ping-pong state), with the classes of
// the arrays chosen?  (tools/ubench, GPU box)  Finds arrays of both classes by pair probes, then times the three-stream sweep pattern in place
// and out of place for several class assignments.  build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/placement_pingpong tools/ubench/placement_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
__global__ __launch_bounds__(256) void sweep(const double *r0, const double *r1, const double *r2, double *w0, double *w1, double *w2, int nvec, long long C, int L)
{
    const int lane = threadIdx.x & 63;
    const long long chain = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (chain >= C) return;
    const double *r[3] = {r0, r1, r2};
    double *w[3] = {w0, w1, w2};
    for (int j = 0; j < L / 128; ++j) {
        double2 x[3];
        for (int k = 0; k < nvec; ++k) x[k] = reinterpret_cast<const double2 *>(r[k] + chain * L)[j * 64 + lane];
        for (int k = 0; k < nvec; ++k) reinterpret_cast<double2 *>(w[k] + chain * L)[j * 64 + lane] = x[k];
    }
}
static double rate(const double *r0, const double *r1, const double *r2, double *w0, double *w1, double *w2, int nvec, long long C, int L)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(sweep, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, r0, r1, r2, w0, w1, w2, nvec, C, L);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(sweep, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, r0, r1, r2, w0, w1, w2, nvec, C, L);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 8.0 * nvec * 2.0 * C * L * 8 / (ms * 1e-3) / 1e9;
}
int main()
{
    const int L = 1024; const long long C = 65536; const size_t A = sizeof(double) * C * L;
    std::vector<double *> a, b;      // class of array 0, the other class
    double *ref = nullptr; (void)hipMalloc(&ref, A); (void)hipMemset(ref, 0, A); (void)hipDeviceSynchronize();
    a.push_back(ref);
    const double single = rate(ref, nullptr, nullptr, ref, nullptr, nullptr, 1, C, L);
    std::vector<void *> spacers;
    for (int i = 0; i < 60 && (a.size() < 4 || b.size() < 4); ++i) {
        if (i >= 3) { void *s = nullptr; if (hipMalloc(&s, (size_t)1 << 30) == hipSuccess) spacers.push_back(s); }
        double *x = nullptr; if (hipMalloc(&x, A) != hipSuccess) break;
        (void)hipMemset(x, 0, A); (void)hipDeviceSynchronize();
        const double r = rate(ref, x, nullptr, ref, x, nullptr, 2, C, L) / single;
        (r >= 1.10 ? b : a).push_back(x);
    }
    printf("one array in place: %.0f GB/s; found %zu arrays of its class, %zu of the other\n", single, a.size(), b.size());
    if (a.size() < 4 || b.size() < 3) { printf("not enough arrays of both classes\n"); return 0; }
    auto R = [&](const char *what, const double *r0, const double *r1, const double *r2, double *w0, double *w1, double *w2) {
        printf("  %-64s %.0f GB/s\n", what, rate(r0, r1, r2, w0, w1, w2, 3, C, L));
    };
    printf("three streams read and three written per chain (the sweep's pattern):\n");
    R("in place, classes A A A", a[0], a[1], a[2], a[0], a[1], a[2]);
    R("in place, classes A B A (what place_state finds)", a[0], b[0], a[1], a[0], b[0], a[1]);
    R("in place, classes A B B", a[0], b[0], b[1], a[0], b[0], b[1]);
    R("out of place, read A A A -> write B B B", a[0], a[1], a[2], b[0], b[1], b[2]);
    R("out of place, read A B A -> write B A B", a[0], b[0], a[1], b[1], a[2], b[2]);
    R("out of place, read A B A -> write A B A (other arrays)", a[0], b[0], a[1], a[2], b[1], a[3]);
    R("out of place, read A A B -> write B B A", a[0], a[1], b[0], b[1], b[2], a[2]);
    return 0;
}
