// placement_scramble.hip -- an array whose PHYSICAL chunks are mapped in a permuted order (hipMemCreate per chunk + hipMemMap): does it
// stream well together with an ordinary contiguous array?  (tools/ubench, GPU box)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/placement_scramble tools/ubench/placement_scramble.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <numeric>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
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
struct Scr { void *va = nullptr; size_t size = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
// an array of `bytes` made of chunks of `chunk` bytes, chunk i of the physical sequence mapped at slot perm[i]
static int make(Scr &s, size_t bytes, size_t chunk, const std::vector<int> &perm, int device)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    const int n = (int)(bytes / chunk);
    s.size = bytes;
    CK(hipMemAddressReserve(&s.va, bytes, 0, nullptr, 0));
    s.h.resize(n);
    for (int i = 0; i < n; ++i) CK(hipMemCreate(&s.h[i], chunk, &prop, 0));
    for (int i = 0; i < n; ++i) CK(hipMemMap((char *)s.va + (size_t)perm[i] * chunk, chunk, 0, s.h[i], 0));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(s.va, bytes, &acc, 1));
    CK(hipMemset(s.va, 0, bytes));
    CK(hipDeviceSynchronize());
    return 0;
}
static void drop(Scr &s)
{
    if (!s.va) return;
    (void)hipMemUnmap(s.va, s.size);
    for (auto &h : s.h) (void)hipMemRelease(h);
    (void)hipMemAddressFree(s.va, s.size);
    s = Scr();
}
int main()
{
    const int L = 1024; const long long C = 65536; const size_t A = sizeof(double) * C * L;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    double *a0 = nullptr, *a1 = nullptr; CK(hipMalloc(&a0, A)); CK(hipMalloc(&a1, A)); CK(hipMemset(a0, 0, A)); CK(hipMemset(a1, 0, A)); CK(hipDeviceSynchronize());
    double single = 0; for (int i = 0; i < 3; ++i) single = std::max(single, rate(a0, nullptr, nullptr, 1, C, L));
    printf("one hipMalloc array alone: %.0f GB/s; two hipMalloc arrays: %.3f\n", single, rate(a0, a1, nullptr, 2, C, L) / single);
    srand(7);
    for (size_t chunk : {(size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20}) {
        if (chunk % gmin) continue;
        const int n = (int)(A / chunk);
        std::vector<int> id(n), sh(n), rv(n), rot(n), sh2(n);
        std::iota(id.begin(), id.end(), 0);
        sh = id; for (int i = n - 1; i > 0; --i) std::swap(sh[i], sh[rand() % (i + 1)]);
        sh2 = id; for (int i = n - 1; i > 0; --i) std::swap(sh2[i], sh2[rand() % (i + 1)]);
        for (int i = 0; i < n; ++i) { rv[i] = n - 1 - i; rot[i] = (i + n / 2) % n; }
        const char *names[] = {"identity", "shuffled", "reversed", "rotated by half"};
        std::vector<int> *perms[] = {&id, &sh, &rv, &rot};
        for (int k = 0; k < 4; ++k) {
            Scr s; if (make(s, A, chunk, *perms[k], 0)) return 2;
            printf("chunks of %3zu MiB, %-16s alone %.3f   with a hipMalloc array %.3f\n", chunk >> 20, names[k], rate((double *)s.va, nullptr, nullptr, 1, C, L) / single,
                   rate(a0, (double *)s.va, nullptr, 2, C, L) / single);
            drop(s);
        }
        Scr s1, s2; if (make(s1, A, chunk, sh, 0) || make(s2, A, chunk, sh2, 0)) return 2;
        printf("chunks of %3zu MiB: hipMalloc + two differently shuffled arrays %.3f; the two shuffled ones %.3f\n", chunk >> 20,
               rate(a0, (double *)s1.va, (double *)s2.va, 3, C, L) / single, rate((double *)s1.va, (double *)s2.va, nullptr, 2, C, L) / single);
        drop(s1); drop(s2);
    }
    return 0;
}
