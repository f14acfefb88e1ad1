// micro-calibration on MI355X: fp64 VALU issue, wave reduction, LDS read latency (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../inplacedhmc.jl_amd/csrc/idhmc_math.hpp"
using namespace idhmc;
__global__ void k(double *out, long long *cyc, int iters)
{
    __shared__ double lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + lane * 1e-3 + i;
    // 1) 16 independent fma chains
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], 0.999999, 1e-7);
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    // 2) one dependent chain
    double b = s * 1e-9 + 1.0;
    long long t2 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) b = __builtin_fma(b, 0.999999, 1e-7);
    }
    long long t3 = clock64();
    // 3) wave_sum2
    double x = b, y = s;
    long long t4 = clock64();
    for (int it = 0; it < iters; ++it) { double sa, sb; wave_sum2(x, y, y, x, sa, sb); x = sa * 1e-3; y = sb * 1e-3; }
    long long t5 = clock64();
    // 4) LDS read b128 dependent-ish: 8 reads then use
    const double2 *p = reinterpret_cast<const double2 *>(lds) + lane;
    double acc = x + y;
    long long t6 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { double2 v = p[j * 64 + (it & 1)]; acc = __builtin_fma(v.x, v.y, acc); }
    }
    long long t7 = clock64();
    // 5) division + dlog
    double d = acc * 1e-6 + 2.0;
    long long t8 = clock64();
    for (int it = 0; it < iters; ++it) d = dlog(d + 2.0) + 3.0;
    long long t9 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d + acc;
    if (lane == 0) {
        long long *c = cyc + ((blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6)) * 5;
        c[0] = t1 - t0; c[1] = t3 - t2; c[2] = t5 - t4; c[3] = t7 - t6; c[4] = t9 - t8;
    }
}
int main()
{
    const int iters = 1000;
    for (int threads : {64, 256, 512}) {
        double *out; long long *cyc;
        hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 5 * 8);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        long long h[5]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("threads/block %d (waves/SIMD %.1f): per-iter cycles: 16 indep fma %.1f (%.2f/fma) | 16 dep fma %.1f (%.2f/fma) | wave_sum2 %.1f | 8x ds_read_b128+fma %.1f | dlog %.1f\n",
               threads, threads / 256.0, h[0] / (double)iters, h[0] / 16.0 / iters, h[1] / (double)iters, h[1] / 16.0 / iters,
               h[2] / (double)iters, h[3] / (double)iters, h[4] / (double)iters);
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
