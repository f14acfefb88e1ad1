// idhmc_xchg.hpp -- the fixed-point record of the global-stepsize exchange (include/idhmc.h, "the global-stepsize
// exchange"): host and device share one definition, the oracle restates it (oracle/idhmc_oracle.c, orc_xchg_*).
// A value x becomes the integer v = rint(x * 2^S), split into hi = v >> B (arithmetic) and lo = v & (2^B - 1); limb sums
// are integers below 2^53, hence exact in doubles whatever the order of a SUM all-reduce.
#pragma once
#include "../../include/idhmc.h"

namespace idhmc {

__host__ __device__ constexpr int xchg_limb_bits(int kind) { return kind == IDHMC_XCHG_ACCEPT ? 26 : 25; }
__host__ __device__ constexpr double xchg_scale(int kind) { return kind == IDHMC_XCHG_ACCEPT ? 0x1p52 : 0x1p40; }
__host__ __device__ constexpr double xchg_inv_scale(int kind) { return kind == IDHMC_XCHG_ACCEPT ? 0x1p-52 : 0x1p-40; }
__host__ __device__ constexpr double xchg_limb_scale(int kind) { return kind == IDHMC_XCHG_ACCEPT ? 0x1p26 : 0x1p25; }

// an acceptance rate lies in [0, 1] (NaN counts as 0); log eps is clamped to [-1024, 1024] (NaN counts as 0)
__host__ __device__ inline double xchg_clamp(int kind, double x)
{
    if (kind == IDHMC_XCHG_ACCEPT) return !(x >= 0.0) ? 0.0 : (x > 1.0 ? 1.0 : x);
    return !(x == x) ? 0.0 : (x < -1024.0 ? -1024.0 : (x > 1024.0 ? 1024.0 : x));
}
__host__ __device__ inline void xchg_limbs(int kind, double x, long long &hi, long long &lo)
{
    const long long v = (long long)__builtin_rint(xchg_clamp(kind, x) * xchg_scale(kind));
    const int B = xchg_limb_bits(kind);
    hi = v >> B;
    lo = v & ((1ll << B) - 1);
}
// the pooled mean from the (all-reduced) totals; -ffp-contract=off: a multiply, an add, a multiply, a divide
__host__ __device__ inline double xchg_mean(int kind, double sum_hi, double sum_lo, double count)
{
    const double v = sum_hi * xchg_limb_scale(kind) + sum_lo;
    return (v * xchg_inv_scale(kind)) / count;
}

}  // namespace idhmc
