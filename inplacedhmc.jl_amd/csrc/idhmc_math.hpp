// idhmc_math.hpp -- device-side deterministic fp64 math, Philox RNG and wavefront reductions.
//
// Everything here is built from IEEE-754 fp64 add / mul / div / sqrt / fma / rint only, so the
// CDNA4 result is bit-identical to a host evaluation of the same operation sequence.  That is
// what makes "same seed => same chain" checkable against a CPU run: the reference draws from
// VectorizedRNG.jl (reference src/rng.jl:8, src/kinetic_energy.jl:63, src/NUTS.jl:33,
// src/tree.jl:144-145), whose streams are not reproducible outside Julia, so this engine defines
// its own counter-based streams (Philox-4x32-10, Salmon et al. SC'11).
//
// Compile with -ffp-contract=off: every fused multiply-add below is explicit.
#pragma once
#ifndef __HIPCC_RTC__   // hipRTC (custom densities) brings its own runtime declarations
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
using __hip_internal::int32_t;
using __hip_internal::uint32_t;
using __hip_internal::int64_t;
using __hip_internal::uint64_t;
#endif

namespace idhmc {

#define IDHMC_DEV __device__ __forceinline__

IDHMC_DEV uint64_t d2u(double x) { return (uint64_t)__double_as_longlong(x); }
IDHMC_DEV double u2d(uint64_t b) { return __longlong_as_double((long long)b); }
IDHMC_DEV double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
IDHMC_DEV bool dfinite(double x) { return __builtin_isfinite(x); }
// fma(a, b, c) with c a compile-time constant: the same IEEE operation, but the constant is handed to v_fma_f64 in a scalar register
// pair.  Left to itself the compiler evaluates a Horner step p = fma(p, z, c) as "move c into a VGPR pair, then accumulate into it"
// (v_mov_b32 x 2 + v_fmac_f64): three vector instructions instead of one -- a fifth of the instructions of the kernel's log / exp /
// sin / cos polynomials (momentum refresh: 27 such steps per normal pair; log-sum-exp: 24 per merge).
IDHMC_DEV double dfma_c(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

constexpr double kLn2Hi = 6.93147180369123816490e-01;
constexpr double kLn2Lo = 1.90821492927058770002e-10;
constexpr double kInvLn2 = 1.44269504088896338700e+00;
constexpr double kHalfPi = 1.57079632679489661923;
constexpr double kInf = __builtin_huge_val();

// ln x: x = 2^e m with m in (sqrt(1/2), sqrt(2)];  ln m = 2 atanh((m-1)/(m+1)), series to s^23
IDHMC_DEV double dlog(double x)
{
    if (x != x) return x;
    if (x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -kInf;
    if (x == kInf) return x;
    uint64_t b = d2u(x);
    int e = 0;
    if (b < 0x0010000000000000ull) { x *= 0x1p54; b = d2u(x); e = -54; }
    e += (int)(b >> 52) - 1023;
    double m = u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (m + 1.0);
    const double z = s * s;
    double P = 1.0 / 23.0;
    P = dfma_c(P, z, 1.0 / 21.0);
    P = dfma_c(P, z, 1.0 / 19.0);
    P = dfma_c(P, z, 1.0 / 17.0);
    P = dfma_c(P, z, 1.0 / 15.0);
    P = dfma_c(P, z, 1.0 / 13.0);
    P = dfma_c(P, z, 1.0 / 11.0);
    P = dfma_c(P, z, 1.0 / 9.0);
    P = dfma_c(P, z, 1.0 / 7.0);
    P = dfma_c(P, z, 1.0 / 5.0);
    P = dfma_c(P, z, 1.0 / 3.0);
    P = dfma_c(P, z, 1.0);
    const double r = (s + s) * P;
    const double de = (double)e;
    return dfma(de, kLn2Hi, dfma(de, kLn2Lo, r));
}

// e^x: x = k ln2 + r, Taylor to r^13, exact power-of-two scale
IDHMC_DEV double dexp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return kInf;
    if (x < -708.3964185322641) return 0.0;
    const double k = __builtin_rint(x * kInvLn2);
    double r = dfma(-k, kLn2Hi, x);
    r = dfma(-k, kLn2Lo, r);
    double p = 1.0 / 6227020800.0;
    p = dfma_c(p, r, 1.0 / 479001600.0);
    p = dfma_c(p, r, 1.0 / 39916800.0);
    p = dfma_c(p, r, 1.0 / 3628800.0);
    p = dfma_c(p, r, 1.0 / 362880.0);
    p = dfma_c(p, r, 1.0 / 40320.0);
    p = dfma_c(p, r, 1.0 / 5040.0);
    p = dfma_c(p, r, 1.0 / 720.0);
    p = dfma_c(p, r, 1.0 / 120.0);
    p = dfma_c(p, r, 1.0 / 24.0);
    p = dfma_c(p, r, 1.0 / 6.0);
    p = dfma_c(p, r, 0.5);
    p = dfma_c(p, r, 1.0);
    p = dfma_c(p, r, 1.0);
    int ki = (int)k;
    if (ki > 1023) { p *= 2.0; ki -= 1; }
    return p * u2d((uint64_t)(ki + 1023) << 52);
}

IDHMC_DEV double dlog1p(double x)
{
    const double u = 1.0 + x;
    if (u == 1.0) return x;
    if (u == kInf) return u;
    const double d = u - 1.0;
    return dlog(u) * (x / d);
}

// (sin, cos)(2 pi u), u in [0,1): quadrant by rint(4u), exact remainder, Taylor to x^17 / x^16
IDHMC_DEV void dsincos2pi(double u, double &sn, double &cs)
{
    const double t = 4.0 * u;
    const double n = __builtin_rint(t);
    const double x = (t - n) * kHalfPi;
    const double z = x * x;
    double S = 1.0 / 355687428096000.0;
    S = dfma_c(S, z, -1.0 / 1307674368000.0);
    S = dfma_c(S, z, 1.0 / 6227020800.0);
    S = dfma_c(S, z, -1.0 / 39916800.0);
    S = dfma_c(S, z, 1.0 / 362880.0);
    S = dfma_c(S, z, -1.0 / 5040.0);
    S = dfma_c(S, z, 1.0 / 120.0);
    S = dfma_c(S, z, -1.0 / 6.0);
    S = dfma_c(S, z, 1.0);
    S = S * x;
    double C = 1.0 / 20922789888000.0;
    C = dfma_c(C, z, -1.0 / 87178291200.0);
    C = dfma_c(C, z, 1.0 / 479001600.0);
    C = dfma_c(C, z, -1.0 / 3628800.0);
    C = dfma_c(C, z, 1.0 / 40320.0);
    C = dfma_c(C, z, -1.0 / 720.0);
    C = dfma_c(C, z, 1.0 / 24.0);
    C = dfma_c(C, z, -0.5);
    C = dfma_c(C, z, 1.0);
    const int q = ((int)n) & 3;
    sn = (q == 0) ? S : (q == 1) ? C : (q == 2) ? -S : -C;
    cs = (q == 0) ? C : (q == 1) ? -S : (q == 2) ? -C : S;
}

// logaddexp, reference src/InplaceDHMC.jl:27-30
IDHMC_DEV double dlogaddexp(double x, double y)
{
    if (!(dfinite(x) && dfinite(y))) return x > y ? x : y;
    return x > y ? x + dlog1p(dexp(y - x)) : y + dlog1p(dexp(x - y));
}

// ---- Philox-4x32-10 ---------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

IDHMC_DEV u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: the 32-bit integer multiplies are
        // quarter-rate instructions, and a momentum refresh is 8 Philox calls per lane
        const uint64_t m0 = (uint64_t)0xD2511F53u * (uint64_t)c0, m1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t h0 = (uint32_t)(m0 >> 32), l0 = (uint32_t)m0;
        const uint32_t h1 = (uint32_t)(m1 >> 32), l1 = (uint32_t)m1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// RNG address: key = seed; counter = (index, transition number, global chain id, stream)
enum : uint32_t { kStreamDir = 0, kStreamMomentum = 1, kStreamExp = 2, kStreamInitQ = 3 };

struct RngKey { uint32_t k0, k1, chain; };

IDHMC_DEV u32x4 rng_draw(const RngKey &k, uint32_t iter, uint32_t stream, uint32_t idx)
{
    return philox4x32_10(idx, iter, k.chain, stream, k.k0, k.k1);
}
IDHMC_DEV double u01_open0(uint32_t lo, uint32_t hi)  // (0,1]
{
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)(v + 1) * 0x1p-53;
}
IDHMC_DEV double u01(uint32_t lo, uint32_t hi)        // [0,1)
{
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)v * 0x1p-53;
}
// two N(0,1) for element pair `pair` (Box-Muller)
IDHMC_DEV void randn_pair(const RngKey &k, uint32_t iter, uint32_t pair, double &n0, double &n1)
{
    const u32x4 x = rng_draw(k, iter, kStreamMomentum, pair);
    const double u1 = u01_open0(x.x, x.y);
    const double u2 = u01(x.z, x.w);
    const double r = __builtin_sqrt(-2.0 * dlog(u1));
    double s, c;
    dsincos2pi(u2, s, c);
    n0 = r * c;
    n1 = r * s;
}
IDHMC_DEV double randexp(const RngKey &k, uint32_t iter, uint32_t draw)
{
    const u32x4 x = rng_draw(k, iter, kStreamExp, draw);
    return -dlog(u01_open0(x.x, x.y));
}
IDHMC_DEV uint32_t rand_directions(const RngKey &k, uint32_t iter)
{
    return rng_draw(k, iter, kStreamDir, 0).x;
}

// ---- wavefront reduction in the canonical order -------------------------------------------------
// A length-L sum (L = 128*NCH) is defined as: 128 stride-128 fma chains (lane l owns residues 2l and
// 2l+1), then an adjacent pairwise tree.  The tree over lanes is a 6-level xor butterfly; every lane
// ends with the same bits.
// Data movement for the butterfly without LDS round trips (ds_bpermute costs ~90 cycles a level):
//   xor 1, 2   DPP quad_perm;
//   xor 4, 8   DPP row_half_mirror / row_mirror -- after the previous levels all lanes of a quad (an octet)
//              already hold the same partial sum, so "the mirrored lane" is "the lane with that bit flipped";
//   xor 16, 32 gfx950's lane-swap instructions (rows_sum below; rounds 1-2 read the four row sums back with v_readlane).
// The additions are the same pairs in the same order as the plain xor butterfly, so the bits are unchanged.
template <int CTRL>
IDHMC_DEV double dpp_mov(double v)
{
    // bound_ctrl = 1: a lane whose source is disabled reads 0 -- what the `old = 0` operand said already, but now the compiler does not
    // have to materialise that zero in the destination first (two v_mov_b32 per double and step: 32 instructions per leaf and merge)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
IDHMC_DEV double read_lane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// The last two levels (xor 16, xor 32) on gfx950's lane-swap instructions: v_permlane16_swap exchanges the odd 16-lane rows of one
// register with the even rows of another, v_permlane32_swap the upper half of one with the lower half of another.  With both
// operands copies of s: A = [r0 r0 r2 r2], B = [r1 r1 r3 r3], A + B = r0 + r1 | r2 + r3 in every lane of the pair of rows; then
// A = [t01 t01], B = [t23 t23], A + B = (r0 + r1) + (r2 + r3) in every lane -- the same two additions with the same operands in the
// same order as the v_readlane form (four scalar reads, moves back into vector registers, three additions), in 10 instead of 15.
IDHMC_DEV double rows_sum(double s)
{
    {
        const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(s), __double2loint(s), false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(s), __double2hiint(s), false, false);
        s = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
    }
    const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(s), __double2loint(s), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(s), __double2hiint(s), false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
IDHMC_DEV double wave_sum(double a0, double a1)
{
    double s = a0 + a1;
    s = s + dpp_mov<0xB1>(s);    // quad_perm [1,0,3,2]
    s = s + dpp_mov<0x4E>(s);    // quad_perm [2,3,0,1]
    s = s + dpp_mov<0x141>(s);   // row_half_mirror
    s = s + dpp_mov<0x140>(s);   // row_mirror
    return rows_sum(s);
}
// two sums at once
IDHMC_DEV void wave_sum2(double a0, double a1, double b0, double b1, double &sa, double &sb)
{
    double s = a0 + a1, t = b0 + b1;
    s = s + dpp_mov<0xB1>(s);  t = t + dpp_mov<0xB1>(t);
    s = s + dpp_mov<0x4E>(s);  t = t + dpp_mov<0x4E>(t);
    s = s + dpp_mov<0x141>(s); t = t + dpp_mov<0x141>(t);
    s = s + dpp_mov<0x140>(s); t = t + dpp_mov<0x140>(t);
    sa = rows_sum(s);
    sb = rows_sum(t);
}

}  // namespace idhmc
