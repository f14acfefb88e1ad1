/*
 * orc_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Deterministic fp64 elementary functions, the counter-based RNG and the
 * canonical reduction order used by the CPU oracle.  The HIP product
 * (inplacedhmc.jl_amd/csrc/) carries its own independent statement of the same
 * published algorithms; the two must agree bit for bit, which is what
 * tests/test_gpu_parity.py checks.
 *
 * Why own log/exp/sincos: the reference draws its randomness from
 * VectorizedRNG.jl (reference src/rng.jl:8, src/kinetic_energy.jl:63,
 * src/NUTS.jl:33, src/tree.jl:144), an un-vendored, un-pinned dependency whose
 * bit streams cannot be known here.  "Identical seeds" is made
 * meaningful by a shared counter-based generator (Philox-4x32-10, Salmon et
 * al., SC'11) and by elementary functions that use only IEEE-754 +,-,*,/,fma
 * and rint, so host libm and device ocml cannot disagree in the last bit.
 *
 * parity unpinned: the reference ships no golden vectors (test/runtests.jl:4-6
 * is empty) and cannot run here (no Julia).  These functions are pinned by
 * published known-answer vectors (Philox) and by accuracy checks against
 * numpy (tests/test_oracle_math.py).
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint64_t orc_bits(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
static inline double orc_frombits(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

#define ORC_LN2_HI 6.93147180369123816490e-01 /* 0x3fe62e42fee00000 */
#define ORC_LN2_LO 1.90821492927058770002e-10 /* 0x3dea39ef35793c76 */
#define ORC_INV_LN2 1.44269504088896338700e+00
#define ORC_PI_2 1.57079632679489661923

/* natural log: x = 2^e * m, m in (sqrt(1/2), sqrt(2)]; log m = 2 atanh(s),
 * s = (m-1)/(m+1), odd series to s^23. */
static inline double orc_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    uint64_t b = orc_bits(x);
    int e = 0;
    if (b < 0x0010000000000000ull) { x *= 0x1p54; b = orc_bits(x); e = -54; }
    e += (int)(b >> 52) - 1023;
    double m = orc_frombits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (m + 1.0);
    double z = s * s;
    double P = 1.0 / 23.0;
    P = fma(P, z, 1.0 / 21.0);
    P = fma(P, z, 1.0 / 19.0);
    P = fma(P, z, 1.0 / 17.0);
    P = fma(P, z, 1.0 / 15.0);
    P = fma(P, z, 1.0 / 13.0);
    P = fma(P, z, 1.0 / 11.0);
    P = fma(P, z, 1.0 / 9.0);
    P = fma(P, z, 1.0 / 7.0);
    P = fma(P, z, 1.0 / 5.0);
    P = fma(P, z, 1.0 / 3.0);
    P = fma(P, z, 1.0);
    double r = (s + s) * P;
    double de = (double)e;
    return fma(de, ORC_LN2_HI, fma(de, ORC_LN2_LO, r));
}

/* exp: x = k ln2 + r, |r| <= ln2/2, Taylor to r^13, scale by 2^k. */
static inline double orc_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -708.3964185322641) return 0.0;
    double k = rint(x * ORC_INV_LN2);
    double r = fma(-k, ORC_LN2_HI, x);
    r = fma(-k, ORC_LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int ki = (int)k;
    if (ki > 1023) { p *= 2.0; ki -= 1; }
    return p * orc_frombits((uint64_t)(ki + 1023) << 52);
}

static inline double orc_log1p(double x)
{
    double u = 1.0 + x;
    if (u == 1.0) return x;
    if (u == INFINITY) return u;
    double d = u - 1.0;
    return orc_log(u) * (x / d);
}

/* sin and cos of 2*pi*u for u in [0,1): quadrant n = rint(4u), exact remainder. */
static inline void orc_sincos2pi(double u, double *sn, double *cs)
{
    double t = 4.0 * u;
    double n = rint(t);
    double x = (t - n) * ORC_PI_2;
    double z = x * x;
    double S = 1.0 / 355687428096000.0;            /* 1/17! */
    S = fma(S, z, -1.0 / 1307674368000.0);         /* 15! */
    S = fma(S, z, 1.0 / 6227020800.0);             /* 13! */
    S = fma(S, z, -1.0 / 39916800.0);              /* 11! */
    S = fma(S, z, 1.0 / 362880.0);                 /* 9! */
    S = fma(S, z, -1.0 / 5040.0);
    S = fma(S, z, 1.0 / 120.0);
    S = fma(S, z, -1.0 / 6.0);
    S = fma(S, z, 1.0);
    S = S * x;
    double C = 1.0 / 20922789888000.0;             /* 1/16! */
    C = fma(C, z, -1.0 / 87178291200.0);           /* 14! */
    C = fma(C, z, 1.0 / 479001600.0);              /* 12! */
    C = fma(C, z, -1.0 / 3628800.0);               /* 10! */
    C = fma(C, z, 1.0 / 40320.0);
    C = fma(C, z, -1.0 / 720.0);
    C = fma(C, z, 1.0 / 24.0);
    C = fma(C, z, -0.5);
    C = fma(C, z, 1.0);
    switch (((int)n) & 3) {
    case 0: *sn = S; *cs = C; break;
    case 1: *sn = C; *cs = -S; break;
    case 2: *sn = -S; *cs = -C; break;
    default: *sn = -C; *cs = S; break;
    }
}

/* reference src/InplaceDHMC.jl:27-30 */
static inline double orc_logaddexp(double x, double y)
{
    if (!(isfinite(x) && isfinite(y))) return x > y ? x : y;
    return x > y ? x + orc_log1p(orc_exp(y - x)) : y + orc_log1p(orc_exp(x - y));
}

/* ---- Philox-4x32-10 (Salmon, Moraes, Dror, Shaw 2011) ------------------- */
static inline void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                  uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* RNG address: key = seed, counter = (index, iteration, global chain id, stream). */
enum { ORC_STREAM_DIR = 0, ORC_STREAM_MOMENTUM = 1, ORC_STREAM_EXP = 2, ORC_STREAM_INITQ = 3 };

static inline void orc_rng(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t stream,
                           uint32_t idx, uint32_t out[4])
{
    orc_philox4x32(idx, iter, chain, stream, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}
static inline double orc_u01_open0(uint32_t lo, uint32_t hi) /* (0,1] */
{
    uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)(v + 1) * 0x1p-53;
}
static inline double orc_u01(uint32_t lo, uint32_t hi) /* [0,1) */
{
    uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)v * 0x1p-53;
}
/* two standard normals for element pair `pair` (Box-Muller). */
static inline void orc_randn_pair(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t pair,
                                  double *n0, double *n1)
{
    uint32_t x[4];
    orc_rng(seed, chain, iter, ORC_STREAM_MOMENTUM, pair, x);
    double u1 = orc_u01_open0(x[0], x[1]);
    double u2 = orc_u01(x[2], x[3]);
    double r = sqrt(-2.0 * orc_log(u1));
    double s, c;
    orc_sincos2pi(u2, &s, &c);
    *n0 = r * c;
    *n1 = r * s;
}
static inline double orc_randexp(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t draw)
{
    uint32_t x[4];
    orc_rng(seed, chain, iter, ORC_STREAM_EXP, draw, x);
    return -orc_log(orc_u01_open0(x[0], x[1]));
}
static inline uint32_t orc_rand_directions(uint64_t seed, uint32_t chain, uint32_t iter)
{
    uint32_t x[4];
    orc_rng(seed, chain, iter, ORC_STREAM_DIR, 0, x);
    return x[0];
}

/* ---- canonical reduction order -------------------------------------------
 * sum_i a_i*b_i over a padded length L (multiple of 128):
 *   acc[r] = fma chain over i = r, r+128, r+256, ... (increasing i), r in [0,128)
 *   then an adjacent pairwise tree over acc[0..127].
 * On the GPU: lane l of the wavefront owns r = 2l, 2l+1; the tree is
 * acc0+acc1 then a 6-level xor butterfly over the 64 lanes. */
static inline double orc_tree128(double *acc)
{
    for (int w = 1; w < 128; w <<= 1)
        for (int i = 0; i < 128; i += 2 * w) acc[i] = acc[i] + acc[i + w];
    return acc[0];
}
static inline double orc_dot(const double *a, const double *b, int L)
{
    double acc[128];
    for (int r = 0; r < 128; ++r) acc[r] = 0.0;
    for (int j = 0; j < L; j += 128)
        for (int r = 0; r < 128; ++r) acc[r] = fma(a[j + r], b[j + r], acc[r]);
    return orc_tree128(acc);
}
#endif
