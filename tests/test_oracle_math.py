"""Pins the oracle's RNG and elementary functions: Philox-4x32-10 against the published known-answer
vectors of Random123 (Salmon et al., SC'11), the fp64 functions against numpy, the reference's
logaddexp (src/InplaceDHMC.jl:27-30) on its edge cases, and the canonical reduction order."""
import ctypes as C

import numpy as np
import pytest


def philox(L, ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    L.orc_philox_export(c, k, o)
    return list(o)


def test_philox_known_answers(oracle):
    L = oracle.lib()
    # Random123 kat_vectors, philox4x32-10
    assert philox(L, [0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox(L, [0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox(L, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_log_exp_log1p_accuracy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(0.5, 2.0, 20000),
                        [1.0, 2.0, 0.5, 1e-310, 5e-324, 1.7976931348623157e308]])
    got = np.array([L.orc_log_export(v) for v in x])
    assert ulps(got, np.log(x)).max() <= 4
    x = np.concatenate([rng.uniform(-708, 709, 20000), rng.uniform(-1, 1, 20000), [0.0]])
    got = np.array([L.orc_exp_export(v) for v in x])
    assert ulps(got, np.exp(x)).max() <= 2
    x = np.concatenate([rng.uniform(0, 1, 20000), 10.0 ** rng.uniform(-300, 0, 5000)])
    got = np.array([L.orc_log1p_export(v) for v in x])
    assert ulps(got, np.log1p(x)).max() <= 4


def test_log_exp_special_values(oracle):
    L = oracle.lib()
    assert L.orc_log_export(0.0) == -np.inf
    assert np.isnan(L.orc_log_export(-1.0)) and np.isnan(L.orc_log_export(np.nan))
    assert L.orc_log_export(np.inf) == np.inf
    assert L.orc_log_export(1.0) == 0.0
    assert L.orc_exp_export(-np.inf) == 0.0 and L.orc_exp_export(-1e9) == 0.0
    assert L.orc_exp_export(np.inf) == np.inf and L.orc_exp_export(710.0) == np.inf
    assert L.orc_exp_export(0.0) == 1.0
    assert np.isnan(L.orc_exp_export(np.nan))


def test_sincos2pi(oracle):
    L = oracle.lib()
    s, c = C.c_double(), C.c_double()
    u = np.concatenate([np.random.default_rng(1).uniform(0, 1, 20000), [0.0, 0.125, 0.25, 0.5, 0.75, 1 - 2.0 ** -53]])
    S, Cc = [], []
    for v in u:
        L.orc_sincos2pi_export(v, C.byref(s), C.byref(c))
        S.append(s.value)
        Cc.append(c.value)
    S, Cc = np.array(S), np.array(Cc)
    # reference values in extended precision through exact octant reduction
    ref_s = np.sin(2 * np.pi * np.asarray(u, dtype=np.longdouble)).astype(np.float64)
    ref_c = np.cos(2 * np.pi * np.asarray(u, dtype=np.longdouble)).astype(np.float64)
    assert np.abs(S - ref_s).max() < 1e-15 and np.abs(Cc - ref_c).max() < 1e-15
    assert np.abs(S * S + Cc * Cc - 1).max() < 1e-15
    L.orc_sincos2pi_export(0.25, C.byref(s), C.byref(c))
    assert (s.value, c.value) == (1.0, 0.0) or (s.value == 1.0 and abs(c.value) == 0.0)


def test_logaddexp_edge_cases(oracle):
    """reference src/InplaceDHMC.jl:27-30: non-finite arguments short-circuit to max"""
    f = oracle.lib().orc_logaddexp_export
    assert f(-np.inf, 1.5) == 1.5 and f(1.5, -np.inf) == 1.5
    assert f(-np.inf, -np.inf) == -np.inf
    assert f(np.inf, 0.0) == np.inf
    assert abs(f(2.0, 2.0) - (2.0 + np.log(2.0))) < 1e-15
    assert abs(f(0.0, -800.0) - 0.0) < 1e-300
    x, y = -3.25, 0.75
    assert abs(f(x, y) - np.logaddexp(x, y)) < 1e-15 and f(x, y) == f(y, x)
    assert f(np.nan, 1.0) == 1.0          # `x > y ? x : y` with NaN picks y, as the reference does


def test_canonical_dot(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(2)
    for n in (128, 256, 1024):
        a, b = rng.standard_normal(n), rng.standard_normal(n)
        got = L.orc_dot_export(oracle._dp(a), oracle._dp(b), n)
        ref = float(np.dot(a.astype(np.longdouble), b.astype(np.longdouble)))
        assert abs(got - ref) <= 1e-13 * np.abs(a * b).sum()
        ai = rng.integers(-8, 9, n).astype(float)
        bi = rng.integers(-8, 9, n).astype(float)
        assert L.orc_dot_export(oracle._dp(ai), oracle._dp(bi), n) == float(np.dot(ai, bi))
    # the defined order: 128 stride-128 fma chains then an adjacent pairwise tree
    a, b = rng.standard_normal(256), rng.standard_normal(256)
    acc = [float(np.float64(a[r] * b[r])) for r in range(128)]
    import math
    acc = [math.fma(a[r + 128], b[r + 128], math.fma(a[r], b[r], 0.0)) if hasattr(math, "fma") else None for r in range(128)]
    if acc[0] is not None:
        w = 1
        while w < 128:
            for i in range(0, 128, 2 * w):
                acc[i] = acc[i] + acc[i + w]
            w *= 2
        assert acc[0] == L.orc_dot_export(oracle._dp(a), oracle._dp(b), 256)


def test_rng_streams(oracle):
    L = oracle.lib()
    z = np.zeros(1 << 15)
    L.orc_randn_export(12345, 7, 3, len(z), oracle._dp(z))
    assert abs(z.mean()) < 4 / np.sqrt(len(z)) and abs(z.var() - 1) < 0.03 and abs((z ** 4).mean() - 3) < 0.15
    z2 = np.zeros(1 << 15)
    L.orc_randn_export(12345, 8, 3, len(z2), oracle._dp(z2))
    assert abs(np.corrcoef(z, z2)[0, 1]) < 0.03           # chains are independent streams
    e = np.array([L.orc_randexp_export(5, 1, 9, i) for i in range(20000)])
    assert e.min() > 0 and abs(e.mean() - 1) < 0.03 and abs(e.var() - 1) < 0.06
    d = np.array([L.orc_rand_directions_export(5, c, 1) for c in range(4000)], dtype=np.uint32)
    bits = ((d[:, None] >> np.arange(32, dtype=np.uint32)) & 1).mean(axis=0)
    assert np.abs(bits - 0.5).max() < 0.04
    assert L.orc_rand_directions_export(5, 1, 1) == L.orc_rand_directions_export(5, 1, 1)
