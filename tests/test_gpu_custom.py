"""The plug-in form of the downward boundary (IDHMC_MODEL_CUSTOM): a user density given as HIP source,
compiled with hipRTC against the engine's kernel templates, runs through evaluation, leapfrog, the
stepsize search, NUTS transitions and the full warm-up schedule -- and is bit-identical to the same density
handed to the CPU oracle as C source (same arithmetic written twice).

The density couples all coordinates:  l(q) = -sum_i [a/4 q_i^4 + b/2 q_i^2] - c/2 (sum_i q_i)^2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HIP_SRC = r"""
template <int NCH>
__device__ double logdensity_and_gradient(const Vec<NCH> &q, Vec<NCH> &grad, const UserCtx &ctx)
{
    const double a = ctx.params[0], b = ctx.params[1], c = ctx.params[2];
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) { s0 = s0 + q.c[j].x; s1 = s1 + q.c[j].y; }
    const double S = wave_sum(s0, s1);
    double l0 = 0.0, l1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i0 = 128 * j + 2 * ctx.lane;
        const double x = q.c[j].x, y = q.c[j].y;
        const double x2 = x * x, y2 = y * y;
        const double gx = -(a * (x2 * x)) - b * x - c * S, gy = -(a * (y2 * y)) - b * y - c * S;
        grad.c[j].x = (i0 < ctx.D) ? gx : 0.0;
        grad.c[j].y = (i0 + 1 < ctx.D) ? gy : 0.0;
        l0 = l0 + (0.25 * a * (x2 * x2) + 0.5 * b * x2);
        l1 = l1 + (0.25 * a * (y2 * y2) + 0.5 * b * y2);
    }
    return -wave_sum(l0, l1) - 0.5 * c * (S * S);
}
"""

C_SRC = r"""
#include "orc_math.h"
double logdensity_and_gradient(const double *q, double *grad, int D, int L, const double *params)
{
    const double a = params[0], b = params[1], c = params[2];
    double acc[128], lac[128];
    for (int r = 0; r < 128; ++r) { acc[r] = 0.0; lac[r] = 0.0; }
    for (int j = 0; j < L; j += 128)
        for (int r = 0; r < 128; ++r) acc[r] = acc[r] + q[j + r];
    const double S = orc_tree128(acc);
    for (int j = 0; j < L; j += 128)
        for (int r = 0; r < 128; ++r) {
            const int i = j + r;
            const double x = q[i], x2 = x * x;
            const double g = -(a * (x2 * x)) - b * x - c * S;
            grad[i] = i < D ? g : 0.0;
            lac[r] = lac[r] + (0.25 * a * (x2 * x2) + 0.5 * b * x2);
        }
    return -orc_tree128(lac) - 0.5 * c * (S * S);
}
"""
PARAMS = [0.7, 0.5, 0.02]


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("D,metric", [(40, "per_chain"), (200, "per_chain"), (1024, "shared")])
def test_custom_density_matches_the_oracle(idhmc, oracle, tmp_path, D, metric):
    C = 6
    opt = idhmc.default_options(max_depth=7, metric_mode=idhmc.METRIC_SHARED if metric == "shared" else idhmc.METRIC_PER_CHAIN)
    eng = idhmc.Engine(idhmc.CustomDensity(D, HIP_SRC, PARAMS), C, opt, seed=31)
    om = oracle.OracleModel.custom(D, C_SRC, PARAMS, str(tmp_path))
    chains = [oracle.OracleChain(om, oracle.default_options(max_depth=7), seed=31, chain_id=c) for c in range(C)]
    eng.random_position()
    for ch in chains:
        ch.random_position()
    q = eng.q[0]
    ref = -np.sum(0.7 / 4 * q ** 4 + 0.5 / 2 * q ** 2) - 0.02 / 2 * q.sum() ** 2
    assert abs(eng.lq[0] - ref) < 1e-10 * abs(ref)
    assert same_bits(eng.lq, [c.lq for c in chains]) and same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))
    eng.refresh_momentum(1)
    eng.leapfrog(0.05, 2)
    eng.leapfrog(-0.05, 1)
    for ch in chains:
        ch.rand_p(1)
        ch.leapfrog(0.05); ch.leapfrog(0.05); ch.leapfrog(-0.05)
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains])) and same_bits(eng.logdensity(), [c.logdensity() for c in chains])
    eng.refresh_momentum(0)
    eng.find_initial_stepsize()
    ref_eps = []
    for ch in chains:
        ch.rand_p(0)
        rc, e = ch.find_initial_stepsize()
        assert rc == 0
        ref_eps.append(e)
    assert same_bits(eng.eps, ref_eps)
    eng.set_eps(0.1)
    for it in (1, 2, 3, 4):
        eng.nuts_transition(it)
        st = eng.tree_stats()
        ost = [ch.sample_tree(0.1, it) for ch in chains]
        assert st["steps"].tolist() == [s.steps for s in ost] and st["depth"].tolist() == [s.depth for s in ost]
        assert same_bits(st["pi"], [s.pi for s in ost])
        assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))


def test_custom_density_full_warmup_and_posterior(idhmc, oracle, tmp_path):
    D, C, N = 32, 8, 25
    short = dict(init_steps=15, middle_steps=10, doubling_stages=2, terminating_steps=10, max_depth=7)
    eng = idhmc.Engine(idhmc.CustomDensity(D, HIP_SRC, PARAMS), C, idhmc.default_options(**short), seed=5)
    draws, stats = eng.mcmc_with_warmup(N)
    om = oracle.OracleModel.custom(D, C_SRC, PARAMS, str(tmp_path))
    rc, och, ost, oeps = oracle.threaded_mcmc(om, N, C, oracle.default_options(**short), seed=5)
    assert rc == 0 and same_bits(eng.eps, oeps)
    for n in range(N):
        assert same_bits(draws[n], och[:, n, :D])
    assert np.array_equal(stats.T, ost[:, :N])
    assert np.isfinite(draws).all() and abs(draws.mean()) < 0.5


def test_compile_errors_are_reported(idhmc):
    with pytest.raises(idhmc.IdhmcError) as e:
        idhmc.Engine(idhmc.CustomDensity(8, "this is not HIP", None), 2)
    assert e.value.code == 1 and "did not compile" in str(e.value)
    with pytest.raises(idhmc.IdhmcError):
        idhmc.Engine(idhmc.Model(3, 8), 2)                      # no source
