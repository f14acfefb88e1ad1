"""FindLocalOptimum (reference src/warmup.jl:137-187) on the device against the CPU oracle.
The reference's optimiser (QuasiNewtonMethods.proptimize!) is not in the reference tree -- parity unpinned -- so
both sides run the engine's own L-BFGS (5 pairs, Armijo backtracking, canonical reductions) behind the stage's
contract: maximise l(q) - penalty/2 sum(q^2) for <= iterations; a non-finite result restarts from a new random
position with the penalty doubled, <= 100 times.  Bit-exact fp64: q, l(q), grad l(q) after the stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def dense_problem(D, seed=7):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    lam = np.logspace(-2, 0, D)
    P = (Q / lam) @ Q.T
    return np.cos(np.arange(D, dtype=np.float64)), 0.5 * (P + P.T)


def models(idhmc, oracle, kind, D):
    if kind == "iso":
        return idhmc.IsoGaussian(D), oracle.OracleModel.iso(D)
    if kind == "diag":
        mu, sig = np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)
        return idhmc.DiagGaussian(mu, sigma=sig), oracle.OracleModel.diag(mu, 1.0 / sig ** 2)
    mu, P = dense_problem(D)
    return idhmc.DenseMVN(mu, P), oracle.OracleModel.dense(mu, P)


@pytest.mark.parametrize("kind,D,C,iters", [("iso", 32, 9, 50), ("diag", 100, 7, 50), ("diag", 1024, 5, 30),
                                            ("diag", 300, 6, 12), ("dense", 40, 6, 50), ("dense", 256, 5, 25)])
def test_local_optimum_matches_oracle(idhmc, oracle, kind, D, C, iters):
    gm, om = models(idhmc, oracle, kind, D)
    eng = idhmc.Engine(gm, C, seed=17)
    chains = [oracle.OracleChain(om, seed=17, chain_id=c) for c in range(C)]
    eng.random_position()
    lq0 = eng.lq.copy()
    eng.find_local_optimum(1e-4, iters)
    for ch in chains:
        ch.random_position()
        assert ch.find_local_optimum(1e-4, iters) == 0
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains]))
    assert same_bits(eng.grad, np.stack([c.grad[:D] for c in chains]))
    assert same_bits(eng.lq, [c.lq for c in chains])
    assert np.all(eng.lq > lq0)                       # it did climb
    if kind == "iso":
        assert np.abs(eng.q).max() < 1e-6             # the mode of N(0, I) (the 1e-4 penalty does not move it)


HIP_HALF = r"""
// l(q) = -1/2 |q|^2 on the half space q_0 >= 0.25, -Inf elsewhere
template <int NCH>
__device__ double logdensity_and_gradient(const Vec<NCH> &q, Vec<NCH> &grad, const UserCtx &ctx)
{
    double l0 = 0.0, l1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        grad.c[j].x = -q.c[j].x; grad.c[j].y = -q.c[j].y;
        l0 = dfma(q.c[j].x, q.c[j].x, l0); l1 = dfma(q.c[j].y, q.c[j].y, l1);
    }
    const double l = -0.5 * wave_sum(l0, l1);
    const double q0 = read_lane(q.c[0].x, 0);
    return q0 >= ctx.params[0] ? l : -kInf;
}
"""
C_HALF = r"""
#include <math.h>
#include "orc_math.h"
double logdensity_and_gradient(const double *q, double *grad, int D, int L, const double *params)
{
    double acc[128];
    for (int r = 0; r < 128; ++r) acc[r] = 0.0;
    for (int j = 0; j < L; j += 128)
        for (int r = 0; r < 128; ++r) { grad[j + r] = -q[j + r]; acc[r] = fma(q[j + r], q[j + r], acc[r]); }
    const double l = -0.5 * orc_tree128(acc);
    return q[0] >= params[0] ? l : -INFINITY;
}
"""


def test_restarts_from_a_nonfinite_start(idhmc, oracle, tmp_path):
    """chains whose random start has no finite density are restarted from fresh draws with the penalty doubled
    (src/warmup.jl:167-171); the line search backs off the -Inf region; identical on both sides"""
    D, C = 24, 24
    eng = idhmc.Engine(idhmc.CustomDensity(D, HIP_HALF, [0.25]), C, seed=8)
    om = oracle.OracleModel.custom(D, C_HALF, [0.25], str(tmp_path))
    chains = [oracle.OracleChain(om, seed=8, chain_id=c) for c in range(C)]
    eng.random_position()
    bad = ~np.isfinite(eng.lq)
    assert 3 < bad.sum() < C - 3                      # both kinds of start are present
    eng.find_local_optimum(1e-4, 40)
    for ch in chains:
        ch.random_position()
        assert ch.find_local_optimum(1e-4, 40) == 0
    assert np.all(np.isfinite(eng.lq)) and np.all(eng.q[:, 0] >= 0.25)
    assert same_bits(eng.q, np.stack([c.q[:D] for c in chains])) and same_bits(eng.lq, [c.lq for c in chains])


def test_failure_is_a_status_code(idhmc):
    """no finite density anywhere: 100 restarts, then IDHMC_ERR_OPTIMIZATION (reference throws, src/warmup.jl:172)"""
    eng = idhmc.Engine(idhmc.CustomDensity(8, HIP_HALF, [1e9]), 3, seed=1)
    eng.random_position()
    with pytest.raises(idhmc.IdhmcError) as e:
        eng.find_local_optimum(1e-4, 10)
    assert e.value.code == 8 and "failed to converge" in str(e.value)


def test_default_stages_run_the_optimum_stage(idhmc, oracle):
    """reference default_warmup_stages starts with FindLocalOptimum() (src/warmup.jl:362): the API-level pipeline
    with it equals the oracle's with local_opt_iterations = 50"""
    D, C, N = 20, 3, 12
    short = dict(init_steps=15, middle_steps=10, doubling_stages=2, terminating_steps=10)
    stages = idhmc.default_warmup_stages(**short)
    assert isinstance(stages[0], idhmc.FindLocalOptimum)
    chains, stats = idhmc.threaded_mcmc(idhmc.IsoGaussian(D), N, nchains=C, warmup_stages=stages,
                                        algorithm=idhmc.NUTS(max_depth=6), seed=4)
    oo = oracle.default_options(max_depth=6, local_opt_iterations=50, local_opt_penalty=1e-4, **short)
    rc, och, ost, _ = oracle.threaded_mcmc(oracle.OracleModel.iso(D), N, C, oo, seed=4)
    assert rc == 0
    NS = chains.shape[1]
    assert same_bits(chains[:, :N, :], och[:, :N, :D]) and NS == och.shape[1]
    # and through the C-level driver with the option set
    eng = idhmc.Engine(idhmc.IsoGaussian(D), C, idhmc.default_options(max_depth=6, local_opt_iterations=50, **short), seed=4)
    draws, _ = eng.mcmc_with_warmup(N)
    assert same_bits(draws.transpose(1, 0, 2), och[:, :N, :D])
