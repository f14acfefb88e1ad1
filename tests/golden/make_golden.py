#!/usr/bin/env python3
"""Generates the committed golden vectors from the CPU oracle (python tests/golden/make_golden.py).

The reference cannot produce fixtures (pure Julia, no `julia` in the image, empty test suite), so these
vectors pin the ORACLE against drift and give the GPU path a fixed target that needs no oracle build on
the GPU box.  They are data only: inputs (seeds, sizes, eps) and expected outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def diag_params(D):
    return np.sin(np.arange(D, dtype=np.float64)), np.logspace(-1, 1, D)


def transitions(model, D, seed, eps, C, T, max_depth, minv=None):
    opt = O.default_options(max_depth=max_depth)
    q = np.zeros((T, C, D))
    stats = np.zeros((T, C), dtype=O.STATS_DTYPE)
    for c in range(C):
        ch = O.OracleChain(model, opt, seed=seed, chain_id=c)
        if minv is not None:
            ch.set_minv(minv)
        ch.random_position()
        for t in range(T):
            st = ch.sample_tree(eps, t + 1)
            q[t, c] = ch.q[:D]
            stats[t, c] = (st.pi, st.acceptance_rate, st.term_left, st.term_right, st.depth, st.steps)
    return q, stats


def dense_problem(D, seed=7):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    lam = np.logspace(-2, 0, D)
    P = (Q / lam) @ Q.T
    return np.cos(np.arange(D, dtype=np.float64)), 0.5 * (P + P.T)


def later_fixtures():
    """fixtures added after the first set (kept separate so that regenerating them leaves the older files alone)"""
    # FindLocalOptimum stage (engine's own L-BFGS): state after the stage from the random start
    mu, sig = diag_params(100)
    m = O.OracleModel.diag(mu, 1 / sig ** 2)
    qs, gs, lqs = [], [], []
    for c in range(4):
        ch = O.OracleChain(m, seed=17, chain_id=c)
        ch.random_position()
        assert ch.find_local_optimum(1e-4, 50) == 0
        qs.append(ch.q[:100].copy()); gs.append(ch.grad[:100].copy()); lqs.append(ch.lq)
    np.savez(os.path.join(HERE, "optimum_diag100.npz"), seed=17, penalty=1e-4, iterations=50,
             q=np.array(qs), grad=np.array(gs), lq=np.array(lqs))
    # dense MVN (configs[3] family): NUTS transitions, D = 64, 20 chains (one full and one ragged group of 16)
    mu, P = dense_problem(64)
    q, st = transitions(O.OracleModel.dense(mu, P), 64, 5, 0.04, 20, 6, 8)
    np.savez(os.path.join(HERE, "transitions_dense64.npz"), seed=5, eps=0.04, max_depth=8, q=q, stats=st)
    print("later fixtures written to", HERE)


def round3_fixtures():
    """long trees (round 3: the GPU evaluates the tree's bookkeeping after the tree, 64 leaves per pass): D = 40 diagonal Gaussian with the
    identity metric, eps = 0.004, max_depth 10 -- depths 5 to 10, doublings that stop between and inside their 64-leaf blocks"""
    mu, sig = diag_params(40)
    q, st = transitions(O.OracleModel.diag(mu, 1 / sig ** 2), 40, 77, 0.004, 8, 4, 10)
    np.savez(os.path.join(HERE, "transitions_diag40_long.npz"), seed=77, eps=0.004, max_depth=10, q=q, stats=st)
    print("round-3 fixture written to", HERE, "depths", np.bincount(st["depth"].ravel()).tolist())


def main():
    if "--round3-only" in sys.argv:
        return round3_fixtures()
    if "--later-only" in sys.argv:
        return later_fixtures()
    L = O.lib()
    z = np.zeros(256)
    L.orc_randn_export(20261004, 3, 5, 256, O._dp(z))
    np.savez(os.path.join(HERE, "rng.npz"), seed=20261004, chain=3, iter=5, randn=z,
             randexp=np.array([L.orc_randexp_export(20261004, 3, 5, i) for i in range(16)]),
             directions=np.array([L.orc_rand_directions_export(20261004, c, 5) for c in range(16)], dtype=np.uint32))

    q, st = transitions(O.OracleModel.iso(32), 32, 2026, 0.3, 8, 10, 5)
    np.savez(os.path.join(HERE, "transitions_iso32.npz"), seed=2026, eps=0.3, max_depth=5, q=q, stats=st)
    mu, sig = diag_params(100)
    q, st = transitions(O.OracleModel.diag(mu, 1 / sig ** 2), 100, 7, 0.08, 6, 8, 8, minv=sig ** 2)
    np.savez(os.path.join(HERE, "transitions_diag100.npz"), seed=7, eps=0.08, max_depth=8, q=q, stats=st)

    mu, sig = diag_params(1024)
    m = O.OracleModel.diag(mu, 1 / sig ** 2)
    qs, ps, lqs = [], [], []
    for c in range(3):
        ch = O.OracleChain(m, seed=1, chain_id=c)
        ch.set_minv(sig ** 2)
        ch.random_position()
        ch.rand_p(1)
        for _ in range(3):
            ch.leapfrog(0.1)
        qs.append(ch.q[:1024].copy()); ps.append(ch.p[:1024].copy()); lqs.append(ch.lq)
    np.savez(os.path.join(HERE, "leapfrog_diag1024.npz"), seed=1, eps=0.1, steps=3,
             q=np.array(qs), p=np.array(ps), lq=np.array(lqs))

    # BASELINE.json configs[0]: full default warmup, 4 chains, max_depth 5, N = 100 draws
    opt = O.default_options(max_depth=5)
    rc, chains, stats, eps = O.threaded_mcmc(O.OracleModel.iso(32), 100, 4, opt, seed=20261004)
    assert rc == 0
    np.savez(os.path.join(HERE, "cfg1_iso32.npz"), seed=20261004, N=100, max_depth=5, eps=eps,
             last_draw=chains[:, 99, :32], draw_sum=chains[:, :100, :32].sum(axis=1), stats=stats[:, :100])
    print("golden vectors written to", HERE)
    later_fixtures()
    round3_fixtures()


if __name__ == "__main__":
    main()
