"""configs[3] split over K contexts (C/K chains each, each on its own stream) on one device: do the HBM phase of one
context's sweep and the matrix phase of another's overlap when the sweeps are issued back to back?  (GPU box)
K=2 GRID=256 python tools/ubench/dense_streams.py      (IDHMC_DENSE_GRID needs a -DIDHMC_DENSE_GRID_ENV build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C = 256, 16384
K = int(os.environ.get("K", 2)); SWEEPS = int(os.environ.get("SWEEPS", 2000))
rng = np.random.default_rng(7)
Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
lam = np.logspace(-2, 0, D)
P = (Q / lam) @ Q.T; P = 0.5 * (P + P.T)
mu = np.cos(np.arange(D, dtype=float))
engs = []
for k in range(K):
    e = pkg.Engine(pkg.DenseMVN(mu, P), C // K, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1, first_chain=k * (C // K))
    e.random_position(); e.refresh_momentum(1); e.leapfrog(0.02, 1); e.synchronize()
    engs.append(e)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(SWEEPS):
        for e in engs:
            e.leapfrog(0.02, 1)
    t_issue = time.perf_counter() - t0
    for e in engs:
        e.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"K={K} grid={os.environ.get('IDHMC_DENSE_GRID', 'auto')}: {best / SWEEPS * 1e6:.1f} us per 16 384-chain sweep = {C * SWEEPS / best:.3e} chain-steps/s "
      f"(host issue {t_issue / SWEEPS * 1e6:.1f} us per sweep)")
