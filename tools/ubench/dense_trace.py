"""short run for rocprofv3 --kernel-trace: MODE=lanes (one context, in-library lanes) or MODE=engines (K contexts)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C = 256, 16384
rng = np.random.default_rng(7)
Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
lam = np.logspace(-2, 0, D)
P = (Q / lam) @ Q.T; P = 0.5 * (P + P.T)
mu = np.cos(np.arange(D, dtype=float))
opt = pkg.default_options(metric_mode=pkg.METRIC_SHARED)
if os.environ.get("MODE", "lanes") == "lanes":
    engs = [pkg.Engine(pkg.DenseMVN(mu, P), C, opt, seed=1)]
else:
    K = 4
    engs = [pkg.Engine(pkg.DenseMVN(mu, P), C // K, opt, seed=1, first_chain=k * (C // K)) for k in range(K)]
for e in engs:
    e.random_position(); e.refresh_momentum(1)
for i in range(300):
    for e in engs:
        e.leapfrog(0.02, 1)
for e in engs:
    e.synchronize()
