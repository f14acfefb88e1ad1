"""configs[3] NUTS only (for tools/profile_dense_nuts_pmc.sh): 2 warm transitions, 3 single-transition launches, then ONE launch of 20
transitions (idhmc_nuts_transitions); prints the leapfrog steps of each part so that counters can be quoted per gradient"""
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
eng = pkg.Engine(pkg.DenseMVN(mu, P), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
Sig = (Q * lam) @ Q.T
q0 = mu + rng.standard_normal((C, D)) @ np.linalg.cholesky(0.5 * (Sig + Sig.T)).T
eng.set_q(q0); eng.refresh_momentum(1); eng.set_eps(0.05)
steps = []
for it in range(1, 6):
    s0 = eng.total_steps(); eng.nuts_transition(it); eng.synchronize(); steps.append(eng.total_steps() - s0)
s0 = eng.total_steps(); eng.nuts_transitions(6, 20); eng.synchronize(); steps.append(eng.total_steps() - s0)
print("leapfrog steps per k_nuts launch:", steps)
eng.close()
