"""configs[3] single-step sweeps only (for tools/profile_dense_pmc.sh): SWEEPS back-to-back sweeps after 20 warm-up sweeps"""
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
eng.random_position(); eng.refresh_momentum(1)
eng.time_leapfrog(0.02, 20)
ms = eng.time_leapfrog(0.02, int(os.environ.get("SWEEPS", 200)))
print(f"{ms * 1e3:.1f} us per sweep")
eng.close()
