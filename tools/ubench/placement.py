"""Does the rate of the headline sweep depend on where the allocator put the state arrays?  Engines created and destroyed in turn in
one process (GPU box); HOLD=1 keeps every engine alive, so that each new one must land somewhere else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C = 1024, 65536
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
keep = []
for i in range(int(os.environ.get("N", 8))):
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
    eng.set_minv(sig ** 2); eng.random_position(); eng.refresh_momentum(1)
    eng.time_leapfrog(0.1, 50)
    ms = min(eng.time_leapfrog(0.1, 200) for _ in range(3))
    print("engine %d: %.4f ms per sweep = %.4e leapfrog-steps/s" % (i, ms, C / ms * 1e3), flush=True)
    if os.environ.get("HOLD"): keep.append(eng)
    else: eng.close()
