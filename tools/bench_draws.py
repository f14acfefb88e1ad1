"""Throughput of the draw-storing caller loop (idhmc_mcmc with draws != NULL): the reference's API keeps every
draw (chain D x N x nchains, src/mcmc.jl:143), so this is the path a reference user hits.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D = int(os.environ.get("D", 1024)); C = int(os.environ.get("C", 4096)); N = int(os.environ.get("N", 60))
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
eng.set_minv(sig ** 2)
eng.set_q(mu + sig * np.random.default_rng(1).standard_normal((C, D)))
eng.set_eps(0.25)
eng.mcmc(5, 0, store_draws=True)
t0 = time.perf_counter(); d, s = eng.mcmc(N, 5, store_draws=True); t1 = time.perf_counter()
t2 = time.perf_counter(); eng.mcmc(N, 5 + N, store_draws=False, store_stats=False); eng.synchronize(); t3 = time.perf_counter()
print(f"C={C} D={D} N={N}: storing draws {N*C/(t1-t0):.3e} transitions/s ({(t1-t0)/N*1e3:.2f} ms each, "
      f"{d.nbytes/(t1-t0)/1e9:.1f} GB/s to the host); not storing {N*C/(t3-t2):.3e} ({(t3-t2)/N*1e3:.2f} ms each); "
      f"checksum {float(d.sum()):.6e}")
