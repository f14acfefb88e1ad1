"""configs[3] with every draw kept on the host (the reference's API keeps them, src/mcmc.jl:143): idhmc_mcmc(N, draws, stats) for the
dense 256-dim MVN, 16 384 chains; run with IDHMC_FUSE=0 for one launch per transition.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C, N = 256, int(os.environ.get("C", 16384)), int(os.environ.get("N", 48))
rng = np.random.default_rng(7)
Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
lam = np.logspace(-2, 0, D)
P = (Q / lam) @ Q.T; P = 0.5 * (P + P.T)
mu = np.cos(np.arange(D, dtype=float))
eng = pkg.Engine(pkg.DenseMVN(mu, P), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
Sig = (Q * lam) @ Q.T
eng.set_q(mu + rng.standard_normal((C, D)) @ np.linalg.cholesky(0.5 * (Sig + Sig.T)).T)
eng.set_eps(0.05)
eng.mcmc(8, 0)
s0 = eng.total_steps(); t0 = time.perf_counter(); d, s = eng.mcmc(N, 8); t1 = time.perf_counter()
steps = eng.total_steps() - s0
print(f"fused={eng.fused_launch_info()[1]} C={C} N={N}: storing draws and records {(t1-t0)/N*1e3:.2f} ms per transition, {steps/(t1-t0):.3e} leapfrog/s, "
      f"{d.nbytes/(t1-t0)/1e9:.1f} GB/s to the host; checksum {float(d.sum()):.9e}")
