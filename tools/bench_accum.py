"""What the on-device accumulators cost per transition (configs[2]'s sampling phase: per-chain metric, running moments, diagnostics counters)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C = 1024, int(os.environ.get("C", 65536))
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, pkg.default_options(), seed=1)
eng.set_minv(sig ** 2)
rng = np.random.default_rng(1)
q0 = np.empty((C, D))
for i in range(0, C, 4096):
    q0[i:i + 4096] = mu + sig * rng.standard_normal((min(4096, C - i), D))
eng.set_q(q0); del q0
eng.set_eps(0.25)
eng.moments_reset(); eng.diag_reset()
for name, fl in (("none", 0), ("moments", pkg.T_ACCUM_MOMENTS), ("diag", pkg.T_ACCUM_DIAG), ("moments+diag", pkg.T_ACCUM_MOMENTS | pkg.T_ACCUM_DIAG),
                 ("metric window", pkg.T_ACCUM_METRIC), ("adapt eps", pkg.T_ADAPT_EPS)):
    if fl & pkg.T_ACCUM_METRIC: eng.metric_begin()
    if fl & pkg.T_ADAPT_EPS: eng.da_init()
    for it in range(1, 4): eng.nuts_transition(it, fl)
    eng.synchronize(); s0 = eng.total_steps(); t0 = time.perf_counter()
    for it in range(4, 24): eng.nuts_transition(it, fl)
    eng.synchronize(); dt = time.perf_counter() - t0; st = eng.total_steps() - s0
    print(f"{name:14s} {dt/20*1e3:6.2f} ms/transition  {st/dt:.3e} leapfrog/s  (mean steps {st/20/C:.1f})", flush=True)
    if fl & pkg.T_ADAPT_EPS: eng.set_eps(0.25)
