"""Sweep launch geometry of the single-step leapfrog kernel (GPU box). Prints ms/sweep and GB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C = 1024, int(os.environ.get("C", 65536))
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
for kind in ("diag", "iso"):
    model = pkg.DiagGaussian(mu, sigma=sig) if kind == "diag" else pkg.IsoGaussian(D)
    eng = pkg.Engine(model, C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
    eng.set_minv(sig ** 2 if kind == "diag" else np.ones(D))
    eng.random_position(); eng.refresh_momentum(1)
    eps = 0.01
    for var in (0, 1, 2, 3):
      for cap in (0, 2048, 4096):
        os.environ["IDHMC_LF_BLOCKS"] = str(cap)
        os.environ["IDHMC_LF_VARIANT"] = str(var)
        eng.time_leapfrog(eps, 20)
        best = min(eng.time_leapfrog(eps, 100) for _ in range(3))
        print(f"{kind} var={var} cap={cap:6d} ms/sweep={best:.4f} GB/s={6*D*8*C/best/1e6:.0f}", flush=True)
    eng.close()
