"""How long is the walk to a good placement of the state arrays, with every candidate set held whole (IDHMC_PLACEMENT_SPACERS=0, rounds 2-3) and with
spacers (the default since round 3: a rejected set keeps one array only)?  Contexts created in turn in one process and KEPT (each new one must land elsewhere),
alternating the two strategies; prints candidates tried, probe ratio and the sweep's rate.  Run after something has churned the allocator."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C = 1024, 65536
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
keep = []
for i in range(int(os.environ.get("N", 10))):
    mode = "spacers" if i % 2 else "whole sets"
    os.environ["IDHMC_PLACEMENT_SPACERS"] = "1" if mode == "spacers" else "0"
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
    eng.set_minv(sig ** 2); eng.random_position(); eng.refresh_momentum(1)
    eng.time_leapfrog(0.1, 50)
    ms = min(eng.time_leapfrog(0.1, 200) for _ in range(3))
    g, n = eng.placement_info(); cost = eng.placement_cost()
    print("context %2d %-10s candidates %2d  probe %.3f x one array  create %.0f ms  peak %.1f GiB  sweep %.4f ms = %.4e leapfrog-steps/s"
          % (i, mode, n, g / cost["single_array_GBps"] if cost["single_array_GBps"] else 0.0, cost["create_ms"], cost["peak_transient_bytes"] / 2**30, ms, C / ms * 1e3), flush=True)
    if os.environ.get("HOLD", "1") == "1" and len(keep) < 6: keep.append(eng)
    else: eng.close()
