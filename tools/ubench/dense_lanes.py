"""configs[3] single-step sweeps with and without the lanes of idhmc_leapfrog (GPU box): IDHMC_DENSE_LANES=0/1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D = 256
rng = np.random.default_rng(7)
Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
lam = np.logspace(-2, 0, D)
P = (Q / lam) @ Q.T; P = 0.5 * (P + P.T)
mu = np.cos(np.arange(D, dtype=float))
for C in [int(x) for x in os.environ.get("CS", "16384").split(",")]:
    eng = pkg.Engine(pkg.DenseMVN(mu, P), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
    eng.random_position(); eng.refresh_momentum(1)
    eng.time_leapfrog(0.02, 20)
    ms = min(eng.time_leapfrog(0.02, 1000) for _ in range(3))
    print(f"lanes={os.environ.get('IDHMC_DENSE_LANES', 'auto')} C={C}: {ms * 1e3:.1f} us per sweep = {C / ms * 1e3:.3e} chain-steps/s "
          f"= {C / ms * 1e3 * 2 * D * D / 1e12:.1f} TFLOP/s, {C / ms * 1e3 * 6 * D * 8 / 1e12:.2f} TB/s of state")
    dc = eng.debug_counters()
    if dc[7] > 0:      # diagnostic build (tools/stamps.sh): 100 MHz wall-clock sums of wavefront 0 per tile
        n = float(dc[7])
        print("   per tile, us: load + loop A %.2f, k loop %.2f, loop B + stores %.2f  (%d tiles)" % (dc[2] / n / 100, dc[3] / n / 100, dc[4] / n / 100, n))
    import time
    eng.synchronize(); t0 = time.perf_counter()
    for i in range(2000):
        eng.leapfrog(0.02, 1)
    ti = time.perf_counter() - t0
    eng.synchronize(); dt = time.perf_counter() - t0
    print(f"   one idhmc_leapfrog call per sweep from Python: {dt / 2000 * 1e6:.1f} us per sweep (host issue {ti / 2000 * 1e6:.1f})")
    eng.close()
