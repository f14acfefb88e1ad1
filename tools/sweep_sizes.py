"""Size sweeps on one MI355X (GPU box): the fused leapfrog sweep and NUTS transitions over chain counts at D = 1024, and over
dimensions (multiples of 128 and not) at 65 536 chains.  Diagonal Gaussian, shared metric M^-1 = sigma^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg


def run(D, C, eps_nuts=0.25):
    sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
    eng.set_minv(sig ** 2)
    eng.random_position()
    eng.refresh_momentum(1)
    eng.time_leapfrog(0.1, 20)
    ms = min(eng.time_leapfrog(0.1, 100) for _ in range(2))
    eng.set_eps(eps_nuts)
    for it in range(1, 4):
        eng.nuts_transition(it)
    eng.synchronize()
    s0 = eng.total_steps()
    msn = eng.time_transitions(5, 3)
    steps = eng.total_steps() - s0
    depth = eng.tree_stats()["depth"].mean()
    L = eng.padded_dim()
    eng.close()
    print(f"D={D:5d} L={L:5d} C={C:7d}  leapfrog {C/ms*1e3:.3e} steps/s = {C/ms*1e3*6*D*8/1e9:6.0f} GB/s of state (6 D 8 B), "
          f"{C/ms*1e3*6*L*8/1e9:6.0f} GB/s moved (6 L 8 B)   |  NUTS {steps/msn*1e3:.3e} leapfrog/s at depth {depth:.2f}", flush=True)


print("== chain count, D = 1024")
for C in (1024, 4096, 16384, 65536, 131072, 262144):
    run(1024, C)
print("== dimension, 65 536 chains (round 1 padded to 128 * 2^k: D = 1100 ran at L = 2048)")
for D in (128, 200, 256, 384, 512, 640, 768, 1024, 1100, 1280, 1536, 2048):
    run(D, 65536)
