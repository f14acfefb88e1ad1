"""configs[3]: 256-dim dense MVN, 16 384 chains: fixed-eps leapfrog sweeps and NUTS transitions (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D, C = 256, int(os.environ.get("C", 16384))
rng = np.random.default_rng(7)
Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
lam = np.logspace(-2, 0, D)
P = (Q / lam) @ Q.T; P = 0.5 * (P + P.T)
mu = np.cos(np.arange(D, dtype=float))
eng = pkg.Engine(pkg.DenseMVN(mu, P), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
Sig = (Q * lam) @ Q.T
q0 = mu + rng.standard_normal((C, D)) @ np.linalg.cholesky(0.5 * (Sig + Sig.T)).T
eng.set_q(q0); eng.refresh_momentum(1)
eps = 0.02
if not os.environ.get("NUTS_ONLY"):       # (diagnostic builds: the leapfrog kernel's stamps share the debug counters)
    eng.time_leapfrog(eps, 20)
    ms = min(eng.time_leapfrog(eps, 500) for _ in range(3))
    print(f"dense leapfrog: ms/sweep={ms:.3f} chain-steps/s={C/ms*1e3:.3e} GFLOP/s(2D^2)={C/ms*1e3*2*D*D/1e9:.0f} state GB/s={C/ms*1e3*6*D*8/1e9:.0f}", flush=True)
    import time
    NS = 64
    eng.leapfrog(eps, NS); eng.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); eng.leapfrog(eps, NS); eng.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"dense leapfrog, {NS} steps per call (state on chip): ms/step={best/NS*1e3:.4f} chain-steps/s={C*NS/best:.3e} "
          f"GFLOP/s(2D^2)={C*NS/best*2*D*D/1e9:.0f}", flush=True)
eng.set_q(q0); eng.refresh_momentum(1)
eng.set_eps(0.05)
for it in (1, 2):
    eng.nuts_transition(it)
s0 = eng.total_steps()
ms = eng.time_transitions(5, 2)
steps = eng.total_steps() - s0
st = eng.tree_stats()
print(f"dense NUTS: ms/transition={ms/5:.2f} steps/s={steps/ms*1e3:.3e} mean depth={st['depth'].mean():.2f} acc={st['acceptance_rate'].mean():.3f}")
dc = eng.debug_counters()
if dc[2:10].sum() > 0:      # diagnostic build only (tools/stamps.sh)
    names = ["prologue_rest", "leapfrog", "merge", "park", "doubling", "epilogue", "momentum"]
    tot = float(dc[2:9].sum())
    print("cycle shares:", {n: round(float(v) / tot, 3) for n, v in zip(names, dc[2:9])}, "cycles/leaf(all phases)", tot / float(dc[0]))
for n in (5, 20, 50):
    eng.set_q(q0); eng.refresh_momentum(1)
    eng.nuts_transitions(1, 2)
    s0 = eng.total_steps()
    ms = eng.time_transitions_fused(n, 100)
    steps = eng.total_steps() - s0
    print(f"dense NUTS, {n} transitions per launch: ms/transition={ms/n:.2f} steps/s={steps/ms*1e3:.3e}")
