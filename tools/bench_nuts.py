"""cfg3-style measurement: full NUTS transitions, 1024-dim diagonal Gaussian, C chains (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D = int(os.environ.get("D", 1024))
C = int(os.environ.get("C", 65536))
EPS = float(os.environ.get("EPS", 0.25))
NT = int(os.environ.get("NT", 10))
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
PERCHAIN = os.environ.get("METRIC", "shared") == "perchain"     # reference semantics: every chain its own M^-1
eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C,
                 pkg.default_options(metric_mode=pkg.METRIC_PER_CHAIN if PERCHAIN else pkg.METRIC_SHARED), seed=1)
eng.set_minv(sig ** 2)
rng = np.random.default_rng(1)
q0 = np.empty((C, D))
for i in range(0, C, 4096):
    q0[i:i + 4096] = mu + sig * rng.standard_normal((min(4096, C - i), D))
eng.set_q(q0); del q0
eng.set_eps(EPS)
print("device MB", eng.device_bytes() / 2**20, flush=True)
for it in range(1, 4):
    eng.nuts_transition(it)
eng.synchronize()
s0 = eng.total_steps()
ms = eng.time_transitions(NT, 3)
s1 = eng.total_steps()
st = eng.tree_stats()
steps = s1 - s0
print(f"C={C} eps={EPS} transitions={NT} ms/transition={ms/NT:.2f} leapfrogs={steps} steps/s={steps/(ms*1e-3):.3e} "
      f"mean depth={st['depth'].mean():.2f} mean steps={st['steps'].mean():.1f} acc={st['acceptance_rate'].mean():.3f} "
      f"equiv GB/s={steps/(ms*1e-3)*49152/1e9:.0f}", flush=True)
print("depth hist", np.bincount(st['depth']).tolist())
dc = eng.debug_counters()
if dc[2:10].sum() > 0:
    names = ["prologue_rest", "leapfrog", "merge", "park", "doubling", "epilogue", "momentum"]
    tot = float(dc[2:9].sum())
    print("cycle shares:", {n: round(float(v) / tot, 3) for n, v in zip(names, dc[2:9])}, "total Gcycles", tot / 1e9,
          "cycles/leaf(all phases)", tot / float(dc[0]))
if os.environ.get("FUSED"):      # the same transitions as ONE launch (idhmc_nuts_transitions)
    eng.nuts_transitions(900, 2)      # (the first launch of this kind pays for the runtime's first memset of that size)
    for n in (NT, NT, 4 * NT, 16 * NT):
        s0 = eng.total_steps()
        ms = eng.time_transitions_fused(n, 1000)
        steps = eng.total_steps() - s0
        print(f"fused, {n} transitions per launch: ms/transition={ms/n:.2f} steps/s={steps/(ms*1e-3):.3e}", flush=True)
