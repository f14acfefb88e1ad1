"""Arena / state traffic of k_nuts by source (diagnostic build -DIDHMC_BYTES, tools/build_variant.sh bytes "-DIDHMC_BYTES"):
IDHMC_LIB=inplacedhmc.jl_amd/libidhmc_bytes.so EPS=0.25 python tools/nuts_bytes.py  -> one JSON line.
Counts the vectors the wavefronts REQUEST from / send to memory (8 * L bytes each); the PMC figures count what misses L2."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
D = int(os.environ.get("D", 1024)); C = int(os.environ.get("C", 65536)); EPS = float(os.environ.get("EPS", 0.25)); NT = int(os.environ.get("NT", 5))
sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
PERCHAIN = os.environ.get("METRIC", "shared") == "perchain"
eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, pkg.default_options(metric_mode=pkg.METRIC_PER_CHAIN if PERCHAIN else pkg.METRIC_SHARED), seed=1)
eng.set_minv(sig ** 2)
rng = np.random.default_rng(1)
q0 = np.empty((C, D))
for i in range(0, C, 4096):
    q0[i:i + 4096] = mu + sig * rng.standard_normal((min(4096, C - i), D))
eng.set_q(q0); del q0
eng.set_eps(EPS)
for it in range(1, 4):
    eng.nuts_transition(it)
eng.synchronize()
c0 = eng.debug_counters().astype(np.int64)
for it in range(4, 4 + NT):
    eng.nuts_transition(it)
eng.synchronize()
c1 = eng.debug_counters().astype(np.int64)
d = c1 - c0
steps = int(d[0])
names = ["prologue (q, p0 store, per-chain M^-1)", "edge swaps at a change of direction", "regeneration: checkpoints + start point", "level-1 summary",
         "level >= 2 summaries", "whole-tree statistic (far edge's momentum, whole-tree rho)", "epilogue (q, grad l)", "stored candidates"]
by = {n: int(v) for n, v in zip(names, d[2:10])}
tot = sum(by.values())
print(json.dumps({"eps": EPS, "chains": C, "D": D, "metric": "per-chain" if PERCHAIN else "shared", "transitions": NT, "leapfrogs": steps,
                  "mean_depth": float(eng.tree_stats()["depth"].mean()),
                  "requested_bytes_per_leapfrog": tot / steps, "requested_bytes_per_transition": tot / (NT * C),
                  "by_source_bytes_per_leapfrog": {n: v / steps for n, v in by.items()},
                  "by_source_share": {n: v / tot for n, v in by.items()}}))
