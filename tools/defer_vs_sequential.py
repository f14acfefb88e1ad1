"""Round-3 check at scale: the deferred tree bookkeeping (nuts_replay) against the sequential form it replaced (-DIDHMC_NUTS_DEFER=0, the form rounds 1-2
validated against the oracle), on far more trees than the CPU oracle can visit: 65 536 chains x T transitions per setting.  Run once per library
(IDHMC_LIB=...), prints one SHA-256 per setting over (q, lq, pi, tree records); the two runs must print the same lines.
  python tools/defer_vs_sequential.py > a.txt; IDHMC_LIB=inplacedhmc.jl_amd/libidhmc_seq.so python tools/defer_vs_sequential.py > b.txt; diff a.txt b.txt"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import inplacedhmc_jl_amd as pkg
C = int(os.environ.get("C", 65536))
settings = [  # D, eps, start, metric, T, max_depth
    (1024, 0.25, "posterior", "shared", 6, 10), (1024, 0.03, "posterior", "shared", 3, 10), (1024, 0.005, "posterior", "shared", 1, 10),
    (1024, 0.23, "uniform", "perchain", 4, 10),      # the first warm-up transitions: deep trees that stop everywhere
    (1024, 0.6, "posterior", "shared", 6, 10),       # large stepsize: divergences, shallow trees
    (200, 0.02, "uniform", "shared", 4, 10), (512, 0.05, "posterior", "perchain", 4, 8), (130, 0.004, "posterior", "shared", 2, 10),
    (1024, 0.25, "posterior", "shared", 4, 3),       # max depth reached all the time
]
for D, eps, start, metric, T, md in settings:
    sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
    opt = pkg.default_options(max_depth=md, metric_mode=pkg.METRIC_SHARED if metric == "shared" else pkg.METRIC_PER_CHAIN)
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, opt, seed=99)
    if start == "posterior":
        eng.set_minv(sig ** 2)
        rng = np.random.default_rng(3)
        q0 = np.empty((C, D))
        for i in range(0, C, 4096):
            q0[i:i + 4096] = mu + sig * rng.standard_normal((min(4096, C - i), D))
        eng.set_q(q0); del q0
    else:
        eng.random_position()
    eng.set_eps(eps)
    h = hashlib.sha256()
    depths = np.zeros(16, dtype=np.int64); kinds = np.zeros(3, dtype=np.int64)
    for it in range(1, T + 1):
        eng.nuts_transition(it)
        st = eng.tree_stats()
        h.update(st.tobytes()); h.update(eng.lq.tobytes()); h.update(eng.logdensity().tobytes())
        depths += np.bincount(st["depth"], minlength=16)[:16]
        kinds += np.array([((st["term_left"] == 1) & (st["term_right"] == 0)).sum(), (st["term_left"] == st["term_right"]).sum(),
                           ((st["term_left"] != st["term_right"]) & ~((st["term_left"] == 1) & (st["term_right"] == 0))).sum()])
    h.update(eng.q.tobytes())
    eng.close()
    print("D=%d eps=%g start=%s metric=%s T=%d max_depth=%d  depths %s  max_depth/divergent/turning %s  sha256 %s"
          % (D, eps, start, metric, T, md, depths[:md + 1].tolist(), kinds.tolist(), h.hexdigest()[:32]), flush=True)
