"""Clocks and socket power (rocm-smi, every 0.3 s) while one of the engine's workloads runs for a few seconds (GPU box).
WORK = idle | leapfrog (configs[1] sweeps) | nuts4 | nuts7 (configs[2] transitions at depth 4 / 7) | dense (configs[3] single-step
sweeps; IDHMC_DENSE_LANES=0 for one kernel per sweep) | dense_fused (64 steps per call) | dense_nuts"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import inplacedhmc_jl_amd as pkg
WORK = os.environ.get("WORK", "idle")
SECONDS = float(os.environ.get("SECONDS", 4))
if WORK.startswith("dense"):
    D, C = 256, 16384
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    lam = np.logspace(-2, 0, D)
    P = (Q / lam) @ Q.T; P = 0.5 * (P + P.T)
    mu = np.cos(np.arange(D, dtype=float))
    eng = pkg.Engine(pkg.DenseMVN(mu, P), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
    Sig = (Q * lam) @ Q.T
    eng.set_q(mu + rng.standard_normal((C, D)) @ np.linalg.cholesky(0.5 * (Sig + Sig.T)).T); eng.refresh_momentum(1)
    eng.set_eps(0.05)
else:
    D, C = 1024, 65536
    sig = np.logspace(-1, 1, D); mu = np.sin(np.arange(D, dtype=float))
    eng = pkg.Engine(pkg.DiagGaussian(mu, sigma=sig), C, pkg.default_options(metric_mode=pkg.METRIC_SHARED), seed=1)
    eng.set_minv(sig ** 2)
    eng.random_position(); eng.refresh_momentum(1)
    eng.set_eps({"nuts7": 0.03}.get(WORK, 0.25))
eng.synchronize()
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20).stdout
        f = out.strip().splitlines()[-1].split(",")
        samples.append((f[5].strip("()"), float(f[-1])))
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); units = 0; it = 1; s0 = eng.total_steps()
while time.time() - t0 < SECONDS:
    if WORK == "leapfrog": eng.time_leapfrog(0.1, 500); units += 500 * C
    elif WORK == "dense": eng.time_leapfrog(0.02, 2000); units += 2000 * C
    elif WORK == "dense_fused": eng.leapfrog(0.02, 64); eng.synchronize(); units += 64 * C
    elif WORK in ("nuts4", "nuts7", "dense_nuts"): eng.time_transitions(5, it); it += 5
    else: time.sleep(0.5)
el = time.time() - t0
if WORK in ("nuts4", "nuts7", "dense_nuts"): units = eng.total_steps() - s0
stop = True; th.join()
busy = samples[2:] if len(samples) > 4 else samples
print("%-12s lanes=%s  %.3e steps/s   sclk %s   socket power %.0f W (median of %d samples; first %.0f W)" % (
    WORK, os.environ.get("IDHMC_DENSE_LANES", "auto"), units / el, busy[len(busy) // 2][0], float(np.median([p for _, p in busy])), len(busy), samples[0][1]))
