"""TEST INFRASTRUCTURE: a SECOND, independent restatement of the reference's NUTS transition, in plain numpy.

Written directly from the Julia (file:line of /root/reference cited per function), recursive like the reference
(`adjacent_tree` calls itself), with the reference's own test affordance as its only inputs: the momentum `p` and the
`directions` are injected (src/NUTS.jl:251-252, kwargs `p=`, `directions=`) and every `randexp(rng, Float64)` the
reference would draw (src/NUTS.jl:33) is taken, in call order, from a caller-supplied iterator.  It shares NO code with
oracle/idhmc_oracle.c or with the HIP kernels: numpy's own elementwise arithmetic (no fma), numpy's `dot`, libm's
`log1p` / `exp`.  Floating point therefore differs from the C oracle in the last bits (documented tolerance 1e-12 on
draws); tree decisions (depth, steps, termination) must be IDENTICAL wherever the oracle reports a decision margin
above 1e-9 (tests/test_numpy_restatement.py).  Only tests/ may import this module.

`@avx` associations assumed here (LoopVectorization reassociates reductions; the order is unspecified in the reference):
  * kinetic_energy (src/kinetic_energy.jl:19-22): (p*M^-1)*p per element, summed by numpy.dot-like pairwise sum;
  * is_turning (src/NUTS.jl:154-158): plain dot products;
  * leapfrog (src/kinetic_energy.jl:146-150): eps*M^-1*p_m evaluated left to right, (eps*M^-1)*p_m.
"""
import math

import numpy as np

REACHED_MAX_DEPTH = (1, 0)                                  # src/tree.jl:300


def logaddexp(x, y):                                        # src/InplaceDHMC.jl:27-30
    if not (math.isfinite(x) and math.isfinite(y)):
        return max(x, y)
    return x + math.log1p(math.exp(y - x)) if x > y else y + math.log1p(math.exp(x - y))


class DiagGaussianDensity:
    """l(q) = -1/2 sum tau (q - mu)^2: the user density behind logdensity_and_gradient! (src/kinetic_energy.jl:73)"""

    def __init__(self, mu, tau):
        self.mu, self.tau = np.asarray(mu, float), np.asarray(tau, float)

    def logdensity_and_gradient(self, q):
        d = q - self.mu
        t = self.tau * d
        return -0.5 * float(np.sum(t * d)), -t


class DenseGaussianDensity:
    """l(q) = -1/2 (q - mu)' P (q - mu), P symmetric: the non-separable benchmark density (BASELINE.json configs[3]); numpy's own
    matrix-vector product, i.e. NOT the engine's summation order (one fma chain per row, ascending column)"""

    def __init__(self, mu, prec):
        self.mu, self.prec = np.asarray(mu, float), np.asarray(prec, float)

    def logdensity_and_gradient(self, q):
        d = q - self.mu
        t = self.prec @ d
        return -0.5 * float(t @ d), -t


class PhasePoint:                                           # src/hamiltonian.jl:237-276
    __slots__ = ("q", "lq", "grad", "p")

    def __init__(self, q, lq, grad, p):
        self.q, self.lq, self.grad, self.p = q, lq, grad, p


class Hamiltonian:                                          # src/hamiltonian.jl:206-220
    def __init__(self, density, minv):
        self.l, self.minv = density, np.asarray(minv, float)

    def kinetic_energy(self, p):                            # src/kinetic_energy.jl:14-24
        return 0.5 * float(np.sum((p * self.minv) * p))

    def psharp(self, p):                                    # calculate_p#, src/kinetic_energy.jl:39-46
        return self.minv * p

    def evaluate(self, q):                                  # evaluate_l!, src/kinetic_energy.jl:72-85
        lq, g = self.l.logdensity_and_gradient(q)
        return (lq, g) if math.isfinite(lq) else (-math.inf, q)

    def logdensity(self, z):                                # src/kinetic_energy.jl:107-112
        if not math.isfinite(z.lq):
            return -math.inf
        K = self.kinetic_energy(z.p)
        return z.lq - (K if math.isfinite(K) else math.inf)

    def leapfrog(self, z, eps):                             # src/kinetic_energy.jl:126-163
        eh = 0.5 * eps
        pm = z.p + eh * z.grad                              # loop A :146-150
        q1 = z.q + (eps * self.minv) * pm
        lq, g = self.evaluate(q1)                           # :154
        return PhasePoint(q1, lq, g, pm + eh * g)           # loop B :159-161


class Trajectory:                                           # TrajectoryNUTS, src/NUTS.jl:5-16
    def __init__(self, H, pi0, eps, min_delta, randexp):
        self.H, self.pi0, self.eps, self.min_delta, self.randexp = H, pi0, eps, min_delta, randexp


def rand_bool_logprob(traj, logprob):                       # src/NUTS.jl:32-34 (no draw when logprob >= 0)
    return logprob >= 0 or next(traj.randexp) > -logprob


def combine_proposals_and_logweights(traj, z1, z2, w1, w2, is_doubling):   # src/tree.jl:238-245, :261-263, NUTS.jl:36-45
    w = logaddexp(w1, w2)
    logprob2 = w2 - (w1 if is_doubling else w)
    return (z2 if rand_bool_logprob(traj, logprob2) else z1), w


def combine_acc(a, b):                                      # src/NUTS.jl:68-70
    return (logaddexp(a[0], b[0]), a[1] + b[1])


def leaf(traj, z, is_initial):                              # src/NUTS.jl:176-191, :76-78, :113-116
    delta = 0.0 if is_initial else traj.H.logdensity(z) - traj.pi0
    v = (-math.inf, 0) if is_initial else ((delta if delta < 0 else 0.0), 1)
    if delta < traj.min_delta:
        return (z, delta, None), v, True
    ps = traj.H.psharp(z.p)
    return (z, delta, (ps, ps, z.p)), v, False              # tau = (p#-, p#+, rho)


def combine_turn_in_direction(t1, t2, is_forward):          # src/tree.jl:230-236, src/NUTS.jl:118-145
    x, y = (t1, t2) if is_forward else (t2, t1)
    return (x[0], y[1], x[2] + y[2])


def is_turning(tau):                                        # src/NUTS.jl:148-170
    psm, psp, rho = tau
    return bool((float(np.dot(rho, psm)) < 0) | (float(np.dot(rho, psp)) < 0))


def adjacent_tree(traj, z, i, depth, is_forward):           # src/tree.jl:321-366
    i1 = i + (1 if is_forward else -1)
    if depth == 0:
        z1 = traj.H.leapfrog(z, traj.eps if is_forward else -traj.eps)      # move, src/NUTS.jl:18-21
        (zeta, w, tau), v, invalid = leaf(traj, z1, False)
        return (zeta, w, tau, z1, i1), v, (invalid, (i1, i1))
    tm, vm, (invalid, it) = adjacent_tree(traj, z, i, depth - 1, is_forward)
    if invalid:
        return tm, vm, (invalid, it)
    zeta_m, w_m, tau_m, z_m, i_m = tm
    tp, vp, (invalid, it) = adjacent_tree(traj, z_m, i_m, depth - 1, is_forward)
    v = combine_acc(vm, vp)
    if invalid:
        return tp, v, (invalid, it)
    zeta_p, w_p, tau_p, z_p, i_p = tp
    tau = combine_turn_in_direction(tau_m, tau_p, is_forward)
    if is_turning(tau):
        return tp, v, (True, (i1, i_p))
    zeta, w = combine_proposals_and_logweights(traj, zeta_m, zeta_p, w_m, w_p, False)
    return (zeta, w, tau, z_p, i_p), v, (False, REACHED_MAX_DEPTH)


def sample_trajectory(traj, z, max_depth, directions):      # src/tree.jl:382-444
    (zeta, w, tau), v, _ = leaf(traj, z, True)
    z_minus = z_plus = z
    depth, termination, i_minus, i_plus = 0, REACHED_MAX_DEPTH, 0, 0
    while depth < max_depth:
        is_forward, directions = bool(directions & 1), directions >> 1       # next_direction :152-155
        zi, ii = (z_plus, i_plus) if is_forward else (z_minus, i_minus)
        t1, v1, (invalid, it) = adjacent_tree(traj, zi, ii, depth, is_forward)
        v = combine_acc(v, v1)
        if invalid:
            termination = it
            break
        zeta1, w1, tau1, z1, i1 = t1
        if is_forward:
            z_plus, i_plus = z1, i1
        else:
            z_minus, i_minus = z1, i1
        zeta, w = combine_proposals_and_logweights(traj, zeta, zeta1, w, w1, True)
        depth += 1
        tau = combine_turn_in_direction(tau, tau1, is_forward)
        if is_turning(tau):
            termination = (i_minus, i_plus)
            break
    return zeta, v, termination, depth


def sample_tree(H, q, p, eps, directions, randexp, max_depth=10, min_delta=-1000.0):   # src/NUTS.jl:251-264
    """One transition from position q with injected momentum and directions.  Returns (new q, stats dict)."""
    lq, g = H.evaluate(np.asarray(q, float))
    z = PhasePoint(np.asarray(q, float), lq, g, np.asarray(p, float))
    traj = Trajectory(H, H.logdensity(z), eps, min_delta, randexp)
    zeta, v, termination, depth = sample_trajectory(traj, z, max_depth, int(directions))
    a = math.exp(v[0]) / v[1]                               # acceptance_rate, src/NUTS.jl:84
    return zeta.q, {"pi": H.logdensity(zeta), "acceptance_rate": a if a < 1 else 1.0,
                    "term_left": termination[0], "term_right": termination[1], "depth": depth, "steps": v[1]}
