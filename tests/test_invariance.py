"""Invariant-distribution tests of the CPU oracle sampler (see tests/invariance_lib.py for the
constructions).  The same tests run against the HIP sampler in tests/test_invariance_gpu.py; the
oracle is what the HIP traces are compared with draw by draw, so it is held to the bar first."""
import math

import numpy as np
import pytest

from oracle import mcmc_oracle as mo
from tests import invariance_lib as IL

CFG_TOY = dict(dmax=2, nmax=2, m=2, occult_nmax=2, num_event_time_updates=3)
CFG_SMALL = dict(dmax=4, nmax=5, m=2, occult_nmax=4, num_event_time_updates=3)


@pytest.fixture(scope="module")
def toy():
    return IL.toy_problem(T=3)


def _run_toy(toy, kernel, R, sweeps, seed, chain_cls=mo.OracleChain):
    rng = np.random.default_rng(seed)
    starts = rng.choice(len(toy["prob"]), size=R, p=toy["prob"])
    ends = np.empty((R,) + toy["states"].shape[1:])
    for r in range(R):
        ch = chain_cls(toy["k"], CFG_TOY, toy["u"], toy["states"][starts[r]], seed=seed, chain_id=r,
                       disable=IL.only(kernel))
        for _ in range(sweeps):
            ch.sweep_once()
        ends[r] = ch.events
    return starts, IL.state_indices(toy, ends)


@pytest.mark.parametrize("kernel", IL.KERNELS + ("all",))
def test_event_kernel_leaves_the_enumerated_posterior_invariant(toy, kernel):
    """Exact draws from the tabulated p(z | theta, y), then 3 sweeps x 3 scans of ONE kernel: the end
    states must still follow the table (chi-square over pooled cells, independent replicates)."""
    R = 500
    starts, ends = _run_toy(toy, kernel, R, 3, seed=11)
    assert (starts != ends).mean() > 0.15, "the kernel hardly moves: the test would have no power"
    stat, dof, p = IL.chi_square(ends, toy["prob"])
    assert p > 1e-3, (kernel, stat, dof, p)


class _NoCorrection(mo.OracleChain):
    """A deliberately wrong sampler: the proposal-density correction log q(rev) - log q(fwd) is dropped."""
    def _mh(self, new_events, valid, logq, logu):
        return super()._mh(new_events, valid, 0.0, logu)


def test_the_chi_square_test_has_teeth(toy):
    """The same experiment with the Hastings correction removed must be rejected decisively --
    otherwise a green invariance test would say nothing about the bookkeeping of log q."""
    _, ends = _run_toy(toy, "all", 500, 3, seed=11, chain_cls=_NoCorrection)
    stat, dof, p = IL.chi_square(ends, toy["prob"])
    assert p < 1e-8, (stat, dof, p)


@pytest.mark.parametrize("kernel", ["all", "occult/E->I"])
def test_event_kernels_preserve_the_joint_distribution_of_simulated_epidemics(kernel):
    """(z, y) ~ p(. | theta) by forward simulation, then sweeps of the event kernels given y: every
    moment of (z, y) must be unchanged (paired z-tests, 12 statistics, independent replicates)."""
    case = IL.small_population_case()
    R, sweeps = 400, 3
    rng = np.random.default_rng(23)
    ev0 = IL.simulate_numpy(case, R, rng)
    ev1 = np.empty_like(ev0)
    moved = 0
    for r in range(R):
        ch = mo.OracleChain(case["k"], CFG_SMALL, case["u"], ev0[r], seed=5, chain_id=r, disable=IL.only(kernel))
        for _ in range(sweeps):
            ch.sweep_once()
        ev1[r] = ch.events
        moved += int((ev1[r] != ev0[r]).any())
    assert moved > 0.3 * R
    assert np.array_equal(ev1[..., 2], ev0[..., 2]), "the observed I->R events must never move"
    z = IL.paired_z(IL.event_statistics(ev0, case["init"]), IL.event_statistics(ev1, case["init"]))
    assert np.abs(z).max() < 4.2, dict(zip(IL.STAT_NAMES, np.round(z, 2)))


def test_hmc_recovers_the_prior_when_the_likelihood_is_flat():
    """No infectives and no events: the likelihood is constant in theta, so the target of the HMC update is
    the prior of model_spec.py:140-198 (through the bijector of inference.py:525-535) -- whose moments
    are known exactly.  Checks momentum scaling, energy, accept rule, Jacobian and prior terms together."""
    case = IL.small_population_case(M=3, T=5)
    k = case["k"]
    import dataclasses
    init = np.stack([k.N, np.zeros(k.M), np.zeros(k.M), np.zeros(k.M)], axis=-1)
    k0 = dataclasses.replace(k, initial_state=init) if dataclasses.is_dataclass(k) else None
    if k0 is None:
        from oracle import seir_oracle as so
        cov = case["cov"]
        k0 = so.make_constants(cov.C, cov.N, cov.W, cov.weekday, cov.area, cov.adjacency, init)
    events = np.zeros((k.M, k.T, 3))
    P, T, M = k0.P, k0.T, k0.M
    # mass matrix = prior variances in the unconstrained space (approximately, for the two softplus ones)
    var = np.ones(P)
    var[0], var[1], var[2], var[3], var[4], var[5] = 0.5, 0.5, 1.0, 1.0e4, 1.0e4, 100.0
    var[6:6 + T - 1] = 0.005 ** 2
    Q = np.diag(case["cov"].adjacency.sum(1)) - 0.25 * case["cov"].adjacency
    var[6 + T - 1:] = np.diag(np.linalg.inv(Q))
    n_chains, n = 6, 500
    draws = []
    for c in range(n_chains):
        ch = mo.OracleChain(k0, CFG_SMALL, np.zeros(P), events, seed=3, chain_id=c,
                            disable=IL.KERNELS, num_leapfrog_steps=8)
        ch.eps, ch.var = 0.35, var
        acc = 0
        th = np.empty((n, P))
        for i in range(n):
            o = ch.sweep_once()
            acc += o["hmc"]["is_accepted"]
            th[i] = o["theta"]
        assert acc > 0.6 * n
        draws.append(th[100:])
    d = np.stack(draws)                                     # [chains, draws, P]
    cm = d.mean(axis=1)                                     # chain means: independent replicates
    def z(col, mean):
        v = cm[:, col]
        return (v.mean() - mean) / (v.std(ddof=1) / math.sqrt(n_chains))
    zs = {"psi": z(0, 0.3), "sigma_space": z(1, 0.1 * math.sqrt(2 / math.pi)), "beta_area": z(2, 0.0),
          "gamma0": z(3, 0.0), "gamma1": z(4, 0.0), "alpha_0": z(5, 0.0), "alpha_t[0]": z(6, 0.0),
          "spatial[0]": z(6 + T - 1, 0.0)}
    # Student t with 5 dof: |t| < 6.9 is p > 1e-3 two-sided
    assert max(abs(v) for v in zs.values()) < 6.9, zs
    # second moments against the prior's, pooled over chains (ratio within 25 %: 2400 correlated draws)
    pooled = d.reshape(-1, P)
    assert abs(pooled[:, 0].var() / 0.03 - 1) < 0.3                       # Gamma(3, rate 10): variance 3/100
    assert abs(pooled[:, 2].var() / 1.0 - 1) < 0.3
    assert abs(pooled[:, 3].var() / 1.0e4 - 1) < 0.3
    assert abs(pooled[:, 5].var() / 100.0 - 1) < 0.3
    assert abs(pooled[:, 6].var() / 0.005 ** 2 - 1) < 0.3
    assert abs(pooled[:, 6 + T - 1].var() / np.linalg.inv(Q)[0, 0] - 1) < 0.3
