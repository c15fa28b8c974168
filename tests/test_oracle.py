"""Pins the CPU oracle (the reference ships nothing for this path: parity
unpinned, see oracle/seir_oracle.py header).  Independent checks:
scipy.stats.binom, mpmath at 50 digits, a hand-written micro case, finite
differences, structural invariants, and the C restatement against NumPy."""
import math

import numpy as np
import pytest
from scipy import stats

from oracle import seir_oracle as so
from tests import helpers as H

RTOL = 1e-9     # stated fp64 tolerance on the summed log-prob


def _rand_u(case, seed, scale=0.2):
    rng = np.random.default_rng(seed)
    u = case["u"] + scale * rng.normal(size=case["u"].shape)
    T = case["k"].T
    u[6:6 + T - 1] = 0.005 * rng.normal(size=T - 1)
    return u


def test_constants_cstar_convention():
    # model_spec.py:216-219: zero diag, C + C^T, diag = -column sums of C
    C = np.array([[7.0, 2.0, 3.0], [4.0, 9.0, 6.0], [8.0, 1.0, 5.0]])
    k = so.make_constants(C, [10, 20, 30], [1, 1], [1, 0], [1e8, 2e8, 3e8],
                          [[0, 1, 0], [1, 0, 1], [0, 1, 0]], np.zeros((3, 4)))
    assert np.array_equal(k.Cstar, np.array([[-12.0, 6.0, 11.0], [6.0, -3.0, 7.0], [11.0, 7.0, -9.0]]))
    assert np.allclose(k.weekday_c, [0.5, -0.5])
    assert abs(k.log_area_c.sum()) < 1e-12
    assert np.allclose(k.Q, np.array([[1, -.25, 0], [-.25, 2, -.25], [0, -.25, 1]]))


def test_compute_state_is_exclusive_cumsum():
    ev = np.zeros((1, 3, 3))
    ev[0, :, 0] = [2, 1, 0]
    ev[0, :, 1] = [0, 1, 1]
    ev[0, :, 2] = [1, 0, 1]
    st = so.compute_state(np.array([[10.0, 1.0, 2.0, 0.0]]), ev)
    assert np.array_equal(st[0], [[10, 1, 2, 0], [8, 3, 1, 1], [7, 3, 2, 1]])
    stc = so.compute_state(np.array([[10.0, 1.0, 2.0, 0.0]]), ev, closed=True)
    assert np.array_equal(stc[0, -1], [7, 2, 2, 2])


@pytest.mark.parametrize("name,seed", [("micro_1x1", 1), ("micro_2x3", 2), ("micro_3x5", 3), ("ni11", 4)])
def test_closed_form_equals_literal_multinomial(name, seed):
    c = H.build_case(name, seed)
    par = so.unpack(so.constrain(_rand_u(c, seed)), c["k"].M, c["k"].T)
    a = so.seir_log_prob_literal(par, c["k"], c["events"])
    b = so.seir_log_prob(par, c["k"], c["events"], "reference")
    assert math.isfinite(a)
    assert abs(a - b) <= 1e-12 * abs(a)


@pytest.mark.parametrize("name,seed", [("micro_2x3", 5), ("ni11", 6)])
def test_seir_term_against_scipy_binom(name, seed):
    c = H.build_case(name, seed)
    k, ev = c["k"], c["events"]
    par = so.unpack(so.constrain(_rand_u(c, seed)), k.M, k.T)
    st = so.compute_state(k.initial_state, ev)
    lam, r_ei, r_ir = so.transition_rates(par, k, st)
    want = stats.binom.logpmf(ev[..., 0], st[..., 0], -np.expm1(-lam)).sum()
    want += stats.binom.logpmf(ev[..., 1], st[..., 1], -np.expm1(-r_ei)).sum()
    want += stats.binom.logpmf(ev[..., 2], st[..., 2], -np.expm1(-r_ir)[None, :]).sum()
    for form in ("reference", "stable"):
        got = so.seir_log_prob(par, k, ev, form)
        assert abs(got - want) <= RTOL * abs(want)


def test_hand_case_single_cell():
    # M=1, T=1: log Binom(S, k; 1-exp(-lam)) etc. written out by hand.
    N, S, E, I = 1000.0, 990.0, 4.0, 6.0
    C = np.array([[5.0]])
    k = so.make_constants(C, [N], [1.0], [1.0], [2e8], [[1.0]], [[S, E, I, 0.0]])
    ev = np.array([[[3.0, 2.0, 1.0]]])
    par = dict(psi=0.4, sigma_space=0.2, beta_area=0.3, gamma0=-1.2, gamma1=0.5, alpha_0=-0.7,
               alpha_t=np.zeros(0), spatial_effect=np.array([0.9]))
    # single LAD: C zero-diag -> Cstar = [[0]]; log_area_c = weekday_c = 0
    lam = math.exp(-0.7 + 0.2 * 0.9) * I / N + 1e-9
    r_ir = math.exp(-1.2)
    want = (stats.binom.logpmf(3, 990, 1 - math.exp(-lam)) + stats.binom.logpmf(2, 4, 1 - math.exp(-0.28))
            + stats.binom.logpmf(1, 6, 1 - math.exp(-r_ir)))
    got = so.seir_log_prob(par, k, ev)
    assert abs(got - want) < 1e-10


@pytest.mark.parametrize("name,seed", [("micro_1x1", 11), ("micro_2x2", 12), ("micro_3x4", 13)])
def test_joint_log_prob_against_mpmath(name, seed):
    c = H.build_case(name, seed)
    u = _rand_u(c, seed)
    want = float(so.joint_log_prob_mp(u, c["events"], c["k"], dps=50))
    for form in ("reference", "stable"):
        got = so.joint_log_prob(u, c["events"], c["k"], form)
        assert abs(got - want) <= RTOL * abs(want), (form, got, want)
    got, _ = so.joint_log_prob_and_grad(u, c["events"], c["k"])
    assert abs(got - want) <= 1e-12 * abs(want)


def test_literal_car_prior_matches_precision_form():
    c = H.build_case("ni11", 21)
    par = so.unpack(so.constrain(_rand_u(c, 21)), c["k"].M, c["k"].T)
    a = so.prior_log_prob(par, c["k"], literal_car=True)
    b = so.prior_log_prob(par, c["k"], literal_car=False)
    assert abs(a - b) <= 1e-11 * abs(a)


@pytest.mark.parametrize("name,seed", [("micro_3x5", 31), ("ni11", 32)])
def test_gradient_against_finite_differences(name, seed):
    c = H.build_case(name, seed, alpha_t_sd=0.005)
    u = _rand_u(c, seed)
    k, ev = c["k"], c["events"]
    lp, g = so.joint_log_prob_and_grad(u, ev, k)
    fd = np.zeros_like(g)
    for i in range(len(u)):
        h = 1e-6 * max(1.0, abs(u[i]))
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        fd[i] = (so.joint_log_prob(up, ev, k, "stable") - so.joint_log_prob(um, ev, k, "stable")) / (2 * h)
    scale = np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
    assert np.max(np.abs(g - fd) / scale) < 1e-6


def test_infeasible_events_give_minus_inf():
    c = H.build_case("micro_2x3", 41)
    ev = c["events"].copy()
    ev[0, 1, 1] = 1e6       # more E->I events than people in E
    assert so.joint_log_prob(c["u"], ev, c["k"]) == -np.inf
    assert so.seir_log_prob_literal(so.unpack(c["theta"], 2, 3), c["k"], ev) == -np.inf


def test_relabelling_lads_leaves_log_prob_unchanged():
    c = H.build_case("ni11", 42)
    cov, k = c["cov"], c["k"]
    perm = np.random.default_rng(0).permutation(k.M)
    k2 = so.make_constants(cov.C[np.ix_(perm, perm)], cov.N[perm], cov.W, cov.weekday, cov.area[perm],
                           cov.adjacency[np.ix_(perm, perm)], c["init"][perm])
    u = _rand_u(c, 42)
    u2 = u.copy()
    u2[6 + k.T - 1:] = u[6 + k.T - 1:][perm]
    a = so.joint_log_prob(u, c["events"], k)
    b = so.joint_log_prob(u2, c["events"][perm], k2)
    assert abs(a - b) <= 1e-11 * abs(a)


def test_psi_zero_decouples_lads():
    c = H.build_case("micro_3x5", 43)
    k, ev = c["k"], c["events"]
    par = so.unpack(c["theta"].copy(), k.M, k.T)
    par["psi"] = 0.0
    tot = so.seir_log_prob(par, k, ev)
    parts = 0.0
    for m in range(k.M):
        km = so.make_constants(np.zeros((1, 1)), [k.N[m]], k.W, c["cov"].weekday, [1.0], [[1.0]],
                               k.initial_state[m:m + 1])
        km.log_area_c = k.log_area_c[m:m + 1]
        pm = dict(par, spatial_effect=par["spatial_effect"][m:m + 1])
        parts += so.seir_log_prob(pm, km, ev[m:m + 1])
    assert abs(tot - parts) <= 1e-12 * abs(tot)


def test_negative_rate_with_zero_events_follows_multiply_no_nan():
    # lam < 0 and k_se = 0: TFP's multiply_no_nan drops the nan log p term.
    k = so.make_constants(np.array([[0.0, 50.0], [50.0, 0.0]]), [100.0, 100.0], [1.0], [1.0], [1e8, 1e8],
                          [[0, 1], [1, 0]], [[90.0, 0.0, 10.0, 0.0], [100.0, 0.0, 0.0, 0.0]])
    par = dict(psi=5.0, sigma_space=0.1, beta_area=0.0, gamma0=-1.0, gamma1=0.0, alpha_0=0.0,
               alpha_t=np.zeros(0), spatial_effect=np.zeros(2))
    ev = np.zeros((2, 1, 3))
    st = so.compute_state(k.initial_state, ev)
    lam, _, _ = so.transition_rates(par, k, st)
    assert lam[0, 0] < 0
    assert math.isfinite(so.seir_log_prob(par, k, ev))
    ev[0, 0, 0] = 1.0
    assert math.isnan(so.seir_log_prob(par, k, ev))


@pytest.mark.parametrize("name,seed", [("micro_2x3", 51), ("ni11", 52), ("uk380", 53)])
def test_c_restatement_matches_numpy(name, seed):
    c = H.build_case(name, seed, alpha_t_sd=0.005)
    u = _rand_u(c, seed, 0.1)
    want, gw = so.joint_log_prob_and_grad(u, c["events"], c["k"])
    got, gg = H.c_oracle_eval(c["k"], u, c["events"], stable=1, want_grad=True)
    assert abs(got - want) <= 1e-12 * abs(want)
    assert np.max(np.abs(gg - gw) / np.maximum(np.abs(gw), 1e-6 * np.abs(gw).max())) < 1e-9
    ref = so.joint_log_prob(u, c["events"], c["k"], "reference")
    got0 = H.c_oracle_eval(c["k"], u, c["events"], stable=0)
    assert abs(got0 - ref) <= 1e-12 * abs(ref)
    assert abs(got0 - got) <= RTOL * abs(ref)
