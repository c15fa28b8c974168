"""Reproduction number R_it: oracle self-checks on CPU, HIP parity on GPU."""
import math

import numpy as np
import pytest

from covid19uk_amd import synth
from oracle import rt_oracle as ro
from oracle import seir_oracle as so
from tests import helpers as H


def _draws(case, n, seed):
    rng = np.random.default_rng(seed)
    k = case["k"]
    u = synth.jitter_params(case["u"], n, scale=0.1, seed=seed, T=k.T)
    theta = so.constrain(u)
    ev = np.stack([case["events"]] * n)
    ev[:, :, :, 0] += rng.integers(0, 2, size=ev.shape[:-1]) * 0     # keep feasible
    return theta, ev


def test_ngm_matches_elementwise_loop_and_quirks():
    case = H.build_case("micro_3x5", 7, alpha_t_sd=0.01)
    k = case["k"]
    theta, ev = _draws(case, 1, 7)
    par = so.unpack(theta[0], k.M, k.T)
    state = so.compute_state(k.initial_state, ev[0])
    for t in (0, 1, 3, 4):
        got = ro.next_generation_matrix(t, state[:, t], par, k, stable=True)
        lit = ro.next_generation_matrix(t, state[:, t], par, k)
        assert np.allclose(lit, got, rtol=1e-6, atol=0)       # the literal 1-exp(-r) is good to ~1e-16/r
        b_t = par["alpha_0"] + np.cumsum(par["alpha_t"])
        a = par["alpha_0"] if t == 0 else b_t[min(t, k.T - 2)]      # indexes t, not t-1 (reference quirk)
        period = 1.0 / (1.0 - math.exp(-math.exp(par["gamma0"])))
        for i in range(k.M):
            for j in range(k.M):
                eta = a + par["beta_area"] * k.log_area_c[i] + par["sigma_space"] * par["spatial_effect"][j]
                rate = math.exp(eta) * ((1.0 if i == j else 0.0) + par["psi"] * k.W[t] * k.Cstar[i, j] / k.N[j]) / k.N[i]
                want = -math.expm1(-rate) * state[i, t, 0] * period
                assert abs(got[i, j] - want) <= 1e-12 * abs(want) + 1e-300
    R, Rt = ro.posterior_rit(theta, ev, k)
    assert R.shape == (1, k.T, k.M) and Rt.shape == (1, k.T)
    assert np.allclose(Rt[0], (R[0] * (k.N / k.N.sum())).sum(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,seed", [("micro_3x5", 3, 1), ("ni11", 5, 2), ("micro_17x70", 2, 3), ("uk380", 2, 4)])
def test_hip_rit_matches_oracle(name, n, seed):
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    case = H.build_case(name, seed, alpha_t_sd=0.01)
    theta, ev = _draws(case, n, seed)
    want, _ = ro.posterior_rit(theta, ev, case["k"], stable=True)
    literal, _ = ro.posterior_rit(theta, ev, case["k"])
    with SeirModel(case["cov"], case["init"], max_chains=2) as model:     # n > max_chains: batched inside
        got = model.reproduction_number(theta, ev)
        # the log-prob path still works after the scan workspace was reused
        lp = model.log_prob(so.unconstrain(theta[:1]), ev[:1])
    assert got.shape == want.shape
    err = np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-12 * np.abs(want).max()))
    assert err < 1e-11, err                                   # against the accurately evaluated formula
    assert np.allclose(got, literal, rtol=1e-6, atol=0)       # and within the literal form's own rounding
    ref = so.joint_log_prob(so.unconstrain(theta[0]), ev[0], case["k"], "stable")
    assert abs(lp[0] - ref) <= 1e-9 * abs(ref)


@pytest.mark.gpu
def test_reproduction_number_stage_end_to_end(tmp_path):
    """thin -> reproduction_number on files, as the reference pipeline chains them."""
    import pickle
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd import hdf5io
    from covid19uk_amd.inference import inference as inf
    from covid19uk_amd.posterior.reproduction_number import reproduction_number
    case = H.build_case("ni11", 9, alpha_t_sd=0.01)
    k = case["k"]
    theta, ev = _draws(case, 4, 9)
    data = str(tmp_path / "data.nc")
    inf.write_inference_data(data, case["cov"], case["events"][..., 2])
    T, M = k.T, k.M
    samples = dict(psi=theta[:, 0], sigma_space=theta[:, 1], beta_area=theta[:, 2], gamma0=theta[:, 3],
                   gamma1=theta[:, 4], alpha_0=theta[:, 5], alpha_t=theta[:, 6:6 + T - 1],
                   spatial_effect=theta[:, 6 + T - 1:], seir=ev, initial_state=case["init"])
    pk = str(tmp_path / "thin.pkl")
    with open(pk, "wb") as f:
        pickle.dump(samples, f)
    out = str(tmp_path / "rt.h5")
    r_it, r_t = reproduction_number([data, pk], out)
    want, want_t = ro.posterior_rit(theta, ev, k, stable=True)
    assert np.allclose(r_it, want, rtol=1e-10) and np.allclose(r_t, want_t, rtol=1e-10)
    with hdf5io.File(out, "r") as f:
        assert f.shape("/posterior_predictive/R_it") == (4, T, M)
        assert np.allclose(f.read("/posterior_predictive/R_t"), want_t, rtol=1e-10)


def _last_states(case, n, seed):
    rng = np.random.default_rng(seed)
    k = case["k"]
    st = so.compute_state(k.initial_state, case["events"])[:, -1, :]
    out = np.stack([st] * n)
    out[:, :, 2] += rng.integers(0, 50, size=(n, k.M))
    return out


def test_pressure_components_match_elementwise_definition():
    case = H.build_case("micro_3x5", 2)
    k, cov = case["k"], case["cov"]
    st = _last_states(case, 2, 1)
    psi = np.array([0.3, 1.1])
    wi, be = ro.pressure_components(psi, st, cov.C, cov.N, k.W[-1])
    C = np.array(cov.C, dtype=np.float64)
    np.fill_diagonal(C, 0.0)
    for d in range(2):
        for m in range(k.M):
            I = st[d, :, 2]
            w = I[m] - psi[d] * I[m] / cov.N[m] * k.W[-1] * C[:, m].sum()
            b = psi[d] * k.W[-1] * sum((C[m, j] + C[j, m]) * I[j] / cov.N[j] for j in range(k.M))
            assert abs(wi[d, m] - w / (w + b)) < 1e-13 and abs(be[d, m] - b / (w + b)) < 1e-13
    assert np.allclose(wi + be, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("micro_3x5", 2), ("ni11", 4), ("uk380", 3)])
def test_hip_within_between_matches_oracle(name, n, tmp_path):
    import pickle

    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.inference import inference as inf
    from covid19uk_amd.posterior import within_between as wb
    case = H.build_case(name, 4)
    k, cov = case["k"], case["cov"]
    st = _last_states(case, n, 3)
    psi = np.linspace(0.2, 0.9, n)
    want_w, want_b = ro.pressure_components(psi, st, cov.C, cov.N, k.W[-1])
    got_w, got_b = wb.calc_pressure_components(cov, psi, st, initial_state=case["init"])
    err = max(np.abs(got_w - want_w).max(), np.abs(got_b - want_b).max())
    assert err < 1e-12, err                       # fractions in [0,1]: absolute tolerance
    # file-level entry point
    data = str(tmp_path / "d.npz")
    inf.write_inference_data(data, cov, case["events"][:, :, 2])
    pk = str(tmp_path / "s.pkl")
    with open(pk, "wb") as f:
        pickle.dump(dict(psi=psi, seir=np.stack([case["events"]] * n), initial_state=case["init"]), f)
    w2, b2 = wb.within_between([data, pk], str(tmp_path / "wb.csv"))
    st2 = np.stack([so.compute_state(case["init"], case["events"])[:, -1, :]] * n)
    ww, bb = ro.pressure_components(psi, st2, cov.C, cov.N, k.W[-1])
    # locations without any infection pressure give 0/0 = NaN, in the reference's arithmetic too
    np.testing.assert_allclose(w2, ww, rtol=0, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(b2, bb, rtol=0, atol=1e-12, equal_nan=True)
    lines = open(str(tmp_path / "wb.csv")).read().splitlines()
    assert lines[0].startswith("location,within_mean") and len(lines) == k.M + 1
