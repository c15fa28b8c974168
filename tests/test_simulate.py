"""Chain-binomial forward simulation: oracle pinned against scipy on CPU,
HIP (`seir_simulate`) against the oracle on the shared Philox stream on GPU."""
import datetime

import numpy as np
import pytest
from scipy import stats

from covid19uk_amd import synth
from covid19uk_amd.posterior import predict as pp
from oracle import mcmc_oracle as mo
from oracle import seir_oracle as so
from oracle import sim_oracle as sim
from tests import helpers as H

BINOMIAL_GRID = [(50, 0.1), (1000, 0.004), (100000, 0.0005), (1000, 0.3), (40, 0.9), (5000, 0.7), (3, 0.5),
                 (1, 0.25), (250000, 4.1e-5), (9, 0.999)]


def _oracle_binomial(n, p, seed, cell):
    return sim.binomial(n, p, lambda att: tuple(float(x[0]) for x in mo.rng_uniform2(seed, 0, cell, sim.RS_SIM_BASE, att)))


@pytest.mark.parametrize("n,p", BINOMIAL_GRID)
def test_oracle_binomial_distribution_matches_scipy(n, p):
    N = 3000
    xs = np.array([_oracle_binomial(n, p, 11, i) for i in range(N)])
    assert xs.min() >= 0 and xs.max() <= n
    # chi-square on bins of expected count >= 10
    lo, hi = int(stats.binom.ppf(1e-4, n, p)), int(stats.binom.ppf(1 - 1e-4, n, p))
    edges = list(range(lo, hi + 2))
    probs = np.diff(stats.binom.cdf(np.array(edges) - 1, n, p))
    probs = np.concatenate([[stats.binom.cdf(lo - 1, n, p)], probs, [stats.binom.sf(hi, n, p)]])
    counts = np.concatenate([[np.sum(xs < lo)], [np.sum(xs == v) for v in edges[:-1]], [np.sum(xs > hi)]])
    # merge sparse bins
    E, O, e_acc, o_acc = [], [], 0.0, 0
    for e, o in zip(probs * N, counts):
        e_acc += e
        o_acc += o
        if e_acc >= 10:
            E.append(e_acc); O.append(o_acc); e_acc, o_acc = 0.0, 0
    if E:
        E[-1] += e_acc; O[-1] += o_acc
    if len(E) >= 2:
        chi2 = float(np.sum((np.array(O) - np.array(E)) ** 2 / np.array(E)))
        assert stats.chi2.sf(chi2, len(E) - 1) > 1e-4, (n, p, chi2, len(E))
    assert abs(xs.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / N) + 1e-12


def test_oracle_binomial_edges():
    u = lambda att: (0.3, 0.6)
    assert sim.binomial(0, 0.5, u) == 0
    assert sim.binomial(10, 0.0, u) == 0
    assert sim.binomial(10, -1.0, u) == 0
    assert sim.binomial(10, float("nan"), u) == 0
    assert sim.binomial(10, 1.0, u) == 10
    assert 0 <= sim.binomial(10, 0.5, u) <= 10


def test_log_baseline_indexing():
    rng = np.random.default_rng(0)
    a0 = rng.normal(size=3)
    at = rng.normal(size=(3, 9)) * 0.1
    for init_step, S in ((0, 6), (4, 12), (9, 3), (20, 2)):
        got = pp.log_baseline_path(a0, at, init_step, S)
        for d in range(3):
            want = sim.log_baseline_path(a0[d], at[d], init_step, S)
            assert np.array_equal(got[d], want)
    # t = 0 -> alpha_0; t >= 1 -> b[t-1]; beyond the end the last value holds (model_spec.py:245-256)
    b = a0[0] + np.cumsum(at[0])
    p = pp.log_baseline_path(a0[:1], at[:1], 0, 12)[0]
    assert p[0] == a0[0] and p[1] == b[0] and p[9] == b[8] and p[11] == b[8]
    assert np.array_equal(pp.log_baseline_path(a0, np.zeros((3, 0)), 2, 4), np.repeat(a0[:, None], 4, 1))


def test_prediction_weekday_matches_calendar():
    wd, days = pp.prediction_weekday(["2021-01-01"], 30, None)
    want = [(datetime.date(2021, 1, 1) + datetime.timedelta(days=i)).weekday() < 5 for i in range(30)]
    assert wd.tolist() == [float(x) for x in want] and str(days[0]) == "2021-01-01"
    wd2, days2 = pp.prediction_weekday(["0", "1"], 5, np.arange(5))
    assert days2 is None and wd2.tolist() == [0, 1, 2, 3, 4]


def _sim_inputs(case, n, S, seed, init_step=0):
    k = case["k"]
    rng = np.random.default_rng(seed)
    theta = so.constrain(synth.jitter_params(case["u"], n, scale=0.05, seed=seed, T=k.T))
    par = theta[:, :5].copy()
    a_path = pp.log_baseline_path(theta[:, 5], theta[:, 6:6 + k.T - 1], init_step, S)
    spatial = theta[:, 6 + k.T - 1:]
    W = pp.clipped(k.W, init_step, S)
    wd = pp.clipped(k.weekday_c, init_step, S)
    state = so.compute_state(k.initial_state, case["events"])
    init = np.stack([state[:, min(init_step, k.T - 1), :]] * n)
    init[:, :, 2] += rng.integers(0, 30, size=init.shape[:2])          # make sure there is an epidemic to run
    return par, a_path, spatial, W, wd, init


def test_oracle_simulation_conserves_population():
    case = H.build_case("micro_3x5", 3)
    par, a_path, spatial, W, wd, init = _sim_inputs(case, 2, 12, 5)
    ev = sim.simulate(case["k"], par, a_path, spatial, W, wd, init, seed=9)
    assert ev.shape == (2, 3, 12, 3) and np.all(ev >= 0) and np.all(ev == np.rint(ev))
    for d in range(2):
        st = so.compute_state(init[d], ev[d])
        assert np.all(st >= 0)
        assert np.array_equal(st.sum(-1), np.repeat(init[d].sum(-1)[:, None], 12, 1))
    # another draw id gives another stream, the same id the same events
    again = sim.simulate(case["k"], par, a_path, spatial, W, wd, init, seed=9)
    assert np.array_equal(ev, again)
    other = sim.simulate(case["k"], par, a_path, spatial, W, wd, init, seed=9, first_draw_id=5)
    assert not np.array_equal(ev, other)


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_binomial_matches_oracle_draw_for_draw():
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    case = H.build_case("micro_3x5", 1)
    rng = np.random.default_rng(4)
    n = np.concatenate([np.repeat([a for a, _ in BINOMIAL_GRID], 40), rng.integers(0, 400000, 600)]).astype(np.int32)
    p = np.concatenate([np.repeat([b for _, b in BINOMIAL_GRID], 40), 10.0 ** rng.uniform(-6, 0, 600)])
    p[-5:] = [0.0, 1.0, 0.5, 1.5, -0.1]
    with SeirModel(case["cov"], case["init"], max_chains=1) as model:
        got = model.selftest_binomial(n, p, seed=77)
    want = np.array([_oracle_binomial(int(n[i]), float(p[i]), 77, i) for i in range(n.size)])
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,S,init_step", [("micro_3x5", 3, 9, 0), ("ni11", 3, 20, 5), ("micro_17x70", 2, 30, 60)])
def test_hip_simulation_matches_oracle(name, n, S, init_step):
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    case = H.build_case(name, 5, alpha_t_sd=0.01)
    par, a_path, spatial, W, wd, init = _sim_inputs(case, n, S, 21, init_step)
    want = sim.simulate(case["k"], par, a_path, spatial, W, wd, init, seed=1234, first_draw_id=3)
    with SeirModel(case["cov"], case["init"], max_chains=1) as model:
        got = model.simulate(par, a_path, spatial, W, wd, init, seed=1234, first_draw_id=3)
    assert got.shape == want.shape
    assert np.array_equal(got, want), (np.argwhere(got != want)[:5], float(np.abs(got - want).max()))
    assert want.sum() > 0


@pytest.mark.gpu
def test_hip_simulation_uk380_properties():
    """Full-size run: conservation, feasibility, batching independence, expected incidence."""
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    case = H.build_case("uk380", 20210101)
    k = case["k"]
    n, S, init_step = 96, 56, 200
    par, a_path, spatial, W, wd, init = _sim_inputs(case, n, S, 8, init_step)
    with SeirModel(case["cov"], case["init"], max_chains=1) as model:
        ev = model.simulate(par, a_path, spatial, W, wd, init, seed=5)
        tail = model.simulate(par[40:], a_path[40:], spatial[40:], W, wd, init[40:], seed=5, first_draw_id=40)
    assert np.array_equal(ev[40:], tail)                      # a draw's stream depends on its id only
    assert np.all(ev >= 0) and np.all(ev == np.rint(ev))
    st = so.compute_state(init, ev)
    assert np.all(st >= 0)
    assert np.array_equal(st.sum(-1), np.repeat(init.sum(-1)[:, :, None], S, 2))
    # first simulated day: E[y] = n p with the oracle's rates, averaged over the identical-state draws
    d = 0
    th = np.concatenate([par[d], spatial[d]])
    lam, nu, rir = sim.day_rates(init[d], th, a_path[d, 0], W[0], wd[0], k)
    exp_se = init[d][:, 0] * -np.expm1(-lam)
    exp_ir = init[d][:, 2] * -np.expm1(-rir)
    same = [j for j in range(n) if np.array_equal(init[j], init[d])]
    assert abs(ev[d, :, 0, 2].sum() - exp_ir.sum()) < 6 * np.sqrt(exp_ir.sum()) + 5
    assert abs(ev[d, :, 0, 0].sum() - exp_se.sum()) < 6 * np.sqrt(exp_se.sum()) + 5
    assert len(same) >= 1


@pytest.mark.gpu
def test_predict_end_to_end(tmp_path):
    import pickle

    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd import hdf5io
    from covid19uk_amd.inference import inference as inf
    if not hdf5io.available():
        pytest.skip("libhdf5 not found")
    case = H.build_case("ni11", 3, alpha_t_sd=0.01)
    k, cov = case["k"], case["cov"]
    n = 4
    theta = so.constrain(synth.jitter_params(case["u"], n, scale=0.05, seed=2, T=k.T))
    samples = dict(psi=theta[:, 0], sigma_space=theta[:, 1], beta_area=theta[:, 2], gamma0=theta[:, 3],
                   gamma1=theta[:, 4], alpha_0=theta[:, 5], alpha_t=theta[:, 6:6 + k.T - 1],
                   spatial_effect=theta[:, 6 + k.T - 1:], seir=np.stack([case["events"]] * n),
                   initial_state=case["init"])
    data = str(tmp_path / "data.nc")
    dates = [str(np.datetime64("2021-01-01") + np.timedelta64(i, "D")) for i in range(k.T)]
    inf.write_inference_data(data, cov, case["events"][:, :, 2], dates)
    pk = str(tmp_path / "thin.pkl")
    with open(pk, "wb") as f:
        pickle.dump(samples, f)
    out = str(tmp_path / "pred.hd5")
    init, ev = pp.predict(data, pk, out, -7, 21, out_of_sample=True, seed=3)
    assert ev.shape == (n, k.M, 21, 3) and init.shape == (n, k.M, 4)
    state = so.compute_state(case["init"], case["events"])
    assert np.array_equal(init[0], state[:, k.T - 7, :])
    with hdf5io.File(out, "r") as f:
        assert np.array_equal(f.read("/predictions/events"), ev)
        # the time coordinate, CF-encoded as xarray writes it (days since the first predicted day)
        assert f.read_str_attr("/predictions/time", "units") == f"days since {dates[k.T - 7]} 00:00:00"
        assert np.array_equal(f.read("/predictions/time"), np.arange(21))
        assert f.shape("/predictions/initial_state") == (n, k.M, 4)
    # in-sample prediction reproduces through the oracle
    samples2 = {kk: v for kk, v in samples.items() if kk != "initial_state"}
    init2, ev2 = pp.predicted_incidence(samples2, case["init"], cov, 3, 10, out_of_sample=False, seed=8)
    a_path = pp.log_baseline_path(theta[:, 5], theta[:, 6:6 + k.T - 1], 3, 10)
    want = sim.simulate(k, theta[:, :5], a_path, theta[:, 6 + k.T - 1:], pp.clipped(k.W, 3, 10),
                        pp.clipped(k.weekday_c, 3, 10), init2, seed=8)
    assert np.array_equal(ev2, want)


@pytest.mark.gpu
def test_simulate_argument_errors_and_degenerate_inputs():
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd import _lib
    from covid19uk_amd.seir import SeirModel
    case = H.build_case("micro_3x5", 2)
    par, a_path, spatial, W, wd, init = _sim_inputs(case, 2, 4, 1)
    with SeirModel(case["cov"], case["init"], max_chains=1) as model:
        with pytest.raises(ValueError):
            model.simulate(par[:, :4], a_path, spatial, W, wd, init)
        with pytest.raises(ValueError):
            model.simulate(par, a_path, spatial, W[:-1], wd, init)
        with pytest.raises(_lib.SeirError):
            model.simulate(par[:0], a_path[:0], spatial[:0], W, wd, init[:0])     # no draws
        # nobody infected, nobody exposed: nothing can happen except the 1e-9 rate floor on S
        quiet = init.copy()
        quiet[:, :, 1:3] = 0
        ev = model.simulate(par, a_path, spatial, W, wd, quiet, seed=3)
        assert ev[..., 1:].sum() == 0 and ev[..., 0].sum() <= 1
        # a negative force of infection (psi * Cstar diagonal dominating) is clamped to probability 0
        neg = par.copy()
        neg[:, 0] = 1e9
        ev = model.simulate(neg, a_path, spatial, W, wd, init, seed=3)
        assert np.all(ev >= 0) and np.all(so.compute_state(init, ev) >= 0)
        want = sim.simulate(case["k"], neg, a_path, spatial, W, wd, init, seed=3)
        assert np.array_equal(ev, want)
