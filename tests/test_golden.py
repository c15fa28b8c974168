"""Golden vectors (tests/golden/seir_golden.npz, made by tests/golden/make_golden.py).

CPU: the oracle (NumPy and C) reproduces them; GPU: the HIP path, through the
C-ABI, matches them (log-prob rtol 1e-9, gradient rtol 1e-6, sampler trace exact
in its integer parts)."""
import os

import numpy as np
import pytest

from covid19uk_amd import model_spec as ms
from oracle import mcmc_oracle as mo
from oracle import seir_oracle as so
from tests import helpers as H

G = np.load(os.path.join(H.GOLDEN, "seir_golden.npz"))
CASES = ["micro_1x1", "micro_2x3", "micro_3x5", "ni11"]
CFG = dict(dmax=8, nmax=6, m=2, occult_nmax=5, num_event_time_updates=2)


def _case(name):
    g = {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith(name + "/")}
    cov = ms.Covariates(C=g["C"], W=g["W"], N=g["N"], adjacency=g["adjacency"], weekday=g["weekday"], area=g["area"])
    return g, cov


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    g, cov = _case(name)
    k = H.oracle_constants(cov, g["initial_state"])
    lp, grad = so.joint_log_prob_and_grad(g["u"], g["events"], k)
    assert abs(lp - float(g["logp"])) <= 1e-13 * abs(lp)
    assert np.allclose(grad, g["grad"], rtol=1e-12, atol=1e-12 * np.abs(g["grad"]).max())
    assert abs(so.joint_log_prob(g["u"], g["events"], k, "reference") - float(g["logp_reference_form"])) <= 1e-13 * abs(lp)
    if "logp_mpmath" in g:
        assert abs(lp - float(g["logp_mpmath"])) <= 1e-12 * abs(lp)
    c_lp, c_grad = H.c_oracle_eval(k, g["u"], g["events"], stable=1, want_grad=True)
    assert abs(c_lp - float(g["logp"])) <= 1e-10 * abs(lp)      # glibc lgamma vs scipy gammaln: ~1 ulp of lgamma(N)
    assert np.allclose(c_grad, g["grad"], rtol=1e-9, atol=1e-9 * np.abs(g["grad"]).max())


def test_oracle_sampler_reproduces_golden_trace():
    c = H.build_case("micro_3x5", 103, alpha_t_sd=0.005)
    ch = mo.OracleChain(c["k"], CFG, c["u"], c["events"], seed=2021, chain_id=3, t_range=(2, 5))
    ch.eps = 0.002
    for i in range(6):
        s = ch.sweep_once()
        assert np.array_equal(s["events"], G["trace/events"][i])
        assert bool(s["hmc"]["is_accepted"]) == bool(G["trace/hmc_accept"][i])
        assert np.allclose(s["theta"], G["trace/theta"][i], rtol=1e-9, atol=1e-12)
        for key in ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I"):
            assert np.array_equal(s[key]["proposed_delta"], G[f"trace/{key}/delta"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_log_prob_matches_golden(name):
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    g, cov = _case(name)
    with SeirModel(cov, g["initial_state"]) as model:
        lp, grad = model.log_prob_grad(g["u"], g["events"])
    want = float(g["logp"])
    assert abs(lp - want) <= 1e-9 * abs(want)
    scale = np.maximum(np.abs(g["grad"]), 1e-6 * np.abs(g["grad"]).max())
    assert np.max(np.abs(grad - g["grad"]) / scale) < 1e-6
    if "logp_mpmath" in g:
        assert abs(lp - float(g["logp_mpmath"])) <= 1e-9 * abs(want)


@pytest.mark.gpu
def test_hip_sampler_matches_golden_trace():
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.sampler import ChainSampler
    from covid19uk_amd.seir import SeirModel
    c = H.build_case("micro_3x5", 103, alpha_t_sd=0.005)
    with SeirModel(c["cov"], c["init"]) as model:
        with ChainSampler(model, CFG, 1, seed=2021, first_chain_id=3, t_range=(2, 5), trace_capacity=6) as s:
            s.set_state(c["u"][None], c["events"][None])
            s.set_kernel(step_size=0.002)
            tr = s.sample(6)
    assert np.array_equal(tr.events[:, 0], G["trace/events"].astype(np.int32))
    assert np.array_equal(tr.hmc["is_accepted"][:, 0], G["trace/hmc_accept"])
    assert np.allclose(tr.hmc["target_log_prob"][:, 0], G["trace/hmc_logp"], rtol=1e-9)
    assert np.allclose(tr.theta[:, 0], G["trace/theta"], rtol=1e-6, atol=1e-9)
    for key in ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I"):
        assert np.array_equal(tr.moves[key]["proposed_delta"][:, 0], G[f"trace/{key}/delta"])
        assert np.array_equal(tr.moves[key]["is_accepted"][:, 0], G[f"trace/{key}/accept"])
        assert np.allclose(tr.moves[key]["target_log_prob"][:, 0], G[f"trace/{key}/logp"], rtol=1e-9)
