"""CPU-only checks of the sampler oracle: the Philox stream against the
published known-answer vectors, proposal reversibility against independently
written proposal densities, and basic invariants of a sweep."""
import math

import numpy as np
import pytest

from oracle import mcmc_oracle as mo
from oracle import seir_oracle as so
from covid19uk_amd import synth
from tests import helpers as H

CFG = dict(dmax=6, nmax=5, m=2, occult_nmax=4, num_event_time_updates=2)


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    r = mo.philox4x32_10(np.array([0], dtype=np.uint64), 0, 0, 0, 0, 0)
    assert [int(x[0]) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = mo.philox4x32_10(np.array([0xffffffff], dtype=np.uint64), 0xffffffff, 0xffffffff, 0xffffffff,
                         0xffffffff, 0xffffffff)
    assert [int(x[0]) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = mo.philox4x32_10(np.array([0x243f6a88], dtype=np.uint64), 0x85a308d3, 0x13198a2e, 0x03707344,
                         0xa4093822, 0x299f31d0)
    assert [int(x[0]) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniforms_are_open_interval_and_reproducible():
    a, b = mo.rng_uniform2(123, 4, 5, 6, np.arange(1000))
    assert a.min() > 0 and a.max() < 1 and b.min() > 0 and b.max() < 1
    a2, _ = mo.rng_uniform2(123, 4, 5, 6, np.arange(1000))
    assert np.array_equal(a, a2)
    assert abs(a.mean() - 0.5) < 0.05


# ---- independent proposal densities (written from DESIGN.md, not from mcmc_oracle) ----
def _closed(k, ev):
    return so.compute_state(k.initial_state, ev, closed=True)


def logq_time_move(k, cfg, ev, tgt, m, t, delta, x):
    """log q(t, x | m, delta) of moving x `tgt` events of row m from t to t+delta."""
    K = ev[..., tgt]
    st = _closed(k, ev)
    D = int((K[m] > 0).sum())
    if K[m, t] <= 0:
        return -math.inf
    t2 = t + delta
    lo, hi = min(t, t2), max(t, t2)
    dec = tgt + 1 if delta > 0 else tgt          # compartment that loses x on (lo, hi]
    bound = math.inf if dec == 0 else st[m, lo + 1:hi + 1, dec].min()
    xmax = int(max(0, min(cfg["nmax"], K[m, t], bound)))
    if x > xmax:
        return -math.inf
    return -math.log(D) - math.log(xmax + 1)


def logq_occult(k, cfg, ev, tgt, t_range, m, t, x, is_del):
    K = ev[..., tgt]
    st = _closed(k, ev)
    lo, hi = t_range
    T = k.T
    rows = K[:, lo:hi].sum(1) > 0
    Hd = int(rows.sum())
    if is_del:
        if Hd == 0 or not rows[m] or K[m, t] <= 0:
            return -math.inf
        Dm = int((K[m, lo:hi] > 0).sum())
        bound = st[m, t + 1:T + 1, tgt + 1].min()
        xmax = int(max(0, min(cfg["occult_nmax"], K[m, t], bound)))
        if x > xmax:
            return -math.inf
        return math.log(0.5) - math.log(Hd) - math.log(Dm) - math.log(xmax + 1)
    bound = math.inf if tgt == 0 else st[m, t + 1:T + 1, tgt].min()
    xmax = int(max(0, min(cfg["occult_nmax"], bound)))
    if x > xmax:
        return -math.inf
    return (math.log(0.5) if Hd > 0 else 0.0) - math.log(k.M) - math.log(hi - lo) - math.log(xmax + 1)


@pytest.fixture(scope="module")
def chain():
    c = H.build_case("micro_5x24", 3)
    ch = mo.OracleChain(c["k"], CFG, c["u"], c["events"], seed=11, chain_id=0, t_range=(14, 24))
    return c, ch


def test_event_time_move_log_q_ratio_is_reverse_minus_forward(chain):
    c, ch = chain
    k = c["k"]
    n_checked = 0
    for sweep in range(60):
        ch.sweep = sweep
        for tgt in (0, 1):
            before = ch.events.copy()
            out = ch.event_time_move(tgt, 0, tgt)
            if not out["valid"]:
                continue
            new = out["proposed_events"]
            want = 0.0
            pd = out["proposed_delta"]
            for j in range(pd.shape[1]):
                m, t, delta, x = (int(v) for v in pd[:, j])
                if delta == 0 or x == 0:       # a null sub-move is its own reverse: no correction
                    continue
                want += logq_time_move(k, CFG, new, tgt, m, t + delta, -delta, x) \
                    - logq_time_move(k, CFG, before, tgt, m, t, delta, x)
            assert abs(out["log_q_ratio"] - want) < 1e-12, (sweep, tgt, out["log_q_ratio"], want)
            n_checked += 1
    assert n_checked > 40


def test_null_sub_move_carries_no_correction():
    """Two rows per proposal: when one row draws x = 0 (nothing moves there) the acceptance of what the
    OTHER row does must not pick up a factor from it.  The first version paired the null sub-move with
    "x = 0 from the arrival day" -- a different bound, possibly an empty day -- and the kernel then did
    not leave the posterior invariant (caught by tests/test_invariance*.py on the enumerated toy)."""
    c = H.build_case("micro_5x24", 3)
    k = c["k"]
    hits = 0
    for chain_id in range(40):
        ch = mo.OracleChain(k, CFG, c["u"], c["events"], seed=99, chain_id=chain_id)
        for tgt in (0, 1):
            before = ch.events.copy()
            out = ch.event_time_move(tgt, 0, tgt)
            pd = out["proposed_delta"]
            xs = [int(v) for v in pd[3]]
            if not out["valid"] or 0 not in xs or max(xs) == 0:
                continue
            want = 0.0
            for j in range(pd.shape[1]):
                m, t, delta, x = (int(v) for v in pd[:, j])
                if x > 0:
                    want += logq_time_move(k, CFG, out["proposed_events"], tgt, m, t + delta, -delta, x) \
                        - logq_time_move(k, CFG, before, tgt, m, t, delta, x)
            assert abs(out["log_q_ratio"] - want) < 1e-12
            hits += 1
    assert hits >= 3


def test_occult_log_q_ratio_is_reverse_minus_forward(chain):
    c, ch = chain
    k = c["k"]
    seen = set()
    for sweep in range(100, 220):
        ch.sweep = sweep
        for tgt in (0, 1):
            before = ch.events.copy()
            out = ch.occult_move(tgt, 0, 2 + tgt)
            m, t, sign, x = (int(v) for v in out["proposed_delta"][:, 0])
            new = out["proposed_events"]
            is_del = sign < 0
            fwd = logq_occult(k, CFG, before, tgt, ch.t_range, m, t, x, is_del)
            rev = logq_occult(k, CFG, new, tgt, ch.t_range, m, t, x, not is_del)
            assert math.isfinite(fwd)
            if math.isinf(rev):
                assert out["log_q_ratio"] == -math.inf
            else:
                assert abs(out["log_q_ratio"] - (rev - fwd)) < 1e-12
            seen.add((tgt, is_del))
    assert seen == {(0, False), (0, True), (1, False), (1, True)}


def test_sweep_keeps_state_feasible_and_log_prob_consistent(chain):
    c, ch = chain
    k = c["k"]
    ch.eps = 0.003
    for _ in range(15):
        out = ch.sweep_once()
        st = so.compute_state(k.initial_state, ch.events, closed=True)
        assert st.min() >= 0
        assert ch.events.min() >= 0
        # I->R events are observed data: never touched (mcmc_kernel_factory.py:129-161)
        assert np.array_equal(ch.events[..., 2], c["events"][..., 2])
        assert abs(ch.logp - so.joint_log_prob(ch.u, ch.events, k, "stable")) <= 1e-9 * abs(ch.logp)
        assert set(out) >= {"hmc", "move/S->E", "move/E->I", "occult/S->E", "occult/E->I", "theta", "events"}


def test_dual_averaging_moves_step_size_toward_target():
    c = H.build_case("micro_3x10", 5)
    ch = mo.OracleChain(c["k"], dict(CFG, num_event_time_updates=0), c["u"], c["events"], seed=3)
    ch.eps = 0.5                                    # far too large: rejects
    ch.set_adaptation(adapt_step=True, num_adaptation_steps=40)
    acc = []
    for _ in range(40):
        acc.append(ch.sweep_once()["hmc"]["is_accepted"])
    assert ch.eps < 0.5
    assert np.mean(acc[20:]) > 0.3


# ---------------------------------------------------------------------------------------------
# oracle/mcmc_oracle.c (the sweep in plain C: what bench.py times on the host cores) against oracle/mcmc_oracle.py
# ---------------------------------------------------------------------------------------------
def _same_sweep(ra, rb, where):
    assert ra["hmc"]["is_accepted"] == rb["hmc"]["is_accepted"], where
    lp = ra["hmc"]["target_log_prob"]
    assert abs(rb["hmc"]["target_log_prob"] - lp) <= 1e-9 * abs(lp), where
    assert abs(rb["hmc"]["step_size"] - ra["hmc"]["step_size"]) <= 1e-7 * ra["hmc"]["step_size"], where
    for key in ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I"):
        assert np.array_equal(ra[key]["proposed_delta"], rb[key]["proposed_delta"]), (where, key)
        assert ra[key]["is_accepted"] == rb[key]["is_accepted"], (where, key)
        qa, qb = ra[key]["log_q_ratio"], rb[key]["log_q_ratio"]
        assert qa == qb or abs(qa - qb) < 1e-12, (where, key)
    assert np.array_equal(ra["events"], rb["events"]), where
    np.testing.assert_allclose(rb["theta"], ra["theta"], rtol=1e-7, atol=1e-12, err_msg=str(where))


@pytest.mark.parametrize("name,cfg,eps,n", [("micro_5x24", dict(dmax=8, nmax=6, m=2, occult_nmax=5, num_event_time_updates=3), 0.002, 10),
                                            ("ni11", dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5), 0.002, 4),
                                            ("micro_2x2", dict(dmax=3, nmax=1, m=1, occult_nmax=1, num_event_time_updates=1), 0.002, 8),
                                            ("micro_17x70", dict(dmax=8, nmax=6, m=4, occult_nmax=5, num_event_time_updates=2), 0.0004, 4)])
def test_c_sampler_follows_the_python_oracle_draw_by_draw(name, cfg, eps, n):
    """Same Philox stream, same proposals, same decisions; the Box-Muller normals differ in the last bit (NumPy's
    vectorised log / sin / cos are not glibc's), so continuous quantities agree to rounding."""
    from oracle import c_binding as cb
    case = H.build_case(name, 3, alpha_t_sd=0.005)
    u = synth.jitter_params(case["u"], 1, scale=0.05, seed=3, T=case["k"].T)[0]
    a = mo.OracleChain(case["k"], cfg, u, case["events"], seed=77, chain_id=5)
    b = cb.COracleChain(case["k"], cfg, u, case["events"], seed=77, chain_id=5)
    a.eps = b.eps = eps
    assert abs(a.logp - b.logp) <= 1e-10 * abs(a.logp)
    for i in range(n):
        _same_sweep(a.sweep_once(), b.sweep_once(), (name, i))
    assert a.n_evals == b.n_evals                     # a full evaluation per proposal, as the reference


def test_c_sampler_adaptation_windows_and_disabled_kernels():
    from oracle import c_binding as cb
    cfg = dict(dmax=8, nmax=6, m=2, occult_nmax=5, num_event_time_updates=2)
    case = H.build_case("micro_5x24", 9, alpha_t_sd=0.005)
    k = case["k"]
    u = synth.jitter_params(case["u"], 1, scale=0.05, seed=4, T=k.T)[0]
    a = mo.OracleChain(k, cfg, u, case["events"], seed=5, chain_id=2)
    b = cb.COracleChain(k, cfg, u, case["events"], seed=5, chain_id=2)
    a.eps = b.eps = 0.004
    for ch in (a, b):
        ch.set_adaptation(adapt_step=True, num_adaptation_steps=6)
    for i in range(8):                                # runs past the window: the averaged step size takes over
        _same_sweep(a.sweep_once(), b.sweep_once(), ("fast", i))
    rv = (5.0, np.full(k.P, 0.1), np.full(k.P, 0.5))
    for ch in (a, b):
        ch.set_adaptation(adapt_step=True, adapt_mass=True, num_adaptation_steps=5, running_variance=rv)
    for i in range(6):
        _same_sweep(a.sweep_once(), b.sweep_once(), ("slow", i))
    np.testing.assert_allclose(b.var, a.var, rtol=1e-7)
    # sub-kernels that draw but never accept
    dis = ("hmc", "occult/E->I")
    a2 = mo.OracleChain(k, cfg, u, case["events"], seed=5, chain_id=2, disable=dis)
    b2 = cb.COracleChain(k, cfg, u, case["events"], seed=5, chain_id=2, disable=dis)
    for i in range(5):
        ra, rb = a2.sweep_once(), b2.sweep_once()
        assert not rb["hmc"]["is_accepted"] and not rb["occult/E->I"]["is_accepted"]
        for key in ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I"):
            assert np.array_equal(ra[key]["proposed_delta"], rb[key]["proposed_delta"]) and ra[key]["is_accepted"] == rb[key]["is_accepted"]
        assert np.array_equal(ra["events"], rb["events"])
    b2.run(3)                                         # the timed form: nothing but C in the loop
    for _ in range(3):
        a2.sweep_once()
    assert np.array_equal(a2.events, b2.events)
