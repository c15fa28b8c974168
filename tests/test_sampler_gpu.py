"""GPU parity of the device-resident sampler (through the C-ABI) against the
CPU oracle sampler on the same Philox stream: proposals, accept decisions and
integer traces must agree exactly, continuous quantities to the stated fp64
tolerances (log-prob rtol 1e-9; parameters rtol 1e-6 after error growth along
16 leapfrogs x several sweeps)."""
import numpy as np
import pytest

from covid19uk_amd import synth
from oracle import mcmc_oracle as mo
from oracle import seir_oracle as so
from tests import helpers as H

pytestmark = pytest.mark.gpu

CFG_SMALL = dict(dmax=8, nmax=6, m=2, occult_nmax=5, num_event_time_updates=3)
CFG_REF = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)   # example_config.yaml:26-30


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    from covid19uk_amd.sampler import ChainSampler
    return SeirModel, ChainSampler


def _start(case, B, seed, scale=0.05):
    u = synth.jitter_params(case["u"], B, scale=scale, seed=seed, T=case["k"].T)
    ev = np.stack([case["events"]] * B)
    return u, ev


def _compare(trace, oracles, n, B, cfg, theta_rtol=1e-6, lp_rtol=1e-9):
    keys = ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I")
    for i in range(n):
        for b in range(B):
            o = oracles[b][i]
            assert bool(trace.hmc["is_accepted"][i, b]) == o["hmc"]["is_accepted"], (i, b, "hmc accept")
            # dual averaging feeds exp(log_accept_ratio) (a difference of ~1e3-sized log-probs) back into eps
            assert abs(trace.hmc["step_size"][i, b] - o["hmc"]["step_size"]) <= 1e-5 * o["hmc"]["step_size"]
            lp = o["hmc"]["target_log_prob"]
            assert abs(trace.hmc["target_log_prob"][i, b] - lp) <= lp_rtol * abs(lp), (i, b)
            for key in keys:
                got, want = trace.moves[key], o[key]
                assert np.array_equal(got["proposed_delta"][i, b], want["proposed_delta"]), (i, b, key)
                assert bool(got["is_accepted"][i, b]) == want["is_accepted"], (i, b, key)
                lp = want["target_log_prob"]
                assert abs(got["target_log_prob"][i, b] - lp) <= lp_rtol * abs(lp), (i, b, key)
            assert np.array_equal(trace.events[i, b], o["events"].astype(np.int32)), (i, b, "events")
            scale = np.maximum(np.abs(o["theta"]), 1e-3)
            assert np.max(np.abs(trace.theta[i, b] - o["theta"]) / scale) < theta_rtol, (i, b, "theta")


@pytest.mark.parametrize("moves", ["paired", "split", "paired-nopre", "paired+single"])
@pytest.mark.parametrize("name,cfg,seed,eps,n", [
    ("micro_5x24", CFG_SMALL, 1, 0.002, 12),
    ("ni11", CFG_REF, 2, 0.002, 8),
    ("micro_17x70", CFG_SMALL, 3, 0.0004, 6),
    # degenerate and pad-boundary shapes: one LAD, two days, exactly one 64-tile, one past it
    ("micro_1x6", CFG_SMALL, 4, 0.002, 6),
    ("micro_2x2", CFG_SMALL, 5, 0.002, 6),
    ("micro_64x64", CFG_SMALL, 6, 0.0001, 3),
    ("micro_65x65", CFG_SMALL, 7, 0.0001, 3),
])
def test_fixed_kernel_sweeps_match_oracle(api, name, cfg, seed, eps, n, moves):
    """The forms of the event-update launches -- paired with the S->E-type proposal pre-drawn one launch
    ahead (k_move_pair, default), paired without the pre-draw, and one proposal kernel per update
    (k_move_pa2, the cross-check) -- against the oracle."""
    # "+single": every leapfrog step by the single-workgroup kernel instead of the 64-lane chunks
    form = dict(moves=moves.split("+")[0], hmc="single" if moves.endswith("+single") else "chunk")
    SeirModel, ChainSampler = api
    case = H.build_case(name, seed, alpha_t_sd=0.005)
    B = 2
    u, ev = _start(case, B, seed)
    oracles = []
    for b in range(B):
        ch = mo.OracleChain(case["k"], cfg, u[b], ev[b], seed=77, chain_id=5 + b)
        ch.eps = eps
        oracles.append([ch.sweep_once() for _ in range(n)])
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, cfg, B, seed=77, first_chain_id=5, trace_capacity=n, **form) as s:
            s.set_state(u, ev)
            lp0 = s.log_prob()
            for b in range(B):
                want = so.joint_log_prob(u[b], ev[b], case["k"], "stable")
                assert abs(lp0[b] - want) <= 1e-9 * abs(want)
            s.set_kernel(step_size=eps)
            tr = s.sample(n)
            _compare(tr, oracles, n, B, cfg)
            # some of everything must actually have happened
            assert tr.hmc["is_accepted"].any()
            assert any(tr.moves[k]["is_accepted"].any() for k in tr.moves)


@pytest.mark.parametrize("name,cfg,eps", [
    ("micro_5x24", dict(dmax=3, nmax=1, m=1, occult_nmax=1, num_event_time_updates=1), 0.002),
    ("micro_5x24", dict(dmax=20, nmax=10, m=4, occult_nmax=8, num_event_time_updates=2), 0.002),
    ("micro_5x24", dict(dmax=8, nmax=6, m=3, occult_nmax=0, num_event_time_updates=2), 0.002),
    ("micro_6x10", dict(dmax=30, nmax=4, m=2, occult_nmax=3, num_event_time_updates=3), 0.002),   # T < 21: whole series is the occult range; dmax > T
    ("micro_3x70", dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=4), 0.001),  # more updates per row than rows
])
def test_move_configurations_match_oracle(api, name, cfg, eps):
    """Corners of the Mcmc configuration (one or four metapopulations per update, no occults, a series
    shorter than the occult window, dmax beyond the series) against the oracle, default launch forms."""
    SeirModel, ChainSampler = api
    case = H.build_case(name, 17, alpha_t_sd=0.005)
    B, n = 2, 8
    u, ev = _start(case, B, 17)
    oracles = []
    for b in range(B):
        ch = mo.OracleChain(case["k"], cfg, u[b], ev[b], seed=9, chain_id=b)
        ch.eps = eps
        oracles.append([ch.sweep_once() for _ in range(n)])
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, cfg, B, seed=9, trace_capacity=n) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=eps)
            tr = s.sample(n)
    _compare(tr, oracles, n, B, cfg)


_ORACLE_CACHE = {}


def _head(tr, n):
    """The first n sweeps of a trace."""
    from types import SimpleNamespace
    return SimpleNamespace(theta=tr.theta[:n], events=tr.events[:n], hmc={k: v[:n] for k, v in tr.hmc.items()},
                           moves={mk: {k: v[:n] for k, v in mv.items()} for mk, mv in tr.moves.items()})


def _same_chain_up_to_rounding(ref, got, rtol=1e-9):
    """Two launch forms whose sums are associated differently: the same events, accept decisions and integer draws; the
    continuous quantities equal up to rounding (1e-16-level differences in a log accept ratio move the adapted step size,
    and with it the next trajectory, by as much -- nothing amplifies them over a few dozen sweeps beyond 1e-9)."""
    assert np.array_equal(ref.events, got.events)
    assert np.array_equal(ref.hmc["is_accepted"], got.hmc["is_accepted"])
    np.testing.assert_allclose(got.theta, ref.theta, rtol=rtol, atol=1e-14)
    np.testing.assert_allclose(got.hmc["target_log_prob"], ref.hmc["target_log_prob"], rtol=1e-11, atol=0.0)
    np.testing.assert_allclose(got.hmc["step_size"], ref.hmc["step_size"], rtol=rtol, atol=0.0)
    for mk in ref.moves:
        assert np.array_equal(ref.moves[mk]["is_accepted"], got.moves[mk]["is_accepted"]), mk
        assert np.array_equal(ref.moves[mk]["proposed_delta"], got.moves[mk]["proposed_delta"]), mk


@pytest.mark.parametrize("moves", ["paired", "split", "paired-nopre", "paired+single"])
@pytest.mark.parametrize("name,cfg,seed,eps,n", [
    # T > 384: the 12-chunk instance of the proposal kernels (k_move_pair<12> / k_move_pa2<12>), few rows and many
    ("slow_3x400", CFG_REF, 11, 3e-5, 5),
    ("slow_70x400", CFG_SMALL, 12, 3e-5, 3),
    # T > 512: two day passes of the single-workgroup HMC kernels (k_hmc_step<*, 2, *>), 9 day chunks (rolled chunk loops)
    ("slow_5x520", CFG_REF, 13, 3e-5, 5),
    # M > 512: two row passes (k_hmc_step<*, *, 2>), the M-chunks no longer sum the row partials themselves
    ("slow_520x70", CFG_SMALL, 14, 3e-5, 3),
    # both, and 12 day chunks as at SYN-2048 x 730
    ("slow_530x730", CFG_REF, 15, 3e-6, 2),
    # T > 768: the 16-chunk instance of the proposal kernels
    ("slower_4x800", CFG_REF, 16, 3e-5, 5),
])
def test_long_series_and_wide_kernel_forms_match_oracle(api, name, cfg, seed, eps, n, moves):
    """Everything BASELINE's largest configuration (SYN-2048 x 730) runs and the smaller cases do not: the
    instances of the event-update kernels for series longer than 384 and 768 days (a row of the proposing wave is 12
    or 16 registers per lane instead of 6) and the multi-pass instances of the single-workgroup HMC kernels
    (T > 512, M > 512), draw by draw against the oracle in every launch form.  The 'slow' epidemics are still
    running on the last day, so every day chunk holds events."""
    form = dict(moves=moves.split("+")[0], hmc="single" if moves.endswith("+single") else "chunk")
    SeirModel, ChainSampler = api
    case = H.build_case(name, seed, alpha_t_sd=0.005)
    B = 2
    u, ev = _start(case, B, seed, scale=0.01)
    T = case["k"].T
    assert case["events"][:, T - 16:, :2].sum() > 0 and (T < 384 or case["events"][:, 384:, :2].sum() > 0)
    if name not in _ORACLE_CACHE:                       # the same oracle trace serves the four launch forms
        oracles = []
        for b in range(B):
            ch = mo.OracleChain(case["k"], cfg, u[b], ev[b], seed=77, chain_id=5 + b)
            ch.eps = eps
            oracles.append([ch.sweep_once() for _ in range(n)])
        _ORACLE_CACHE[name] = oracles
    oracles = _ORACLE_CACHE[name]
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, cfg, B, seed=77, first_chain_id=5, trace_capacity=n, **form) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=eps)
            tr = s.sample(n)
            _compare(tr, oracles, n, B, cfg)
            assert tr.hmc["is_accepted"].any()
            assert any(tr.moves[k]["is_accepted"].any() for k in tr.moves)
    if T > 384:
        # the proposals did reach the last day chunks
        days = np.concatenate([tr.moves[k]["proposed_delta"][..., 1, :].ravel() for k in tr.moves])
        assert (days >= 384).any() and (T < 768 or (days >= 768).any())


def test_sampler_matches_oracle_draw_by_draw_at_uk380(api):
    """The headline size against the oracle sampler itself (not only launch-form equivalence): three sweeps of two
    chains, the oracle with the C restatement of the density (full re-evaluation for every proposal, as the
    reference does: mcmc_kernel_factory.py:72-83).  Covers what the micro cases cannot: six-word ballot masks over
    380 rows, six-register rows over 365 days, 24 band workgroups."""
    from oracle import c_binding
    SeirModel, ChainSampler = api
    case = H.build_case("uk380", 6)
    k = case["k"]
    B, n, eps = 2, 3, 1.5e-5
    u, ev = _start(case, B, 6, scale=0.002)
    oracles = []
    for b in range(B):
        ch = mo.OracleChain(k, CFG_REF, u[b], ev[b], seed=21, chain_id=3 + b,
                            log_prob_fn=lambda u_, e_: c_binding.evaluate(k, u_, e_, 1),
                            log_prob_grad_fn=lambda u_, e_: c_binding.evaluate(k, u_, e_, 1, want_grad=True))
        ch.eps = eps
        oracles.append([ch.sweep_once() for _ in range(n)])
    for moves in ("paired", "split"):
        with SeirModel(case["cov"], case["init"], max_chains=B) as model:
            with ChainSampler(model, CFG_REF, B, seed=21, first_chain_id=3, trace_capacity=n, moves=moves) as s:
                s.set_state(u, ev)
                s.set_kernel(step_size=eps)
                tr = s.sample(n)
        _compare(tr, oracles, n, B, CFG_REF)
        rows = np.concatenate([tr.moves[key]["proposed_delta"][..., 0, :].ravel() for key in tr.moves])
        assert (rows >= 64).any() and (rows >= 320).any()
    assert tr.hmc["is_accepted"].any()
    assert any(tr.moves[key]["is_accepted"].any() for key in tr.moves)
    # BASELINE.json's third configuration exactly: ONE chain of UK-380 on the GPU (the layout of eight with seven chains
    # absent: a single XCD carries the sweep), in the default launch forms -- and in both tile shapes of the persistent
    # leapfrog launch -- against the first oracle chain
    for rows_ in (0, 32):
        with SeirModel(case["cov"], case["init"], max_chains=1) as model:
            with ChainSampler(model, CFG_REF, 1, seed=21, first_chain_id=3, trace_capacity=n, leap_rows=rows_) as s:
                s.set_state(u[:1], ev[:1])
                s.set_kernel(step_size=eps)
                tr1 = s.sample(n)
                assert s.launch_form() == ("chunk", "paired") and not s.recoveries
        _compare(tr1, oracles[:1], n, 1, CFG_REF)


def test_adaptation_windows_match_oracle(api):
    """Dual averaging, then dual averaging + diagonal mass adaptation (the fast and slow
    windows of inference.py:60-196), compared step by step.  Each window restarts from the
    same state so that leapfrog's amplification of 1e-9-level differences (the exploring step
    sizes sit at the edge of stability) does not compound across windows."""
    SeirModel, ChainSampler = api
    case = H.build_case("micro_5x24", 4, alpha_t_sd=0.005)
    cfg, B, n = CFG_SMALL, 2, 8
    u, ev = _start(case, B, 4)
    P = case["k"].P
    rv = (np.full(B, 5.0), np.tile(u.mean(0), (B, 1)), np.full((B, P), 0.5))
    eps0 = 0.0005

    def oracle_window(b, first_sweep, mass):
        ch = mo.OracleChain(case["k"], cfg, u[b], ev[b], seed=5, chain_id=b)
        ch.sweep = first_sweep
        ch.eps = eps0
        if mass:
            ch.set_adaptation(adapt_step=True, adapt_mass=True, num_adaptation_steps=n,
                              running_variance=(rv[0][b], rv[1][b], rv[2][b]))
        else:
            ch.set_adaptation(adapt_step=True, num_adaptation_steps=n)
        return [ch.sweep_once() for _ in range(n)], ch

    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, cfg, B, seed=5, trace_capacity=n) as s:
            for w, mass in enumerate((False, True)):
                s.set_state(u, ev)
                s.set_kernel(step_size=eps0)
                s.set_adaptation(adapt_step_size=True, adapt_mass=mass, num_adaptation_steps=n,
                                 running_variance=rv if mass else None)
                tr = s.sample(n)
                eps, var = s.get_kernel()
                outs = [oracle_window(b, w * n, mass) for b in range(B)]
                _compare(tr, [o[0] for o in outs], n, B, cfg, theta_rtol=1e-3, lp_rtol=1e-6)
                for b in range(B):
                    ch = outs[b][1]
                    assert abs(eps[b] - ch.eps) <= 1e-5 * ch.eps
                    assert np.max(np.abs(var[b] - ch.var) / ch.var) < 1e-4
                    if mass:
                        assert not np.allclose(var[b], 1.0)


def test_running_log_prob_matches_full_reevaluation_uk380(api):
    """Incremental bookkeeping at the BASELINE size: after sweeps with the reference's
    move configuration the running log-prob equals a from-scratch evaluation of the
    exported state (C oracle), the cached F equals a fresh contraction (refresh is a
    no-op within rounding), and the state stays a valid epidemic."""
    SeirModel, ChainSampler = api
    case = H.build_case("uk380", 6)
    B, n = 4, 6
    u, ev = _start(case, B, 6, scale=0.002)
    u[:, 6:6 + case["k"].T - 1] = 0.0
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, CFG_REF, B, seed=9, trace_capacity=n) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=2e-5)
            tr = s.sample(n)
            u1, ev1, lp_run = s.get_state()
            s.refresh()
            lp_fresh = s.log_prob()
    accepted = sum(int(tr.moves[k]["is_accepted"].sum()) for k in tr.moves)
    assert accepted > 0 and tr.hmc["is_accepted"].any()
    assert not np.array_equal(ev1, ev), "no event move was applied"
    for b in range(B):
        want = H.c_oracle_eval(case["k"], u1[b], ev1[b], stable=1)
        assert abs(lp_run[b] - want) <= 1e-9 * abs(want), (b, lp_run[b], want)
        assert abs(lp_fresh[b] - want) <= 1e-9 * abs(want)
        st = so.compute_state(case["init"], ev1[b], closed=True)
        assert st.min() >= 0
        assert np.array_equal(ev1[b][..., 2], ev[b][..., 2])
        assert np.array_equal(tr.events[-1, b], ev1[b].astype(np.int32))
        assert abs(tr.moves["occult/E->I"]["target_log_prob"][-1, b] - lp_run[b]) <= 1e-12 * abs(want)


def test_running_log_prob_matches_full_reevaluation_syn2048(api):
    """BASELINE's largest configuration (2048 regions x 730 days, T and M beyond one 512-thread
    pass of the single-workgroup kernels): two sweeps, then the same bookkeeping checks."""
    SeirModel, ChainSampler = api
    case = H.build_case("syn2048", 3)
    B, n = 2, 2
    u, ev = _start(case, B, 3, scale=0.001)
    u[:, 6:6 + case["k"].T - 1] = 0.0
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, CFG_REF, B, seed=4, trace_capacity=n) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=2e-6)
            tr = s.sample(n)
            u1, ev1, lp_run = s.get_state()
            s.refresh()
            lp_fresh = s.log_prob()
    assert np.isfinite(lp_run).all()
    assert sum(int(tr.moves[k]["is_accepted"].sum()) for k in tr.moves) > 0
    for b in range(B):
        want = H.c_oracle_eval(case["k"], u1[b], ev1[b], stable=1)
        assert abs(lp_run[b] - want) <= 1e-9 * abs(want), (b, lp_run[b], want)
        assert abs(lp_fresh[b] - want) <= 1e-9 * abs(want)
        assert so.compute_state(case["init"], ev1[b], closed=True).min() >= 0
        assert np.array_equal(tr.events[-1, b], ev1[b].astype(np.int32))


@pytest.mark.parametrize("hook", ["1", "2", "4", "8", "10"])
def test_pair_kernel_handshake_paths(api, hook):
    """k_move_pair's role 0 waits for the speculative roles' tokens before its first store (hooks 1 / 4:
    the token of role 1 / role 2 comes late) and discards their output when they never show up (hooks
    2 / 8; 10 = both): same traces.  The time-outs are counted (seir_sampler_pair_timeouts); an
    undisturbed run has none."""
    SeirModel, ChainSampler = api
    case = H.build_case("micro_5x24", 1, alpha_t_sd=0.005)
    B, n = 2, 6
    u, ev = _start(case, B, 1)
    oracles = []
    for b in range(B):
        ch = mo.OracleChain(case["k"], CFG_SMALL, u[b], ev[b], seed=77, chain_id=5 + b)
        ch.eps = 0.002
        oracles.append([ch.sweep_once() for _ in range(n)])
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, CFG_SMALL, B, seed=77, first_chain_id=5, trace_capacity=n, debug_pair=int(hook)) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=0.002)
            tr = s.sample(n)
            late = s.pair_timeouts()
        with ChainSampler(model, CFG_SMALL, B, seed=77, first_chain_id=5, trace_capacity=n) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=0.002)
            s.sample(n)
            assert not s.pair_timeouts().any()
    _compare(tr, oracles, n, B, CFG_SMALL)
    assert (late > 0).all() if int(hook) & 10 else not late.any()


@pytest.mark.parametrize("name,B,n,eps", [("uk380", 8, 40, 1.5e-5), ("micro_17x70", 5, 40, 0.0004)])
def test_repeated_runs_are_bitwise_identical(api, name, B, n, eps):
    """Nothing orders the workgroups of a launch, so any read of something another workgroup of the same
    launch writes shows up as run-to-run variation: eight repeats of the same sweeps must agree bit
    for bit (continuous quantities included)."""
    SeirModel, ChainSampler = api
    case = H.build_case(name, 31, alpha_t_sd=0.005)
    u, ev = _start(case, B, 31, scale=0.002 if name == "uk380" else 0.05)
    cfg = CFG_REF if name == "uk380" else CFG_SMALL
    ref = None
    for rep in range(8):
        with SeirModel(case["cov"], case["init"], max_chains=B) as model:
            with ChainSampler(model, cfg, B, seed=8, trace_capacity=n) as s:
                s.set_state(u, ev)
                s.set_kernel(step_size=eps)
                tr = s.sample(n)
                _, _, lp = s.get_state()
        got = (tr.theta.copy(), tr.events.copy(), tr.hmc["target_log_prob"].copy(), lp.copy(),
               np.stack([tr.moves[k]["target_log_prob"] for k in sorted(tr.moves)]))
        if ref is None:
            ref = got
        else:
            for a, b_ in zip(ref, got):
                assert np.array_equal(a, b_), rep


@pytest.mark.parametrize("forms", ["chunk,paired", "single,split", "chunk,paired-nopre"])
@pytest.mark.parametrize("name,B,n,eps", [("uk380", 8, 25, 1.5e-5), ("micro_17x70", 5, 25, 0.0004), ("ni11", 2, 25, 0.002)])
def test_results_do_not_depend_on_workgroup_timing(api, name, B, n, eps, forms):
    """The debug-skew option delays a pseudo-random third of the workgroups of every launch by ~30 us -- longer
    than any kernel of the sweep runs -- so a workgroup that reads what another one of the same launch
    writes gets the other version.  All three thirds against the undisturbed run, bit for bit."""
    SeirModel, ChainSampler = api
    form = dict(hmc=forms.split(",")[0], moves=forms.split(",")[1])
    case = H.build_case(name, 33, alpha_t_sd=0.005)
    u, ev = _start(case, B, 33, scale=0.002 if name == "uk380" else 0.05)
    cfg = CFG_REF if name != "micro_17x70" else CFG_SMALL
    ref = None
    for skew in (0, 1, 2, 3):
        with SeirModel(case["cov"], case["init"], max_chains=B) as model:
            model.set_option(debug_skew=skew)
            with ChainSampler(model, cfg, B, seed=8, trace_capacity=n, **form) as s:
                s.set_state(u, ev)
                s.set_kernel(step_size=eps)
                tr = s.sample(n)
                _, _, lp = s.get_state()
        got = (tr.theta.copy(), tr.events.copy(), tr.hmc["target_log_prob"].copy(), lp.copy(),
               np.stack([tr.moves[k]["target_log_prob"] for k in sorted(tr.moves)]))
        if ref is None:
            ref = got
        else:
            for a, b_ in zip(ref, got):
                assert np.array_equal(a, b_), skew


@pytest.mark.parametrize("name,cfg,eps", [("micro_5x24", CFG_SMALL, 0.002), ("micro_2x2", CFG_SMALL, 0.002),
                                          ("ni11", CFG_REF, 0.002)])
def test_paired_and_split_agree_over_many_sweeps_with_frequent_conflicts(api, name, cfg, eps):
    """Few rows: the speculative E->I proposal collides with the updates accepted before it in a large
    share of the launches, so the re-draw path and every certification rule run hundreds of times."""
    SeirModel, ChainSampler = api
    case = H.build_case(name, 41, alpha_t_sd=0.005)
    B, n = 4, 300
    u, ev = _start(case, B, 41)
    out = {}
    for mode in ("paired", "split", "paired-nopre"):
        with SeirModel(case["cov"], case["init"], max_chains=B) as model:
            with ChainSampler(model, cfg, B, seed=5, trace_capacity=n, moves=mode) as s:
                s.set_state(u, ev)
                s.set_kernel(step_size=eps)
                out[mode] = s.sample(n)
                assert not s.pair_timeouts().any()
    a = out["paired"]
    for other in ("split", "paired-nopre"):
        b_ = out[other]
        assert np.array_equal(a.events, b_.events), other
        for key in a.moves:
            assert np.array_equal(a.moves[key]["proposed_delta"], b_.moves[key]["proposed_delta"]), (other, key)
            assert np.array_equal(a.moves[key]["is_accepted"], b_.moves[key]["is_accepted"]), (other, key)
        assert np.array_equal(a.hmc["is_accepted"], b_.hmc["is_accepted"]), other
    # with and without the pre-draw the same arithmetic runs: every continuous quantity bit for bit
    c_ = out["paired-nopre"]
    assert np.array_equal(a.theta, c_.theta)
    for key in a.moves:
        assert np.array_equal(a.moves[key]["target_log_prob"], c_.moves[key]["target_log_prob"]), key
    assert sum(int(a.moves[k]["is_accepted"].sum()) for k in a.moves) > 200


def test_paired_and_split_launch_forms_agree_at_uk380(api):
    """The paired form (k_move_pair: S->E updates inside the proposing workgroup, speculative E->I
    proposal certified by row comparison, deferred F band) against one-kernel-per-update on the
    BASELINE size, where row conflicts are rare events: the same proposals, decisions and events."""
    SeirModel, ChainSampler = api
    case = H.build_case("uk380", 12)
    B, n = 4, 40
    u, ev = _start(case, B, 12, scale=0.002)
    out = {}
    for mode in ("paired", "split", "paired-nopre"):
        with SeirModel(case["cov"], case["init"], max_chains=B) as model:
            with ChainSampler(model, CFG_REF, B, seed=21, trace_capacity=n, moves=mode) as s:
                s.set_state(u, ev)
                s.set_kernel(step_size=1.5e-5)
                out[mode] = s.sample(n)
    a, c_ = out["paired"], out["paired-nopre"]
    assert np.array_equal(a.events, c_.events) and np.array_equal(a.theta, c_.theta)
    for key in a.moves:
        assert np.array_equal(a.moves[key]["target_log_prob"], c_.moves[key]["target_log_prob"]), key
    a, b_ = out["paired"], out["split"]
    assert np.array_equal(a.events, b_.events)
    for key in a.moves:
        assert np.array_equal(a.moves[key]["proposed_delta"], b_.moves[key]["proposed_delta"]), key
        assert np.array_equal(a.moves[key]["is_accepted"], b_.moves[key]["is_accepted"]), key
        assert np.allclose(a.moves[key]["target_log_prob"], b_.moves[key]["target_log_prob"], rtol=1e-11, atol=0)
    assert np.array_equal(a.hmc["is_accepted"], b_.hmc["is_accepted"])
    assert np.allclose(a.theta, b_.theta, rtol=1e-9, atol=1e-12)
    assert sum(int(a.moves[k]["is_accepted"].sum()) for k in a.moves) > 50


def test_chains_are_independent_of_batch_composition(api):
    """Chain c's draws depend only on (seed, global chain id): running chains {0,1,2,3}
    together or chain 2 alone (first_chain_id=2) gives bit-identical traces -- the
    property the multi-GPU sharding relies on."""
    SeirModel, ChainSampler = api
    case = H.build_case("ni11", 7)
    u, ev = _start(case, 4, 7)
    n = 5
    with SeirModel(case["cov"], case["init"], max_chains=4) as model:
        with ChainSampler(model, CFG_REF, 4, seed=3, trace_capacity=n) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=0.002)
            tr_all = s.sample(n)
    with SeirModel(case["cov"], case["init"], max_chains=1) as model:
        with ChainSampler(model, CFG_REF, 1, seed=3, first_chain_id=2, trace_capacity=n) as s:
            s.set_state(u[2:3], ev[2:3])
            s.set_kernel(step_size=0.002)
            tr_one = s.sample(n)
    assert np.array_equal(tr_all.theta[:, 2], tr_one.theta[:, 0])
    assert np.array_equal(tr_all.events[:, 2], tr_one.events[:, 0])
    assert np.array_equal(tr_all.hmc["target_log_prob"][:, 2], tr_one.hmc["target_log_prob"][:, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 16])
def test_handoffs_of_the_persistent_launches_do_not_depend_on_timing(api, B):
    """k_leap hands partial sums and tables from workgroup to workgroup as self-validating 16-byte words (value + step number in
    both halves, no acknowledgement, no flag), k_move_pairs its proposal descriptor; at 16 chains the leapfrog launch runs as two
    launches of 8 that share the step numbers.  Three hundred sweeps at the headline size from one seed, the second run with the
    workgroups' start skewed: a torn or stale word, a missed wait or a race between steps shows as a difference in the bits --
    and no wait may time out.  (tools/dev/ll_soak.py is the long form: 5 000 sweeps, three skews.)"""
    SeirModel, ChainSampler = api
    case = H.build_case("uk380", 43, alpha_t_sd=0.005)
    u = synth.jitter_params(case["u"], B, scale=0.002, seed=3, T=case["k"].T)
    ev = np.stack([case["events"]] * B)
    out = []
    for skew in (0, 3):
        with SeirModel(case["cov"], case["init"], max_chains=B) as model:
            model.set_option(debug_skew=skew)
            with ChainSampler(model, CFG_REF, B, seed=21, trace_capacity=100, record_events=False) as s:
                if not s.xcd_local():
                    pytest.skip("this GPU does not place block ids congruent mod 8 on one XCD: the persistent forms are not used")
                s.set_state(u, ev)
                s.set_kernel(step_size=1.2e-5)
                for _ in range(3):
                    s.reset_trace()
                    s.run(100)
                tr = s.read_trace(100, events=False)
                assert not s.pair_timeouts().any()
                out.append((s.get_state(), tr))
    (q0, e0, l0), t0 = out[0]
    (q1, e1, l1), t1 = out[1]
    assert np.array_equal(q0, q1) and np.array_equal(e0, e1) and np.array_equal(l0, l1)
    assert np.array_equal(t0.theta, t1.theta)
    for k in t0.hmc:
        assert np.array_equal(t0.hmc[k], t1.hmc[k]), k
    assert t0.hmc["is_accepted"].any() and sum(int(v["is_accepted"].sum()) for v in t0.moves.values()) > 0


@pytest.mark.parametrize("B,groups,affinity,graph", [(8, 1, 3, 0), (8, 1, 0, 0), (3, 1, 3, 0), (16, 1, 3, 0),
                                                    (8, 2, 3, 0), (6, 4, 3, 0), (8, 1, 3, 1), (6, 2, 3, 1)])
def test_launch_geometries_give_identical_chains(api, B, groups, affinity, graph):
    """Block-to-chain mappings (XCD affinity for 1/2/4/8 chains per launch, natural grids
    otherwise, chain groups on separate streams) only move work around: every chain's trace must
    be bit-identical to the same chain run alone, under stream launches and under graph replay."""
    SeirModel, ChainSampler = api
    case = H.build_case("micro_17x70", 9, alpha_t_sd=0.005)
    u, ev = _start(case, B, 9)
    n = 4
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        model.set_option(xcd_affinity=affinity)
        # stream launches (default) or hipGraph replay; chain groups on separate streams
        with ChainSampler(model, CFG_SMALL, B, seed=5, trace_capacity=n, chain_groups=groups, use_graph=bool(graph)) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=0.0004)
            tr_all = s.sample(n)
    for b in (0, B - 1):
        with SeirModel(case["cov"], case["init"], max_chains=1) as model:
            with ChainSampler(model, CFG_SMALL, 1, seed=5, first_chain_id=b, trace_capacity=n) as s:
                s.set_state(u[b:b + 1], ev[b:b + 1])
                s.set_kernel(step_size=0.0004)
                tr_one = s.sample(n)
        assert np.array_equal(tr_all.theta[:, b], tr_one.theta[:, 0]), b
        assert np.array_equal(tr_all.events[:, b], tr_one.events[:, 0]), b
        for key in tr_all.moves:
            assert np.array_equal(tr_all.moves[key]["proposed_delta"][:, b], tr_one.moves[key]["proposed_delta"][:, 0])
    assert tr_all.hmc["is_accepted"].any()


def test_sampler_argument_errors(api):
    from covid19uk_amd import _lib
    SeirModel, ChainSampler = api
    case = H.build_case("micro_5x24", 8)
    with SeirModel(case["cov"], case["init"], max_chains=1) as model:
        with pytest.raises(_lib.SeirError):
            ChainSampler(model, dict(CFG_SMALL, m=9), 1)
        with pytest.raises(_lib.SeirError):
            ChainSampler(model, CFG_SMALL, 2)                 # more chains than the context allows
        with ChainSampler(model, CFG_SMALL, 1) as s:
            with pytest.raises(_lib.SeirError):
                s.run(1)                                      # no state yet
            bad = case["events"][None].copy()
            bad[0, 0, 0, 0] = 0.5
            with pytest.raises(_lib.SeirError):
                s.set_state(case["u"][None], bad)


def test_overlapped_bursts_deliver_the_same_draws(api):
    """sample_bursts (burst buffer in two halves, copies on a second stream into page-locked memory, consumer
    on a worker thread) against plain blocking reads of the same sweeps: identical draws, in order."""
    SeirModel, ChainSampler = api
    case = H.build_case("micro_17x70", 9, alpha_t_sd=0.005)
    B, burst, nb = 3, 7, 5
    u, ev = _start(case, B, 9)
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, CFG_SMALL, B, seed=5, trace_capacity=burst * nb) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=0.0004)
            ref = s.sample(burst * nb)
        got = []
        with ChainSampler(model, CFG_SMALL, B, seed=5, trace_capacity=2 * burst) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=0.0004)
            s.sample_bursts(nb, burst, lambda tr, i: got.append((i, tr.theta.copy(), tr.events.copy(),
                                                                  tr.hmc["target_log_prob"].copy(),
                                                                  tr.moves["move/E->I"]["proposed_delta"].copy())))
            with pytest.raises(ValueError):
                s.sample_bursts(1, burst + 1, lambda tr, i: None)
    assert [g[0] for g in got] == list(range(nb))
    assert np.array_equal(np.concatenate([g[1] for g in got]), ref.theta)
    assert np.array_equal(np.concatenate([g[2] for g in got]), ref.events)
    assert np.array_equal(np.concatenate([g[3] for g in got]), ref.hmc["target_log_prob"])
    assert np.array_equal(np.concatenate([g[4] for g in got]), ref.moves["move/E->I"]["proposed_delta"])


def test_compact_event_trace_is_lossless_and_guards_its_range(api):
    """record_events="u16": the same draws with the events recorded as uint16 (half the bytes over PCIe), and a
    chain whose counts do not fit 16 bits makes the read fail instead of delivering truncated counts."""
    from covid19uk_amd import _lib
    SeirModel, ChainSampler = api
    case = H.build_case("micro_17x70", 9, alpha_t_sd=0.005)
    B, n = 2, 5
    u, ev = _start(case, B, 9)
    out = {}
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        for mode in (True, "u16"):
            with ChainSampler(model, CFG_SMALL, B, seed=5, trace_capacity=n, record_events=mode) as s:
                s.set_state(u, ev)
                s.set_kernel(step_size=0.0004)
                out[mode] = s.sample(n)
    assert out["u16"].events.dtype == np.uint16 and out[True].events.dtype == np.int32
    assert np.array_equal(out["u16"].events, out[True].events)
    assert np.array_equal(out["u16"].theta, out[True].theta)
    # counts beyond 65535: populations of millions with a fast epidemic
    cov = case["cov"]
    import dataclasses
    big = dataclasses.replace(cov, N=cov.N * 400.0)
    init = case["init"].copy()
    init[:, 0] = big.N - init[:, 1:].sum(1)
    init[:, 2] += 3.0e5
    init[:, 0] -= 3.0e5
    evb = np.zeros_like(case["events"])
    evb[:, 0, 2] = 7.0e4                              # 70 000 removals on day 0
    with SeirModel(big, init, max_chains=1) as model:
        with ChainSampler(model, CFG_SMALL, 1, seed=5, trace_capacity=2, record_events="u16") as s:
            s.set_state(case["u"][None], evb[None])
            s.set_kernel(step_size=1e-6)
            with pytest.raises(_lib.SeirError, match="65535"):
                s.sample(1)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 8, 12, 16])
def test_chunk_roles_inside_the_gradient_launch_give_the_same_bits(api, B):
    """hmc="chunk-launch" runs the chunk roles of a leapfrog step inside that step's gradient launch (k_se_chunk) and
    hmc="chunk-split" launches tiles and roles separately: the same code in the same order, every traced quantity equal
    to the last bit, with and without workgroup skew (any number of chains, in the layout of the next multiple of 8: chain
    b on XCD b mod 8; used only when the GPU places block ids congruent mod 8 on one XCD each).  hmc="chunk" runs the
    whole trajectory in ONE persistent launch (k_leap: the gradient tiles keep their cells in
    registers; tiles and chunk roles hand each other partial sums and tables through the XCD's L2; the first step draws
    the momentum as k_hmc_step<0> does in the other forms).  There the start point's energy and log-probability are
    summed in another order, so it is held to the same draws and decisions, the continuous quantities equal up to
    rounding -- and to its own bits under workgroup skew."""
    case = H.build_case("ni11", 31)
    u = synth.jitter_params(case["u"], B, scale=0.01, seed=3, T=case["k"].T)
    ev = np.stack([case["events"]] * B)
    cfg = dict(dmax=10, nmax=5, m=2, occult_nmax=5, num_event_time_updates=2)
    out = {}
    for mode, skew in (("chunk-split", 0), ("chunk", 0), ("chunk", 2), ("chunk-launch", 0), ("chunk-launch", 1)):
        with api[0](case["cov"], case["init"], max_chains=B) as model:
            model.set_option(debug_skew=skew)
            with api[1](model, cfg, B, seed=77, trace_capacity=40, hmc=mode) as s:
                if mode != "chunk-split" and not s.xcd_local():
                    pytest.skip("this GPU does not place block ids congruent mod 8 on one XCD: the fused form is not used")
                s.set_state(u, ev)
                s.set_kernel(step_size=0.02)
                s.set_adaptation(adapt_step_size=True, num_adaptation_steps=20)
                s.run(40)
                tr = s.read_trace(40)
                assert not s.pair_timeouts().any()
                out[(mode, skew)] = tr
    ref = out[("chunk-split", 0)]
    for key in (("chunk-launch", 0), ("chunk-launch", 1)):      # the same code in the same order: the same bits
        got = out[key]
        assert np.array_equal(ref.theta, got.theta), key
        assert np.array_equal(ref.events, got.events), key
        for k in ref.hmc:
            assert np.array_equal(ref.hmc[k], got.hmc[k]), (key, k)
    # (the step size explores the edge of stability here -- dual averaging from 0.02 -- and a leapfrog trajectory there
    # amplifies rounding differences by orders of magnitude per sweep: the forms are compared over the first sweeps)
    _same_chain_up_to_rounding(_head(ref, 4), _head(out[("chunk", 0)], 4), rtol=1e-6)
    base = out[("chunk", 0)]                                    # the persistent launch against itself under workgroup skew: bits
    got = out[("chunk", 2)]
    assert np.array_equal(base.theta, got.theta) and np.array_equal(base.events, got.events)
    for k in base.hmc:
        assert np.array_equal(base.hmc[k], got.hmc[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,eps", [("uk380", 8, 1.2e-5), ("uk380", 3, 1.2e-5), ("uk380", 16, 1.2e-5), ("slow_520x60", 2, 3e-5),
                                        ("micro_17x70", 8, 0.0004), ("micro_20x60", 8, 0.0004)])
def test_leapfrog_launch_forms_agree_at_size(api, name, B, eps):
    """The same comparison where the persistent launch has something to get wrong: the headline size (72 tile workgroups
    and 12 four-wave roles per chain, six day chunks, eight counter shards), a partial layout of 8, 16 chains (the persistent
    launch as two launches of 8 chains one after the other: they share the step numbers), and M > 512 (the tiles
    also form the row scalars: the second instance of the kernels).  Three sweeps with all updates on; "chunk-leap" is the
    persistent launch for the inner steps alone, "chunk-stage" the one that also carries the trajectory's first step and both
    end-point gradients, "chunk" the whole trajectory (last half kick, accept test, adaptation and trace by the roles too)."""
    case = H.build_case(name, 43, alpha_t_sd=0.005)
    u = synth.jitter_params(case["u"], B, scale=0.002 if name == "uk380" else 0.01, seed=3, T=case["k"].T)
    ev = np.stack([case["events"]] * B)
    cfg = CFG_REF if name == "uk380" else CFG_SMALL
    out = {}
    rows_of = {}
    for mode, skew in (("chunk-split", 0), ("chunk-launch", 0), ("chunk-leap", 0), ("chunk-stage", 0), ("chunk", 0), ("chunk", 3),
                       ("chunk/32", 0), ("chunk-stage/32", 0), ("chunk-launch-fold", 0), ("chunk-launch-fold", 2)):
        # "/32": the 32-row tile shape of the persistent launch where the default is the 24-row one (UK-380)
        mode, _, rows_ = mode.partition("/")
        rows_of[(mode, skew)] = rows_
        if rows_:
            if name != "uk380":
                continue
            skew = 32                                        # (key only)
        with api[0](case["cov"], case["init"], max_chains=B) as model:
            model.set_option(debug_skew=skew if skew != 32 else 0)
            with api[1](model, cfg, B, seed=13, trace_capacity=3, hmc=mode, leap_rows=int(rows_ or 0)) as s:
                if mode != "chunk-split" and not s.xcd_local():
                    pytest.skip("this GPU does not place block ids congruent mod 8 on one XCD: the fused forms are not used")
                s.set_state(u, ev)
                s.set_kernel(step_size=eps)
                out[(mode, skew)] = (s.sample(3), s.get_state())
                assert not s.pair_timeouts().any()
    ref, ref_state = out[("chunk-split", 0)]
    assert ref.hmc["is_accepted"].any()
    got, got_state = out[("chunk-launch", 0)]                   # the same code in the same order: the same bits
    assert np.array_equal(ref.theta, got.theta) and np.array_equal(ref.events, got.events)
    for k in ref.hmc:
        assert np.array_equal(ref.hmc[k], got.hmc[k]), k
    assert np.array_equal(ref_state[0], got_state[0]) and np.array_equal(ref_state[2], got_state[2])
    for mode in ("chunk-leap", "chunk-stage", "chunk"):         # the persistent launches: other orders of summation
        got, got_state = out[(mode, 0)]
        _same_chain_up_to_rounding(ref, got)
        np.testing.assert_allclose(got_state[2], ref_state[2], rtol=1e-11, atol=0.0)
    # the per-step form that carries the trajectory's ends itself (what 16+ chains and SYN-2048 run by default): against the
    # stage form up to rounding, against itself under workgroup skew to the bit
    got, got_state = out[("chunk-launch-fold", 0)]
    _same_chain_up_to_rounding(ref, got)
    np.testing.assert_allclose(got_state[2], ref_state[2], rtol=1e-11, atol=0.0)
    sk, sk_state = out[("chunk-launch-fold", 2)]
    assert np.array_equal(sk.theta, got.theta) and np.array_equal(sk.events, got.events) and np.array_equal(sk_state[2], got_state[2])
    for k in got.hmc:
        assert np.array_equal(sk.hmc[k], got.hmc[k]), k
    for key in (("chunk", 32), ("chunk-stage", 32)):             # ... in the other tile shape too
        if key in out:
            got, got_state = out[key]
            _same_chain_up_to_rounding(ref, got)
            np.testing.assert_allclose(got_state[2], ref_state[2], rtol=1e-11, atol=0.0)
    base, base_state = out[("chunk", 0)]                        # ... and their own bits under workgroup skew
    got, got_state = out[("chunk", 3)]
    assert np.array_equal(base.theta, got.theta) and np.array_equal(base.events, got.events)
    for k in base.hmc:
        assert np.array_equal(base.hmc[k], got.hmc[k]), k
    assert np.array_equal(base_state[2], got_state[2])


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,eps,adapt", [("micro_20x60", 8, 0.0004, False), ("micro_20x60", 5, 0.0004, True), ("uk380", 8, 2e-5, False)])
def test_trajectory_end_inside_the_launch_accepts_rejects_and_adapts_like_the_stage_kernel(api, name, B, eps, adapt):
    """hmc="chunk" closes the trajectory inside the persistent launch -- every chunk role makes the accept test from the
    roles' parts and applies it to its own entries, restoring the start point and rebuilding its share of the tables on
    rejection (hmc_final_apply) -- where hmc="chunk-stage" launches k_hmc_step<2> for it (sizes at which the persistent
    launch is used: one or six day chunks, an even number of row tiles).  With step sizes at which part of the proposals
    is rejected, and with dual averaging and the running variance on: the same decisions, a rejected sweep leaves the
    parameters exactly where they were, every traced quantity and the final state equal up to rounding, the adapted step
    sizes and variances too."""
    case = H.build_case(name, 19, alpha_t_sd=0.005)
    u = synth.jitter_params(case["u"], B, scale=0.002 if name == "uk380" else 0.01, seed=4, T=case["k"].T)
    ev = np.stack([case["events"]] * B)
    cfg = CFG_REF if name == "uk380" else CFG_SMALL
    P = case["k"].P
    n = 10
    seen = []
    for scale in (1.0, 2.5):
        out = {}
        for mode in ("chunk-stage", "chunk", "chunk-launch-fold"):
            with api[0](case["cov"], case["init"], max_chains=B) as model:
                with api[1](model, cfg, B, seed=29, trace_capacity=n, hmc=mode) as s:
                    if not s.xcd_local():
                        pytest.skip("this GPU does not place block ids congruent mod 8 on one XCD: the fused forms are not used")
                    s.set_state(u, ev)
                    s.set_kernel(step_size=eps * scale)
                    if adapt:
                        s.set_adaptation(adapt_step_size=True, adapt_mass=True, num_adaptation_steps=n,
                                         running_variance=(np.full(B, 5.0), np.tile(u.mean(0), (B, 1)), np.full((B, P), 0.5)))
                    out[mode] = (s.sample(n), s.get_state(), s.get_kernel())
                    assert not s.pair_timeouts().any()
        ref, ref_state, ref_k = out["chunk-stage"]
        acc = ref.hmc["is_accepted"].astype(bool)
        seen.append(acc)
        for form in ("chunk", "chunk-launch-fold"):          # the end inside the persistent launch / by k_hmc_final after the per-step launches
            got, got_state, got_k = out[form]
            assert np.array_equal(acc, got.hmc["is_accepted"].astype(bool)), form
            _same_chain_up_to_rounding(_head(ref, 5), _head(got, 5), rtol=1e-7)
            # a rejected HMC update leaves the parameters of the previous draw (the event updates do not touch them)
            for t in range(1, n):
                for b in range(B):
                    if not got.hmc["is_accepted"][t, b]:
                        assert np.array_equal(got.theta[t, b], got.theta[t - 1, b]), (form, t, b)
            np.testing.assert_allclose(got_k[0], ref_k[0], rtol=1e-5)
            np.testing.assert_allclose(got_k[1], ref_k[1], rtol=1e-5)
    allacc = np.concatenate([a.ravel() for a in seen])
    assert allacc.any() and not allacc.all(), allacc.mean()     # both branches were taken


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,cfg,eps", [("ni11", 8, dict(dmax=10, nmax=5, m=2, occult_nmax=5, num_event_time_updates=3), 0.02),
                                            ("ni11", 1, dict(dmax=10, nmax=5, m=2, occult_nmax=5, num_event_time_updates=3), 0.02),
                                            ("ni11", 5, dict(dmax=10, nmax=5, m=2, occult_nmax=5, num_event_time_updates=3), 0.02),
                                            ("uk380", 2, dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5), 1.2e-5),
                                            ("ni11", 11, dict(dmax=10, nmax=5, m=2, occult_nmax=5, num_event_time_updates=3), 0.02),
                                            ("ni11", 16, dict(dmax=10, nmax=5, m=2, occult_nmax=5, num_event_time_updates=3), 0.02),
                                            ("uk380", 8, dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5), 1.2e-5),
                                            # sixteen chains, two per XCD: 12 band workgroups of 32 rows (four per wave) per chain
                                            ("uk380", 16, dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5), 1.2e-5)])
def test_band_workgroups_inside_the_pair_launch_give_the_same_bits(api, name, B, cfg, eps):
    """moves="paired" with a multiple of 8 chains whose workgroups all fit the chip at once evaluates the band part of the E->I-type log-ratio with more workgroups of the
    k_move_pair launch (pair_band_block: done-tokens and an XCD-local hand-off, the F band of the update accepted in
    the launch added on the fly and applied once every role is done) where the GPU places block ids congruent mod 8
    on one XCD each; moves="paired-delta" always launches k_move_delta for it.  The two sum the band's cells in a
    different order (~1e-16 on a log-ratio): every integer draw and accept flag and the final state must be identical,
    continuous quantities agree to 1e-12 -- with and without workgroup skew and with the speculative roles made late
    (among themselves the in-pair runs must agree to the last bit)."""
    case = H.build_case(name, 41)
    n = 30 if name == "ni11" else 12
    u = synth.jitter_params(case["u"], B, scale=0.01 if name == "ni11" else 0.002, seed=5, T=case["k"].T)
    ev = np.stack([case["events"]] * B)
    out = {}
    for mode, skew, dbg in (("paired-delta", 0, 0), ("paired", 0, 0), ("paired", 1, 0), ("paired", 0, 5), ("paired-launch", 0, 0)):
        with api[0](case["cov"], case["init"], max_chains=B) as model:
            model.set_option(debug_skew=skew)
            with api[1](model, cfg, B, seed=91, trace_capacity=n, moves=mode, debug_pair=dbg) as s:
                if mode != "paired-delta" and not s.xcd_local():
                    pytest.skip("this GPU does not place block ids congruent mod 8 on one XCD: the fused form is not used")
                s.set_state(u, ev)
                s.set_kernel(step_size=eps)
                s.run(n)
                tr = s.read_trace(n)
                if dbg == 0:
                    assert not s.pair_timeouts().any()
                out[(mode, skew, dbg)] = (tr, s.get_state())
    ref, ref_state = out[("paired-delta", 0, 0)]
    base, base_state = out[("paired", 0, 0)]
    for key in (("paired", 0, 0), ("paired", 1, 0), ("paired", 0, 5), ("paired-launch", 0, 0)):
        got, got_state = out[key]
        assert np.array_equal(ref.events, got.events), key
        np.testing.assert_allclose(got.theta, ref.theta, rtol=1e-12, atol=0.0, err_msg=str(key))
        for k in ref.hmc:
            np.testing.assert_allclose(got.hmc[k], ref.hmc[k], rtol=1e-12, atol=0.0, err_msg=str((key, k)))
        for mk in ref.moves:
            for k in ref.moves[mk]:
                if k == "target_log_prob":
                    np.testing.assert_allclose(got.moves[mk][k], ref.moves[mk][k], rtol=1e-12, atol=0.0, err_msg=str((key, mk)))
                else:
                    assert np.array_equal(ref.moves[mk][k], got.moves[mk][k]), (key, mk, k)
        assert np.array_equal(ref_state[1], got_state[1]), key
        np.testing.assert_allclose(got_state[2], ref_state[2], rtol=1e-12, atol=0.0)
        # the in-pair runs among themselves: same code, same order
        assert np.array_equal(base.theta, got.theta) and np.array_equal(base.events, got.events), key
        for mk in base.moves:
            for k in base.moves[mk]:
                assert np.array_equal(base.moves[mk][k], got.moves[mk][k]), (key, mk, k)
        assert np.array_equal(base_state[2], got_state[2]), key
