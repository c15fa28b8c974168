"""Surviving a placement failure of the persistent launches (include/seir_hip.h, "Surviving a placement failure"):
snapshot / restore of the chain state, the sticky hand-off error, and `ChainSampler`'s own recovery -- the burst loop of
covid19uk/inference/inference.py:453-468 must deliver the draws of an undisturbed run whatever else is on the GPU."""
import threading

import numpy as np
import pytest

from covid19uk_amd import _lib, synth
from tests import helpers as H
from tests.test_sampler_gpu import CFG_REF, CFG_SMALL, api  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _case(name, B, seed=43):
    case = H.build_case(name, seed, alpha_t_sd=0.005)
    u = synth.jitter_params(case["u"], B, scale=0.002 if name == "uk380" else 0.01, seed=3, T=case["k"].T)
    ev = np.stack([case["events"]] * B)
    return case, u, ev, (CFG_REF if name == "uk380" else CFG_SMALL), (1.2e-5 if name == "uk380" else 0.0004)


def _same_draws(ref, got, rtol=1e-7, atol=1e-14):
    """Two launch forms: the same events, accept decisions and integer draws; continuous quantities up to the order of
    summation (with the adaptations on, a 1e-16 difference in a log accept ratio moves step size and variances, and through
    them every later draw: entries near zero then differ by ~1e-9 absolutely)."""
    assert np.array_equal(ref.events, got.events)
    assert np.array_equal(ref.hmc["is_accepted"], got.hmc["is_accepted"])
    np.testing.assert_allclose(got.theta, ref.theta, rtol=rtol, atol=atol)
    np.testing.assert_allclose(got.hmc["target_log_prob"], ref.hmc["target_log_prob"], rtol=max(1e-10, 1e-3 * rtol), atol=0.0)
    np.testing.assert_allclose(got.hmc["step_size"], ref.hmc["step_size"], rtol=max(rtol, 1e-6), atol=0.0)
    for mk in ref.moves:
        assert np.array_equal(ref.moves[mk]["is_accepted"], got.moves[mk]["is_accepted"]), mk
        assert np.array_equal(ref.moves[mk]["proposed_delta"], got.moves[mk]["proposed_delta"]), mk


def _same_bits(a, b):
    assert np.array_equal(a.theta, b.theta) and np.array_equal(a.events, b.events)
    for k in a.hmc:
        assert np.array_equal(a.hmc[k], b.hmc[k]), k
    for mk in a.moves:
        for k in a.moves[mk]:
            assert np.array_equal(a.moves[mk][k], b.moves[mk][k]), (mk, k)


@pytest.mark.parametrize("name,B,adapt", [("uk380", 8, False), ("micro_20x60", 5, True), ("uk380", 1, True)])
def test_restore_reproduces_the_burst(api, name, B, adapt):
    """snapshot -> n sweeps -> restore -> the same n sweeps: bit for bit in the same launch form (every counter, token and
    table is back where it was), draw for draw in the per-step forms a recovery falls back on -- with dual averaging and
    the running variance on, so that the adaptation state is part of what comes back."""
    case, u, ev, cfg, eps = _case(name, B)
    n = 6
    P = case["k"].P
    with api[0](case["cov"], case["init"], max_chains=B) as model:
        with api[1](model, cfg, B, seed=13, trace_capacity=n, auto_recover=False) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=eps)
            if adapt:
                s.set_adaptation(adapt_step_size=True, adapt_mass=True, num_adaptation_steps=3 * n,
                                 running_variance=(np.full(B, 5.0), np.tile(u.mean(0), (B, 1)), np.full((B, P), 0.5)))
            s.sample(3)                                   # somewhere inside a run, not at the initial state
            s.snapshot(1)
            first, st_first, k_first = s.sample(n), s.get_state(), s.get_kernel()
            s.restore(1)
            again, st_again, k_again = s.sample(n), s.get_state(), s.get_kernel()
            _same_bits(first, again)
            for a, b in zip(st_first + k_first, st_again + k_again):
                assert np.array_equal(a, b)
            for form in (("chunk-launch", "paired-launch"), ("chunk-split", "paired-delta")):
                s.restore(1)
                s.set_launch_form(*form)
                assert s.launch_form() == form
                other, st_other = s.sample(n), s.get_state()
                _same_draws(first, other, rtol=1e-6 if adapt else 1e-7, atol=1e-8 if adapt else 1e-14)
                assert np.array_equal(st_first[1], st_other[1])
                np.testing.assert_allclose(st_other[2], st_first[2], rtol=1e-10, atol=0.0)
            assert not s.pair_timeouts().any()


def test_a_hand_off_time_out_is_sticky_until_the_state_is_rebuilt(api):
    """After a fatal time-out (injected: the chain's counter raised as a timed-out wait leaves it) the read of the trace
    fails, and so does everything after it -- another read, another run -- until restore / refresh / set_state; a caller
    that swallows the first error cannot go on with a state whose F is out of step with its events."""
    case, u, ev, cfg, eps = _case("micro_20x60", 8)
    with api[0](case["cov"], case["init"], max_chains=8) as model:
        with api[1](model, cfg, 8, seed=13, trace_capacity=4, auto_recover=False) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=eps)
            s.snapshot(0)
            good = s.sample(4)
            s.restore(0)
            _lib.check(s._lib.seir_sampler_debug_fail_handoff(s._s, 3))
            s.reset_trace()
            s.run(4)
            with pytest.raises(_lib.HandoffTimeout, match="chain 3"):
                s.read_trace(4)
            with pytest.raises(_lib.HandoffTimeout):
                s.read_trace(4)                            # still there
            with pytest.raises(_lib.HandoffTimeout):
                s.run(1)                                   # no sweeps from an unreliable state
            with pytest.raises(_lib.SeirError):
                s.snapshot(1)                              # and no snapshot of it
            s.restore(0)                                   # back to the last good state: the burst can be run again
            _same_bits(good, s.sample(4))
            _lib.check(s._lib.seir_sampler_debug_fail_handoff(s._s, 0))
            s.run(1)
            with pytest.raises(_lib.HandoffTimeout):
                model.sync(), s.read_trace(1)
            s.refresh()                                    # the other way out: everything recomputed from the planes
            s.sample(2)
            assert not s.pair_timeouts().any()


@pytest.mark.parametrize("name,B", [("uk380", 8), ("micro_20x60", 8)])
def test_sampler_recovers_from_a_time_out_by_itself(api, name, B):
    """ChainSampler.sample / sample_bursts with auto_recover (the default): a time-out in the middle of a run costs the
    burst it happened in -- restored from the snapshot taken at its start, run again one launch form down -- and nothing
    else: the run delivers the draws of an undisturbed one (integers exactly, continuous quantities up to the order of
    summation of the other form), logs the recovery, and returns to the preferred form after `retry_after` clean bursts."""
    case, u, ev, cfg, eps = _case(name, B)
    nb, burst = 6, 4

    def collect(into):
        def consume(tr, i):
            into[i] = (tr.theta.copy(), tr.events.copy(), {k: v.copy() for k, v in tr.hmc.items()},
                       {mk: {k: v.copy() for k, v in mv.items()} for mk, mv in tr.moves.items()})
        return consume

    runs = {}
    for disturb in (False, True):
        with api[0](case["cov"], case["init"], max_chains=B) as model:
            with api[1](model, cfg, B, seed=13, trace_capacity=2 * burst, log=None) as s:
                s.retry_after = 2
                s.set_state(u, ev)
                s.set_kernel(step_size=eps)
                got = {}
                if disturb:
                    inner = collect(got)

                    def consume(tr, i, inner=inner, s=s):
                        inner(tr, i)
                        if i == 1 and not s.recoveries:    # while burst 2 or 3 is in flight
                            _lib.check(s._lib.seir_sampler_debug_fail_handoff(s._s, B - 1))
                else:
                    consume = collect(got)
                s.sample_bursts(nb, burst, consume)
                tail = s.sample(burst)                    # and the blocking form after it
                if disturb:
                    _lib.check(s._lib.seir_sampler_debug_fail_handoff(s._s, 0))
                tail2 = s.sample(burst)
                runs[disturb] = (got, tail, tail2, list(s.recoveries), s.launch_form(), s.get_state())
    ref, got = runs[False], runs[True]
    # (the micro case's trajectories amplify a last-bit difference between two launch forms by ~1e8 over the 32 sweeps -- the
    # integer draws and the events, compared exactly, are what shows that it is the same chain)
    tol = dict(rtol=1e-7, atol=1e-14) if name == "uk380" else dict(rtol=1e-3, atol=1e-6)
    assert not ref[3] and len(got[3]) == 2, got[3]
    assert got[3][0]["failed_form"] == ("chunk", "paired") and got[3][0]["rerun_form"] == ("chunk-launch", "paired-launch")
    assert sorted(got[0]) == list(range(nb))
    from types import SimpleNamespace
    for i in range(nb):
        a, b = (SimpleNamespace(theta=x[0], events=x[1], hmc=x[2], moves=x[3]) for x in (ref[0][i], got[0][i]))
        _same_draws(a, b, **tol)
    _same_draws(ref[1], got[1], **tol)
    _same_draws(ref[2], got[2], **tol)
    assert np.array_equal(ref[5][1], got[5][1])
    np.testing.assert_allclose(got[5][2], ref[5][2], rtol=max(1e-10, 1e-3 * tol["rtol"]), atol=0.0)


def test_two_samplers_started_together_on_one_gpu_both_deliver_their_solo_draws(api):
    """Two UK-380 x 8-chain samplers in the default (persistent, whole-chip) launch forms on ONE GPU at the same time, from
    two host threads: each launch needs (nearly) every CU, so the two grids can each hold part of the chip and wait for
    the rest -- the bounded waits time out and the samplers fall back by themselves.  Whether or not that happens in a
    given run, both must finish, with the draws they produce when they have the GPU to themselves."""
    B, nb, burst = 8, 5, 6
    case, u, ev, cfg, eps = _case("uk380", 2 * B)

    def run(idx, out, barrier=None):
        with api[0](case["cov"], case["init"], max_chains=B) as model:
            with api[1](model, cfg, B, seed=29, first_chain_id=idx * B, trace_capacity=2 * burst, log=None) as s:
                s.set_state(u[idx * B:(idx + 1) * B], ev[idx * B:(idx + 1) * B])
                s.set_kernel(step_size=eps)
                if barrier is not None:
                    barrier.wait()
                bursts = {}
                s.sample_bursts(nb, burst, lambda tr, i: bursts.__setitem__(i, (tr.theta.copy(), tr.events.copy(),
                                                                                {k: v.copy() for k, v in tr.hmc.items()})))
                out[idx] = (bursts, s.get_state(), list(s.recoveries), int(s.pair_timeouts().sum()))

    solo, both = {}, {}
    for idx in (0, 1):
        run(idx, solo)
    barrier = threading.Barrier(2)
    errors = []

    def guarded(idx):
        try:
            run(idx, both, barrier)
        except Exception as e:                            # surfaced below: a thread's exception is otherwise lost
            errors.append((idx, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass
    threads = [threading.Thread(target=guarded, args=(idx,)) for idx in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    for idx in (0, 1):
        assert not solo[idx][2]
        for i in range(nb):
            a, b = solo[idx][0][i], both[idx][0][i]
            assert np.array_equal(a[1], b[1]), (idx, i)
            assert np.array_equal(a[2]["is_accepted"], b[2]["is_accepted"]), (idx, i)
            np.testing.assert_allclose(b[0], a[0], rtol=1e-7, atol=1e-14)
            np.testing.assert_allclose(b[2]["target_log_prob"], a[2]["target_log_prob"], rtol=1e-11, atol=0.0)
        assert np.array_equal(solo[idx][1][1], both[idx][1][1])
    print("recoveries while sharing the GPU:", [len(both[i][2]) for i in (0, 1)])
