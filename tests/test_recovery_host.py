"""Host logic of `ChainSampler`'s recovery (covid19uk_amd/sampler.py) without a GPU: the burst loop of
covid19uk/inference/inference.py:453-468 must deliver every burst exactly once and in order whatever hand-off time-outs
occur -- a time-out surfaces at the read of some LATER burst, everything enqueued since the last good snapshot is run again,
one launch form down the ladder, and the preferred form comes back after enough clean bursts.  The device is replaced by a
deterministic stand-in for the C-ABI calls the loop makes (the draws of burst k are a function of the state the burst
started from, as on the device); tests/test_recovery_gpu.py runs the same logic on the real one."""
import numpy as np
import pytest

from covid19uk_amd import _lib
from covid19uk_amd.sampler import FALLBACK_FORMS, ChainSampler


class FakeDevice(ChainSampler):
    """The C-ABI calls of sample / sample_bursts replaced by a model of them: `state` is an integer advanced by every sweep,
    a burst's draws are the states it went through; snapshot / restore copy it; a poisoned sampler fails like the library."""

    def __init__(self, fail_at=(), fail_forms=(("chunk", "paired"),), cap=8):
        self.B, self.P, self.M, self.T, self.mmax = 1, 1, 1, 1, 1
        self.cap = cap
        self.record_events = False
        self.events_dtype = np.int32
        self.auto_recover = True
        self.preferred_form = ("chunk", "paired")
        self.recoveries, self.retry_after = [], 2
        self._fallback_level, self._clean_bursts, self._log = 0, 0, None
        self._pinned, self._pinned_key = [], None
        self.state, self.snap = 0, {}
        self.form = self.preferred_form
        self.poisoned = False
        self.fail_at = set(fail_at)          # sweeps (state values) at which a launch in one of `fail_forms` times out
        self.fail_forms = set(fail_forms)
        self.trace = {}                      # slot -> state recorded there
        self.slot = 0
        self.inflight = None
        self.log = []

    def _fail(self):
        raise _lib.HandoffTimeout("libseirhip call failed (-3): chain 0: 1 in-launch hand-off(s) timed out", -3)

    def snapshot(self, slot=0):
        if self.poisoned:
            raise _lib.SeirError("unreliable", -3)
        self.snap[slot] = self.state

    def restore(self, slot=0):
        self.state = self.snap[slot]
        self.poisoned = False
        self.inflight = None
        self.log.append(("restore", slot))

    def set_launch_form(self, hmc, moves):
        self.form = (hmc, moves)

    def launch_form(self):
        return self.form

    def reset_trace(self, at=0):
        self.slot = at

    def run(self, n):
        if self.poisoned:
            self._fail()
        for _ in range(n):
            self.state += 1
            if self.state in self.fail_at and self.form in self.fail_forms:
                self.fail_at.discard(self.state)                 # a placement accident, not a property of the sweep
                self.poisoned_pending = True
            # (a burst that timed out still "runs": its draws are garbage)
            self.trace[self.slot] = -1 if getattr(self, "poisoned_pending", False) else self.state
            self.slot += 1

    def _check(self):
        if getattr(self, "poisoned_pending", False):
            self.poisoned, self.poisoned_pending = True, False
        if self.poisoned:
            self._fail()

    def read_trace(self, count, first=0, events=True):
        self._check()
        return [self.trace[first + i] for i in range(count)]

    def read_trace_async(self, count, first, into):
        self.inflight = (count, first, into)

    def trace_wait(self):
        if self.inflight is None:
            return
        count, first, into = self.inflight
        self.inflight = None
        self._check()
        into.data = [self.trace[first + i] for i in range(count)]

    def trace_view(self, buf, count):
        return list(buf.data)

    def close(self):
        pass


class _Buf:
    data = None

    def close(self):
        pass


@pytest.fixture
def pinned(monkeypatch):
    import covid19uk_amd.sampler as S
    monkeypatch.setattr(S, "PinnedTrace", lambda sampler, burst, events=True: _Buf())


@pytest.mark.parametrize("fail_at", [(), (1,), (6,), (9, 10), (4, 13, 22), (24,), tuple(range(1, 25, 5))])
def test_overlapped_bursts_are_delivered_once_and_in_order(pinned, fail_at):
    nb, burst = 6, 4
    s = FakeDevice(fail_at=fail_at, cap=2 * burst)
    got = {}

    def consume(tr, i):
        assert i not in got, "a burst was delivered twice"
        got[i] = list(tr)
    s.sample_bursts(nb, burst, consume)
    assert sorted(got) == list(range(nb))
    for i in range(nb):
        assert got[i] == list(range(i * burst + 1, (i + 1) * burst + 1)), (i, got[i])     # the draws of an undisturbed run
    assert s.state == nb * burst
    assert len(s.recoveries) == (len(fail_at) > 0) * len(s.recoveries) and (not fail_at or s.recoveries)
    for r in s.recoveries:
        assert r["rerun_form"] in FALLBACK_FORMS


def test_blocking_sample_recovers_and_returns_to_the_preferred_form(pinned):
    s = FakeDevice(fail_at=(3,), cap=8)
    assert s.sample(4) == [1, 2, 3, 4]
    assert len(s.recoveries) == 1 and s.launch_form() == FALLBACK_FORMS[0]
    # retry_after = 2 clean bursts in the fall-back form (the burst that was run again is the first of them): back
    assert s.sample(4) == [5, 6, 7, 8] and s.launch_form() == ("chunk", "paired")
    assert s.sample(4) == [9, 10, 11, 12] and s.launch_form() == ("chunk", "paired")
    s.fail_at = {14}
    assert s.sample(4) == [13, 14, 15, 16] and s.launch_form() == FALLBACK_FORMS[0]
    assert len(s.recoveries) == 2 and s.retry_after == 4                                # twice as patient the second time
    assert s.sample(4) == [17, 18, 19, 20] and s.launch_form() == FALLBACK_FORMS[0]
    for k in range(2):
        s.sample(4)
    assert s.launch_form() == ("chunk", "paired")


def test_a_burst_that_fails_in_every_form_raises(pinned):
    forms = {("chunk", "paired"), *FALLBACK_FORMS}
    s = FakeDevice(fail_at=(), fail_forms=forms, cap=8)

    def always(n, run=FakeDevice.run):
        s.fail_at = {s.state + 1}
        run(s, n)
    s.run = always
    with pytest.raises(_lib.HandoffTimeout):
        s.sample(4)
    assert len(s.recoveries) == len(FALLBACK_FORMS)          # it went down the whole ladder first


def test_without_auto_recover_the_error_reaches_the_caller(pinned):
    s = FakeDevice(fail_at=(2,), cap=8)
    s.auto_recover = False
    with pytest.raises(_lib.HandoffTimeout):
        s.sample(4)
    s2 = FakeDevice(fail_at=(2,), cap=8)
    s2.auto_recover = False
    with pytest.raises(_lib.HandoffTimeout):
        s2.sample_bursts(3, 4, lambda tr, i: None)
