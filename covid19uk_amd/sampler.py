"""`ChainSampler`: host handle on the device-resident Metropolis-within-Gibbs
sampler of libseirhip (C-ABI section "Device-resident Metropolis-within-Gibbs
sampler" of include/seir_hip.h).

It plays the role of the `GibbsKernel` + `tfp.mcmc.sample_chain` pair the
reference builds per window (covid19uk/inference/inference.py:60-242): the host
chooses the window mode (fixed / dual-averaging / dual-averaging + diagonal
mass adaptation), asks for `n` sweeps, and reads draws and kernel-result
traces back.  All sampling arithmetic runs in HIP kernels.
"""
from __future__ import annotations

import ctypes
import dataclasses
import sys

import numpy as np

from . import _lib
from .seir import SeirModel, _dptr

MOVE_KEYS = ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I")   # inference.py:277-280
MOVES_MODES = {"paired": 0, "split": 1, "paired-nopre": 2, "paired-delta": 3, "paired-launch": 4}
HMC_MODES = {"chunk": 0, "single": 1, "chunk-split": 2, "chunk-launch": 3, "chunk-leap": 4, "chunk-stage": 5,
             "chunk-launch-fold": 6}
# Launch forms to fall back on when a hand-off inside a persistent launch times out (the GPU is shared with something):
# first the per-step forms, whose waiting workgroups are always placed behind the ones they wait for, then the forms
# without any hand-off inside a launch.  What is sampled is the same in all of them.
FALLBACK_FORMS = (("chunk-launch", "paired-launch"), ("chunk-split", "paired-delta"))


@dataclasses.dataclass
class Trace:
    """Draws and kernel results of `count` sweeps for B chains (leading axes [count, B])."""
    theta: np.ndarray        # [n,B,P] constrained parameter draws
    events: np.ndarray       # [n,B,M,T,3] int32 or uint16 (or None)
    hmc: dict                # is_accepted, target_log_prob, step_size  -> [n,B]
    moves: dict              # MOVE_KEYS -> dict(is_accepted [n,B], target_log_prob [n,B], proposed_delta [n,B,4,m])


class PinnedTrace:
    """Page-locked host arrays for `count` sweeps of the burst buffer (seir_host_alloc): the target of
    `ChainSampler.read_trace_async`.  Views are valid until close()."""

    def __init__(self, sampler: "ChainSampler", count: int, events: bool = True):
        self._lib = sampler._lib
        self.count = int(count)
        B, P, M, T = sampler.B, sampler.P, sampler.M, sampler.T
        self._ptrs = []
        self.theta = self._alloc((count, B, P), np.float64)
        self.events = self._alloc((count, B, M, T, 3), sampler.events_dtype) if (events and sampler.record_events) else None
        self.hmc = self._alloc((count, B, 3), np.float64)
        self.moves = self._alloc((count, B, 4, _lib.MOVE_TRACE), np.float64)

    def _alloc(self, shape, dtype):
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        _lib.check(self._lib.seir_host_alloc(ctypes.byref(p), max(nbytes, 8)))
        self._ptrs.append(p)
        buf = (ctypes.c_char * max(nbytes, 8)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)

    def close(self):
        self.theta = self.events = self.hmc = self.moves = None
        for p in self._ptrs:
            self._lib.seir_host_free(p)
        self._ptrs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChainSampler:
    def __init__(self, model: SeirModel, config: dict, num_chains: int, seed: int = 0,
                 t_range=None, num_leapfrog_steps: int = 16, trace_capacity: int = 100,
                 first_chain_id: int = 0, record_events: bool = True, moves: str = "paired",
                 hmc: str = "chunk", use_graph: bool = False, chain_groups: int = 1,
                 disable: tuple = (), debug_pair: int = 0, auto_recover: bool = True, log=sys.stderr,
                 leap_rows: int = 0):
        """`config` is the reference's config["Mcmc"] dict: dmax, nmax, m,
        occult_nmax, num_event_time_updates (mcmc_kernel_factory.py:79-81,106,123).

        Launch form (seir_sampler_desc, ABI v2; none of it changes what is sampled):
        `moves` "paired" (k_move_pair with pre-drawn S->E-type proposals, default) | "paired-nopre" |
        "split" (one proposal kernel per update, cross-check);
        `hmc` "chunk" (default) | "single"; `use_graph`; `chain_groups`.
        `disable`: sub-kernels that draw their proposal but always reject, any of
        "hmc", "move/S->E", "move/E->I", "occult/S->E", "occult/E->I" (invariant-distribution tests).

        `auto_recover`: `sample` and `sample_bursts` snapshot the chain state at the start of every burst and, should a
        hand-off inside a persistent launch time out (`_lib.HandoffTimeout`: the launch could not get all its workgroups on
        the GPU -- another sampler or process holds part of it), restore it, switch to the per-step launch forms
        (`FALLBACK_FORMS`), run the burst again and carry on; after `retry_after` clean bursts the preferred forms are tried
        again (twice as many after every further failure).  Every recovery is logged and kept in `self.recoveries`."""
        names = ("hmc",) + MOVE_KEYS
        mask = 0
        for name in disable:
            mask |= 1 << names.index(name)
        self._lib = _lib.load()
        self.model = model
        self.B = int(num_chains)
        self.P, self.M, self.T = model.P, model.M, model.T
        self.mmax = int(config["m"])
        if t_range is None:                       # inference.py:336-339
            t_range = (max(self.T - 21, 0), self.T)
        self.cap = int(trace_capacity)
        # record_events: False, True (int32 counts) or "u16" (uint16 counts: half the burst buffer and half the
        # bytes over PCIe; the read fails loudly if a count exceeds 65535)
        self.record_events = bool(record_events)
        self.events_dtype = np.uint16 if record_events == "u16" else np.int32
        desc = _lib.SeirSamplerDesc(
            num_chains=self.B, dmax=int(config["dmax"]), nmax=int(config["nmax"]), m=self.mmax,
            occult_nmax=int(config["occult_nmax"]),
            num_event_time_updates=int(config["num_event_time_updates"]),
            t_range_lo=int(t_range[0]), t_range_hi=int(t_range[1]),
            num_leapfrog_steps=int(num_leapfrog_steps), trace_capacity=self.cap,
            first_chain_id=int(first_chain_id), record_events=(2 if record_events == "u16" else int(self.record_events)),
            seed=int(seed) & (2 ** 64 - 1),
            moves_mode=MOVES_MODES[moves], hmc_mode=HMC_MODES[hmc],
            use_graph=int(bool(use_graph)), chain_groups=int(chain_groups), disable_mask=mask,
            debug_pair=int(debug_pair), leap_rows=int(leap_rows))
        self._s = ctypes.c_void_p()
        _lib.check(self._lib.seir_sampler_create(model._ctx, ctypes.byref(desc), ctypes.byref(self._s)))
        self.auto_recover = bool(auto_recover) and debug_pair == 0
        self.preferred_form = (hmc, moves)
        self.recoveries = []              # one dict per recovery: burst form that failed, form it was re-run in, message
        self.retry_after = 8              # clean bursts in a fall-back form before the preferred one is tried again
        self._fallback_level = 0          # 0: preferred form; k: FALLBACK_FORMS[k-1]
        self._clean_bursts = 0
        self._log = log

    def close(self):
        for bf in getattr(self, "_pinned", []):
            bf.close()
        self._pinned, self._pinned_key = [], None
        if getattr(self, "_s", None) is not None and self._s:
            self._lib.seir_sampler_destroy(self._s)
            self._s = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state ---------------------------------------------------------------
    def set_state(self, u, events):
        u = np.ascontiguousarray(u, dtype=np.float64)
        ev = np.ascontiguousarray(events, dtype=np.float64)
        if u.shape != (self.B, self.P) or ev.shape != (self.B, self.M, self.T, 3):
            raise ValueError(f"need u [{self.B},{self.P}] and events [{self.B},{self.M},{self.T},3]")
        _lib.check(self._lib.seir_sampler_set_state(self._s, _dptr(u), _dptr(ev)))

    def get_state(self):
        u = np.empty((self.B, self.P))
        ev = np.empty((self.B, self.M, self.T, 3))
        lp = np.empty(self.B)
        _lib.check(self._lib.seir_sampler_get_state(self._s, _dptr(u), _dptr(ev), _dptr(lp)))
        return u, ev, lp

    def log_prob(self):
        lp = np.empty(self.B)
        _lib.check(self._lib.seir_sampler_get_state(self._s, None, None, _dptr(lp)))
        return lp

    def refresh(self):
        _lib.check(self._lib.seir_sampler_refresh(self._s))

    # -- surviving a placement failure of the persistent launches (include/seir_hip.h) ----------------
    def snapshot(self, slot: int = 0):
        """Copy the chain state (everything the next sweep's draws are a function of; not the trace) to shadow slot 0 / 1,
        in stream order."""
        _lib.check(self._lib.seir_sampler_snapshot(self._s, int(slot)))

    def restore(self, slot: int = 0):
        """Back to the snapshot in `slot`; clears a hand-off time-out."""
        _lib.check(self._lib.seir_sampler_restore(self._s, int(slot)))

    def set_launch_form(self, hmc: str, moves: str):
        _lib.check(self._lib.seir_sampler_set_launch_form(self._s, HMC_MODES[hmc], MOVES_MODES[moves]))

    def launch_form(self):
        h, m = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.seir_sampler_launch_form(self._s, ctypes.byref(h), ctypes.byref(m)))
        inv_h = {v: k for k, v in HMC_MODES.items()}
        inv_m = {v: k for k, v in MOVES_MODES.items()}
        return inv_h[h.value], inv_m[m.value]

    def _recover(self, slot: int, err: Exception):
        """A burst failed with a hand-off time-out: back to the snapshot taken at its start, one step down the ladder of
        launch forms.  Raises `err` when there is nothing further down."""
        if self._fallback_level >= len(FALLBACK_FORMS):
            raise err
        failed = self.launch_form()
        self.restore(slot)
        self._fallback_level += 1
        form = FALLBACK_FORMS[self._fallback_level - 1]
        self.set_launch_form(*form)
        self._clean_bursts = 0
        self.retry_after = min(2 * self.retry_after, 4096) if self.recoveries else self.retry_after
        self.recoveries.append(dict(failed_form=failed, rerun_form=form, error=str(err)))
        if self._log is not None:
            print(f"[seir] hand-off time-out in launch form {failed}: the burst's draws are discarded, the chain state is "
                  f"restored from the snapshot taken at its start and the burst runs again as {form}; the preferred form is "
                  f"tried again after {self.retry_after} clean bursts ({err})", file=self._log, flush=True)

    def _burst_ok(self):
        """A burst was delivered: after enough clean ones in a fall-back form, try the preferred form again."""
        if self._fallback_level == 0:
            return
        self._clean_bursts += 1
        if self._clean_bursts >= self.retry_after:
            self._fallback_level = 0
            self._clean_bursts = 0
            self.set_launch_form(*self.preferred_form)
            if self._log is not None:
                print(f"[seir] back to launch form {self.preferred_form}", file=self._log, flush=True)

    # -- kernel parameters -----------------------------------------------------
    def set_kernel(self, step_size=None, variance=None):
        ss = None if step_size is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(step_size, dtype=np.float64), (self.B,)))
        var = None if variance is None else np.ascontiguousarray(
            np.broadcast_to(np.asarray(variance, dtype=np.float64), (self.B, self.P)))
        _lib.check(self._lib.seir_sampler_set_kernel(
            self._s, None if ss is None else _dptr(ss), None if var is None else _dptr(var)))

    def get_kernel(self):
        ss, var = np.empty(self.B), np.empty((self.B, self.P))
        _lib.check(self._lib.seir_sampler_get_kernel(self._s, _dptr(ss), _dptr(var)))
        return ss, var

    def set_adaptation(self, adapt_step_size=False, adapt_mass=False, num_adaptation_steps=0,
                       target_accept_prob=0.75, running_variance=None):
        """running_variance = (count [B], mean [B,P], variance [B,P])."""
        if adapt_mass:
            cnt, mean, var = (np.ascontiguousarray(x, dtype=np.float64) for x in running_variance)
            args = (_dptr(cnt), _dptr(mean), _dptr(var))
        else:
            args = (None, None, None)
        _lib.check(self._lib.seir_sampler_set_adaptation(
            self._s, int(bool(adapt_step_size)), int(bool(adapt_mass)), int(num_adaptation_steps),
            float(target_accept_prob), *args))

    # -- sampling ---------------------------------------------------------------
    def reset_trace(self, at: int = 0):
        """The next sweep is recorded in trace slot `at` (0: start of the burst buffer)."""
        if at:
            _lib.check(self._lib.seir_sampler_reset_trace_at(self._s, int(at)))
        else:
            _lib.check(self._lib.seir_sampler_reset_trace(self._s))

    def run(self, num_sweeps: int):
        """Enqueue sweeps (asynchronous)."""
        _lib.check(self._lib.seir_sampler_run(self._s, int(num_sweeps)))

    def _as_trace(self, n, theta, ev, hmc, mv) -> Trace:
        hmc_d = dict(is_accepted=hmc[..., 0] != 0, target_log_prob=hmc[..., 1], step_size=hmc[..., 2])
        moves = {}
        for i, key in enumerate(MOVE_KEYS):
            delta = mv[:, :, i, 2:].reshape(n, self.B, 4, _lib.MMAX)[..., :self.mmax]
            moves[key] = dict(is_accepted=mv[:, :, i, 0] != 0, target_log_prob=mv[:, :, i, 1],
                              proposed_delta=delta.astype(np.int64))
        return Trace(theta=theta, events=ev, hmc=hmc_d, moves=moves)

    def read_trace(self, count: int, first: int = 0, events: bool = True) -> Trace:
        n = int(count)
        theta = np.empty((n, self.B, self.P))
        ev = np.empty((n, self.B, self.M, self.T, 3), dtype=self.events_dtype) if (events and self.record_events) else None
        hmc = np.empty((n, self.B, 3))
        mv = np.empty((n, self.B, 4, _lib.MOVE_TRACE))
        _lib.check(self._lib.seir_sampler_read_trace(
            self._s, int(first), n, _dptr(theta),
            None if ev is None else ev.ctypes.data_as(ctypes.c_void_p), _dptr(hmc), _dptr(mv)))
        return self._as_trace(n, theta, ev, hmc, mv)

    def read_trace_async(self, count: int, first: int, into: PinnedTrace):
        """Enqueue the device->host copy of trace slots [first, first+count) behind the sweeps queued so far;
        returns at once.  `trace_wait()` then `trace_view(into, count)` give the burst."""
        n = int(count)
        if n > into.count:
            raise ValueError("pinned buffer too small")
        _lib.check(self._lib.seir_sampler_read_trace_async(
            self._s, int(first), n, _dptr(into.theta),
            None if into.events is None else into.events.ctypes.data_as(ctypes.c_void_p),
            _dptr(into.hmc), _dptr(into.moves)))

    def trace_wait(self):
        _lib.check(self._lib.seir_sampler_trace_wait(self._s))

    def trace_view(self, buf: PinnedTrace, count: int) -> Trace:
        n = int(count)
        return self._as_trace(n, buf.theta[:n], None if buf.events is None else buf.events[:n], buf.hmc[:n],
                              buf.moves[:n])

    def sample_bursts(self, num_bursts: int, burst: int, consume, events: bool = True):
        """`num_bursts` x `burst` sweeps with the burst buffer used as two halves: while burst k+1 runs on
        the device, burst k crosses PCIe into page-locked memory on a copy stream and `consume(trace, k)`
        (e.g. the HDF5 writer) runs on a worker thread -- the sampler only waits when the consumer is
        the slower side.  Needs trace_capacity >= 2 * burst.  The trace handed to `consume` is a view of a
        pinned buffer that is re-used two bursts later: copy what must outlive the call."""
        from concurrent.futures import ThreadPoolExecutor
        burst, num_bursts = int(burst), int(num_bursts)
        if 2 * burst > self.cap:
            raise ValueError(f"sample_bursts needs trace_capacity >= 2 * burst = {2 * burst}, have {self.cap}")
        # page-locking GBs of host memory takes tenths of a second: the two buffers are kept for the next call
        key = (burst, bool(events))
        if getattr(self, "_pinned_key", None) != key:
            for bf in getattr(self, "_pinned", []):
                bf.close()
            self._pinned = [PinnedTrace(self, burst, events), PinnedTrace(self, burst, events)]
            self._pinned_key = key
        bufs = self._pinned
        futs = [None, None]
        # i: next burst to enqueue; prev: the burst whose copy is in flight (-1: none).  A hand-off time-out surfaces at
        # trace_wait (or at the next call after it): the oldest burst not yet handed to `consume` is then run again from
        # the snapshot taken at its start, in the next launch form down the ladder (_recover) -- and so is everything
        # enqueued after it, which ran from a state that cannot be trusted.
        i, prev, retried = 0, -1, -1
        try:
            with ThreadPoolExecutor(max_workers=1) as pool:
                while i < num_bursts or prev >= 0:
                    try:
                        h = i & 1
                        if i < num_bursts:
                            if futs[h] is not None:
                                futs[h].result()                 # the consumer is done with host buffer h
                                futs[h] = None
                            if self.auto_recover:
                                self.snapshot(h)                 # stream order: the state burst i starts from
                            self.reset_trace(at=h * burst)
                            self.run(burst)                      # asynchronous
                        if prev >= 0:
                            self.trace_wait()                    # burst prev has landed (it crossed while burst i ran)
                            futs[prev & 1] = pool.submit(consume, self.trace_view(bufs[prev & 1], burst), prev)
                            prev = -1
                            self._burst_ok()
                        if i < num_bursts:
                            self.read_trace_async(burst, h * burst, bufs[h])
                            prev = i
                            i += 1
                    except _lib.HandoffTimeout as e:
                        bad = prev if prev >= 0 else i
                        if not self.auto_recover or bad >= num_bursts:
                            raise
                        if bad == retried and self._fallback_level >= len(FALLBACK_FORMS):
                            raise
                        retried = bad
                        self._recover(bad & 1, e)
                        i, prev = bad, -1
                for f in futs:
                    if f is not None:
                        f.result()
        finally:
            try:
                self.trace_wait()
            except _lib.HandoffTimeout:
                pass

    def sample(self, num_sweeps: int, events: bool = True) -> Trace:
        """reset_trace + run + read: the analogue of one `sample_chain` call."""
        if num_sweeps > self.cap:
            raise ValueError(f"num_sweeps={num_sweeps} exceeds trace_capacity={self.cap}")
        while True:
            if self.auto_recover:
                self.snapshot(0)
            self.reset_trace()
            self.run(num_sweeps)
            try:
                tr = self.read_trace(num_sweeps, events=events)
            except _lib.HandoffTimeout as e:
                if not self.auto_recover:
                    raise
                self._recover(0, e)          # raises e again when no launch form is left to fall back on
                continue
            self._burst_ok()
            return tr

    def xcd_local(self) -> bool:
        """True if workgroups with ids congruent mod 8 share an XCD on this GPU (probed at creation): the condition for
        the chunk roles of an HMC step to run inside the gradient launch (hmc="chunk") and the band workgroups inside
        the pair launch (moves="paired")."""
        return bool(self._lib.seir_sampler_xcd_local(self._s))

    def pair_timeouts(self) -> np.ndarray:
        """Per chain: k_move_pair launches whose authoritative workgroup gave up on the speculative one."""
        out = np.zeros(self.B, dtype=np.uint32)
        _lib.check(self._lib.seir_sampler_pair_timeouts(self._s, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        return out

    def time_leapfrog(self, sweeps: int = 50):
        """(mean ms, launches, gradient evaluations) of the inner leapfrog steps of a sweep -- HIP events around that
        section of `sweeps` ordinary sweeps, on the stream the kernels run on (seir_sampler_time_leapfrog)."""
        ms_, nl, ne = ctypes.c_float(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.seir_sampler_time_leapfrog(self._s, int(sweeps), ctypes.byref(ms_), ctypes.byref(nl), ctypes.byref(ne)))
        return float(ms_.value), int(nl.value), int(ne.value)

    def time_grad_kernel(self, iters: int = 100) -> float:
        """Mean duration (ms) of the sweep's gradient kernel, HIP events on the context stream."""
        ms_ = ctypes.c_float()
        _lib.check(self._lib.seir_sampler_time_grad_kernel(self._s, int(iters), ctypes.byref(ms_)))
        return float(ms_.value)
