"""CPU oracle of the Metropolis-within-Gibbs sweep (NumPy, fp64, one chain).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by covid19uk_amd/.

PARITY UNPINNED.  The kernels restated here (gemlib's GibbsKernel,
MultiScanKernel, UncalibratedEventTimesUpdate, UncalibratedOccultUpdate;
TFP's PreconditionedHamiltonianMonteCarlo, DualAveragingStepSizeAdaptation,
DiagonalMassMatrixAdaptation, MetropolisHastings) are third-party code that is
not under /root/reference and not installed; the reference seeds no RNG
(inference.py:68,114,134,205,239), so its draws are not reproducible even in
principle.  This file *defines* the build's sampler semantics ("MCMC
kernels") from the reference's call sites:

  kernel wiring / config keys   covid19uk/inference/mcmc_kernel_factory.py:14-168
  sweep composition             covid19uk/inference/inference.py:86-101,219-228
  hmc kwargs, t_range           covid19uk/inference/inference.py:324-339
  trace layout                  covid19uk/inference/inference.py:245-282

and, like the reference, evaluates the FULL joint log-prob (oracle/seir_oracle.py)
for every proposal.  The HIP sampler evaluates the same ratio incrementally;
tests compare the two draw by draw on a shared Philox4x32-10 stream.
"""
from __future__ import annotations

import math

import numpy as np

from . import seir_oracle as so

MMAX = 4
RS_MOMENTUM, RS_HMC_ACCEPT, RS_MOVE_BASE = 0, 1, 16
INF = 0x7FFFFFFF
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over c0 (uint32 array); returns four uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.full_like(c0, c1)
    c2 = np.full_like(c0, c2)
    c3 = np.full_like(c0, c3)
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(_M0) * c0
        p1 = np.uint64(_M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & _MASK
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(_W0)) & _MASK
        k1 = (k1 + np.uint64(_W1)) & _MASK
    return c0, c1, c2, c3


def _u01(hi, lo):
    x = ((hi << np.uint64(32)) | lo) >> np.uint64(12)
    return (x.astype(np.float64) + 0.5) * 2.220446049250313e-16


def _philox_scalar(c0, c1, c2, c3, k0, k1):
    """The same ten rounds on Python ints (an order of magnitude faster than NumPy for one counter)."""
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, \
            ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def rng_uniform2(seed, chain, sweep, stream, idx):
    """Two uniforms per draw slot; idx may be an array."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    if np.ndim(idx) == 0:
        r = _philox_scalar(int(idx) & 0xFFFFFFFF, int(stream), int(sweep), int(chain), k0, k1)
        f = lambda hi, lo: ((((hi << 32) | lo) >> 12) + 0.5) * 2.220446049250313e-16   # noqa: E731
        return np.array([f(r[0], r[1])]), np.array([f(r[2], r[3])])
    r = philox4x32_10(np.atleast_1d(np.asarray(idx, dtype=np.uint64)), stream, sweep, chain, k0, k1)
    return _u01(r[0], r[1]), _u01(r[2], r[3])


def rng_index(u, n):
    return min(int(u * n), n - 1)


class OracleChain:
    """One chain of the sampler with the device's exact draw protocol."""

    def __init__(self, k: so.ModelConstants, config: dict, u, events, seed=0, chain_id=0,
                 t_range=None, num_leapfrog_steps=16, log_prob_fn=None, log_prob_grad_fn=None, disable=()):
        """log_prob_fn(u, events) / log_prob_grad_fn(u, events) default to the NumPy oracle;
        the C restatement (oracle/seir_oracle.c) can be passed for large cases."""
        self.k = k
        self._lp_fn = log_prob_fn or (lambda u_, ev_: so.joint_log_prob(u_, ev_, k, "stable"))
        self._lpg_fn = log_prob_grad_fn or (lambda u_, ev_: so.joint_log_prob_and_grad(u_, ev_, k))
        self.cfg = dict(config)
        # sub-kernels that draw their proposal but always reject (seir_sampler_desc::disable_mask):
        # any of "hmc", "move/S->E", "move/E->I", "occult/S->E", "occult/E->I"
        self.disabled = set(disable)
        self.u = np.array(u, dtype=np.float64)
        self.events = np.array(events, dtype=np.float64)
        self.seed, self.chain, self.sweep = int(seed), int(chain_id), 0
        self.L = int(num_leapfrog_steps)
        self.t_range = t_range if t_range is not None else (max(k.T - 21, 0), k.T)
        self.eps = 0.1
        self.var = np.ones(k.P)
        self.adapt_step = self.adapt_mass = False
        self.n_adapt, self.target = 0, 0.75
        self.da = dict(err=0.0, step=0.0, logavg=0.0, mu=math.log(10 * self.eps))
        self.rv_n, self.rv_mean, self.rv_m2 = 0.0, np.zeros(k.P), np.zeros(k.P)
        self.logp = self._lp_fn(self.u, self.events)
        self.n_evals = 0

    # -- configuration ------------------------------------------------------
    def set_adaptation(self, adapt_step=False, adapt_mass=False, num_adaptation_steps=0,
                       target_accept_prob=0.75, running_variance=None):
        self.adapt_step, self.adapt_mass = bool(adapt_step), bool(adapt_mass)
        self.n_adapt, self.target = int(num_adaptation_steps), float(target_accept_prob)
        self.da = dict(err=0.0, step=0.0, logavg=0.0, mu=math.log(10 * self.eps))
        if adapt_mass:
            cnt, mean, var = running_variance
            self.rv_n, self.rv_mean = float(cnt), np.array(mean, dtype=np.float64)
            self.rv_m2 = np.array(var, dtype=np.float64) * float(cnt)
            self.var = np.array(var, dtype=np.float64)

    def _u2(self, stream, idx):
        a, b = rng_uniform2(self.seed, self.chain, self.sweep, stream, idx)
        return (a, b) if np.ndim(idx) else (float(a[0]), float(b[0]))

    def _lp(self, u, events):
        self.n_evals += 1
        return self._lp_fn(u, events)

    def _lp_grad(self, u):
        self.n_evals += 1
        return self._lpg_fn(u, self.events)

    # -- HMC (PreconditionedHamiltonianMonteCarlo, diag mass M = 1/var) -------
    def hmc_step(self):
        P = self.k.P
        eps, var = self.eps, self.var
        q0 = self.u.copy()
        lp0, g = self._lp_grad(q0)
        u1, u2 = self._u2(RS_MOMENTUM, np.arange((P + 1) // 2))
        rad, ang = np.sqrt(-2.0 * np.log(u1)), 6.283185307179586 * u2
        z = np.empty(2 * len(u1))
        z[0::2], z[1::2] = rad * np.cos(ang), rad * np.sin(ang)
        p = z[:P] / np.sqrt(var)
        k0 = np.sum(0.5 * var * p * p)
        p = p + 0.5 * eps * g
        q = q0 + eps * var * p
        for _ in range(1, self.L):
            _, g = self._lp_grad(q)
            p = p + eps * g
            q = q + eps * var * p
        lp1, g = self._lp_grad(q)
        p = p + 0.5 * eps * g
        k1 = np.sum(0.5 * var * p * p)
        log_ratio = (lp1 - lp0) - (k1 - k0)
        ua, _ = self._u2(RS_HMC_ACCEPT, 0)
        acc = bool(math.log(ua) < log_ratio) and "hmc" not in self.disabled          # NaN -> False
        used_eps = eps
        if acc:
            self.u, self.logp = q, lp1
        else:
            self.logp = lp0
        if self.adapt_step:
            a = min(1.0, math.exp(log_ratio)) if math.isfinite(log_ratio) else 0.0
            d = self.da
            prev = d["step"]
            n = prev + 1.0
            d["err"] += self.target - a
            log_step = d["mu"] - d["err"] * math.sqrt(n) / ((n + 10.0) * 0.05)
            eta = n ** (-0.75)
            d["logavg"] = eta * log_step + (1.0 - eta) * d["logavg"]
            d["step"] = n
            if prev <= self.n_adapt:
                self.eps = math.exp(log_step) if prev < self.n_adapt else math.exp(d["logavg"])
        if self.adapt_mass:
            n1 = self.rv_n + 1.0
            dlt = self.u - self.rv_mean
            self.rv_mean = self.rv_mean + dlt / n1
            self.rv_m2 = self.rv_m2 + dlt * (self.u - self.rv_mean)
            self.rv_n = n1
            self.var = self.rv_m2 / n1
        # step_size as the reference traces it: read from the kernel results after
        # DualAveragingStepSizeAdaptation wrote new_step_size back (inference.py:255-261), i.e. the
        # step size of the NEXT step; `used_step_size` is the one this trajectory ran with
        return dict(is_accepted=acc, target_log_prob=self.logp, step_size=self.eps, used_step_size=used_eps,
                    log_accept_ratio=log_ratio)

    # -- event moves ---------------------------------------------------------
    def _closed_state(self):
        # [M,T+1,4]; recomputed only when the event tensor has changed (an accepted update replaces self.events)
        if getattr(self, "_st_for", None) is not self.events:
            self._st = so.compute_state(self.k.initial_state, self.events, closed=True)
            self._st_for = self.events
        return self._st

    def _mh(self, new_events, valid, logq, logu):
        if valid:
            lp_new = self._lp(self.u, new_events)
            ratio = (lp_new - self.logp) + logq
            acc = bool(logu < ratio)
        else:
            acc = False
        if acc:
            self.events, self.logp = new_events, lp_new
        return acc

    def event_time_move(self, tgt, scan, slot):
        cfg, k = self.cfg, self.k
        T = k.T
        stream = RS_MOVE_BASE + scan * 4 + slot
        logu = math.log(self._u2(stream, 15)[0])
        K = self.events[..., tgt]
        st = self._closed_state()
        hot = np.flatnonzero(K.sum(axis=1) > 0)
        H = len(hot)
        nsel = min(cfg["m"], MMAX, H)
        chosen, valid, logq = [], True, 0.0
        new = self.events.copy()
        tr = np.zeros((4, MMAX), dtype=np.int64)
        for j in range(nsel):
            u_m, u_t = self._u2(stream, 2 * j)
            u_d, u_x = self._u2(stream, 2 * j + 1)
            pos = rng_index(u_m, H - j)
            for a in chosen:
                if pos >= a:
                    pos += 1
            chosen = sorted(chosen + [pos])
            m = int(hot[pos])
            days = np.flatnonzero(K[m] > 0)
            D = len(days)
            t = int(days[rng_index(u_t, D)])
            v = rng_index(u_d, 2 * cfg["dmax"])
            delta = v - cfg["dmax"] if v < cfg["dmax"] else v - cfg["dmax"] + 1
            t2 = t + delta
            if t2 < 0 or t2 >= T:
                valid = False
                tr[:, j] = (m, t, delta, 0)
                continue
            lo, hi = min(t, t2), max(t, t2)
            src, dst = tgt, tgt + 1
            dec, inc = (dst, src) if delta > 0 else (src, dst)
            min_dec = INF if dec == 0 else int(st[m, lo + 1:hi + 1, dec].min())
            min_inc = INF if inc == 0 else int(st[m, lo + 1:hi + 1, inc].min())
            kt, kt2 = int(K[m, t]), int(K[m, t2])
            xmax = max(0, min(cfg["nmax"], kt, min_dec))
            x = rng_index(u_x, xmax + 1)
            Dn = D - (1 if (x > 0 and x == kt) else 0) + (1 if (x > 0 and kt2 == 0) else 0)
            binc = INF if inc == 0 else min_inc + x
            xmax_r = max(0, min(cfg["nmax"], kt2 + x, binc))
            # The Hastings correction pairs this sub-move with its reverse (day t2, shift -delta, same x).
            # A null sub-move (x == 0) leaves the row as it is and is its OWN reverse: correction 0.  (Pairing
            # it with "x = 0 from day t2" -- a day that may hold no event, i.e. an impossible draw -- would put
            # the factor (xmax+1)/(xmax_r+1) != 1 into the acceptance of whatever the OTHER rows of the same
            # proposal do, and the kernel would no longer leave the posterior invariant: found by
            # tests/test_invariance.py on the enumerated toy.)
            if x > 0:
                logq += (-math.log(Dn) - math.log(xmax_r + 1)) - (-math.log(D) - math.log(xmax + 1))
            new[m, t, tgt] -= x
            new[m, t2, tgt] += x
            tr[:, j] = (m, t, delta, x)
        enabled = ("move/S->E", "move/E->I")[tgt] not in self.disabled
        acc = self._mh(new, valid and enabled, logq, logu)
        return dict(is_accepted=acc, target_log_prob=self.logp, proposed_delta=tr[:, :cfg["m"]],
                    log_q_ratio=logq, valid=valid, proposed_events=new)

    def occult_move(self, tgt, scan, slot):
        cfg, k = self.cfg, self.k
        T, M = k.T, k.M
        lo_r, hi_r = self.t_range
        R = hi_r - lo_r
        nmax = cfg["occult_nmax"]
        stream = RS_MOVE_BASE + scan * 4 + slot
        logu = math.log(self._u2(stream, 15)[0])
        u_br, u_m = self._u2(stream, 0)
        u_t, u_x = self._u2(stream, 1)
        K = self.events[..., tgt]
        st = self._closed_state()
        rng_tot = K[:, lo_r:hi_r].sum(axis=1)
        hotrows = np.flatnonzero(rng_tot > 0)
        Hd = len(hotrows)
        is_del = (u_br < 0.5) and Hd > 0
        src, dst = tgt, tgt + 1
        if not is_del:
            m = rng_index(u_m, M)
            t = lo_r + rng_index(u_t, R)
        else:
            m = int(hotrows[rng_index(u_m, Hd)])
        hotdays = np.flatnonzero(K[m, lo_r:hi_r] > 0)
        Dm = len(hotdays)
        if is_del:
            t = lo_r + int(hotdays[rng_index(u_t, Dm)])
        rt_m = int(rng_tot[m])
        min_src = INF if src == 0 else int(st[m, t + 1:T + 1, src].min())
        min_dst = int(st[m, t + 1:T + 1, dst].min())
        kt = int(K[m, t])
        l2, lM, lR = math.log(2.0), math.log(M), math.log(R)
        new = self.events.copy()
        if not is_del:
            xmax = max(0, min(nmax, min_src))
            x = rng_index(u_x, xmax + 1)
            qf = (-l2 if Hd > 0 else 0.0) - lM - lR - math.log(xmax + 1)
            Hd2 = Hd + (1 if (rt_m == 0 and x > 0) else 0)
            Dm2 = Dm + (1 if (kt == 0 and x > 0) else 0)
            xmax_r = max(0, min(nmax, kt + x, min_dst + x))
            qr = (-l2 - math.log(Hd2) - math.log(Dm2) - math.log(xmax_r + 1)) if (Hd2 > 0 and kt + x > 0) \
                else -math.inf
            new[m, t, tgt] += x
        else:
            xmax = max(0, min(nmax, kt, min_dst))
            x = rng_index(u_x, xmax + 1)
            qf = -l2 - math.log(Hd) - math.log(Dm) - math.log(xmax + 1)
            Hd2 = Hd - (1 if (x > 0 and rt_m == x) else 0)
            bs = INF if src == 0 else min_src + x
            xmax_r = max(0, min(nmax, bs))
            qr = (-l2 if Hd2 > 0 else 0.0) - lM - lR - math.log(xmax_r + 1)
            new[m, t, tgt] -= x
        acc = self._mh(new, ("occult/S->E", "occult/E->I")[tgt] not in self.disabled, qr - qf, logu)
        tr = np.zeros((4, MMAX), dtype=np.int64)
        tr[:, 0] = (m, t, -1 if is_del else 1, x)
        return dict(is_accepted=acc, target_log_prob=self.logp, proposed_delta=tr[:, :cfg["m"]],
                    log_q_ratio=qr - qf, valid=True, proposed_events=new)

    # -- one posterior draw --------------------------------------------------
    def sweep_once(self):
        if "hmc" in self.disabled:
            # a disabled HMC update rejects whatever it proposes; its Philox streams (0, 1) are its own,
            # so skipping the trajectory changes nothing downstream
            out = {"hmc": dict(is_accepted=False, target_log_prob=self.logp, step_size=self.eps,
                               used_step_size=self.eps, log_accept_ratio=float("nan"))}
        else:
            out = {"hmc": self.hmc_step()}
        last = {}
        for scan in range(self.cfg["num_event_time_updates"]):
            last["move/S->E"] = self.event_time_move(0, scan, 0)
            last["move/E->I"] = self.event_time_move(1, scan, 1)
            last["occult/S->E"] = self.occult_move(0, scan, 2)
            last["occult/E->I"] = self.occult_move(1, scan, 3)
        out.update(last)
        out["theta"] = so.constrain(self.u)
        out["events"] = self.events.copy()
        self.sweep += 1
        return out
