"""ctypes handle on oracle/libseir_oracle.so (the C restatement).  Test
infrastructure only, like everything under oracle/."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libseir_oracle.so")
        src = os.path.join(_HERE, "seir_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        h = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        h.seir_oracle_eval_flat.restype = ctypes.c_double
        h.seir_oracle_eval_flat.argtypes = [ctypes.c_int, ctypes.c_int] + [dp] * 6 + \
            [ctypes.c_double, dp, dp, dp, ctypes.c_int, dp]
        h.seir_oracle_set_threads.argtypes = [ctypes.c_int]
        h.seir_oracle_max_threads.restype = ctypes.c_int
        _LIB = h
    return _LIB


def set_threads(n):
    lib().seir_oracle_set_threads(int(n))


def evaluate(k, u, events, stable=1, want_grad=False):
    """joint log-prob (and gradient) of oracle/seir_oracle.c for ModelConstants k."""
    h = lib()
    dp = ctypes.POINTER(ctypes.c_double)

    def p(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data_as(dp)
    keep = [p(x) for x in (k.Cstar, k.N, k.W, k.weekday_c, k.log_area_c, k.Q)]
    init = p(k.initial_state)
    uu, ev = p(u), p(events)
    g = np.zeros(k.P)
    lp = h.seir_oracle_eval_flat(k.M, k.T, *[x[1] for x in keep], k.half_logdet_Q,
                                 init[1], uu[1], ev[1], int(stable),
                                 g.ctypes.data_as(dp) if want_grad else None)
    return (lp, g) if want_grad else lp


# ---------------------------------------------------------------------------------------------
# The sampler in C (oracle/mcmc_oracle.c): the same interface as oracle/mcmc_oracle.py's OracleChain
# ---------------------------------------------------------------------------------------------
MC_MMAX = 4
MOVE_KEYS = ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I")
DISABLE_BITS = {"hmc": 1, "move/S->E": 2, "move/E->I": 4, "occult/S->E": 8, "occult/E->I": 16}


class _HmcResult(ctypes.Structure):
    _fields_ = [("is_accepted", ctypes.c_int), ("target_log_prob", ctypes.c_double), ("step_size", ctypes.c_double),
                ("used_step_size", ctypes.c_double), ("log_accept_ratio", ctypes.c_double)]


class _MoveResult(ctypes.Structure):
    _fields_ = [("is_accepted", ctypes.c_int), ("valid", ctypes.c_int), ("target_log_prob", ctypes.c_double),
                ("log_q_ratio", ctypes.c_double), ("proposed_delta", (ctypes.c_int64 * MC_MMAX) * 4)]


class _SweepResult(ctypes.Structure):
    _fields_ = [("hmc", _HmcResult), ("move", _MoveResult * 4)]


def _mc_lib():
    h = lib()
    if not getattr(h, "_mc_ready", False):
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        h.mcmc_oracle_create.restype = ctypes.c_void_p
        h.mcmc_oracle_create.argtypes = [ctypes.c_int, ctypes.c_int] + [dp] * 6 + [ctypes.c_double, dp, dp, dp, ip,
                                                                                  ctypes.c_uint64, ctypes.c_uint32]
        h.mcmc_oracle_destroy.argtypes = [ctypes.c_void_p]
        h.mcmc_oracle_set_eps.argtypes = [ctypes.c_void_p, ctypes.c_double]
        h.mcmc_oracle_set_adaptation.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                 ctypes.c_double, dp, dp]
        h.mcmc_oracle_sweep.argtypes = [ctypes.c_void_p, ctypes.POINTER(_SweepResult)]
        h.mcmc_oracle_run.argtypes = [ctypes.c_void_p, ctypes.c_int]
        h.mcmc_oracle_get_state.argtypes = [ctypes.c_void_p, dp, dp, dp]
        h.mcmc_oracle_get_variance.argtypes = [ctypes.c_void_p, dp]
        h._mc_ready = True
    return h


class COracleChain:
    """oracle/mcmc_oracle.c behind OracleChain's interface (sweep_once() returns the same dict; `eps`, `var`, `logp`,
    `n_evals`, `u`, `events` read the chain's state)."""

    def __init__(self, k, config, u, events, seed=0, chain_id=0, t_range=None, num_leapfrog_steps=16, disable=()):
        self._h = _mc_lib()
        self.k, self.cfg = k, dict(config)
        dp = ctypes.POINTER(ctypes.c_double)
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in
                (k.Cstar, k.N, k.W, k.weekday_c, k.log_area_c, k.Q, k.initial_state, u, events)]
        t_range = t_range if t_range is not None else (max(k.T - 21, 0), k.T)
        mask = 0
        for name in disable:
            mask |= DISABLE_BITS[name]
        cfg = (ctypes.c_int * 9)(int(config["dmax"]), int(config["nmax"]), int(config["m"]), int(config["occult_nmax"]),
                                 int(config["num_event_time_updates"]), int(t_range[0]), int(t_range[1]),
                                 int(num_leapfrog_steps), mask)
        p = [a.ctypes.data_as(dp) for a in arrs]
        self._c = self._h.mcmc_oracle_create(k.M, k.T, p[0], p[1], p[2], p[3], p[4], p[5], float(k.half_logdet_Q), p[6],
                                             p[7], p[8], cfg, int(seed) & (2 ** 64 - 1), int(chain_id))
        self._res = _SweepResult()

    def __del__(self):
        if getattr(self, "_c", None):
            self._h.mcmc_oracle_destroy(self._c)
            self._c = None

    def _scal(self):
        s = np.zeros(3)
        self._h.mcmc_oracle_get_state(self._c, None, None, s.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return s

    @property
    def eps(self):
        return float(self._scal()[1])

    @eps.setter
    def eps(self, v):
        self._h.mcmc_oracle_set_eps(self._c, float(v))

    @property
    def logp(self):
        return float(self._scal()[0])

    @property
    def n_evals(self):
        return int(self._scal()[2])

    @property
    def u(self):
        u = np.empty(self.k.P)
        self._h.mcmc_oracle_get_state(self._c, u.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), None, None)
        return u

    @property
    def events(self):
        ev = np.empty((self.k.M, self.k.T, 3))
        self._h.mcmc_oracle_get_state(self._c, None, ev.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), None)
        return ev

    @property
    def var(self):
        v = np.empty(self.k.P)
        self._h.mcmc_oracle_get_variance(self._c, v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return v

    def set_adaptation(self, adapt_step=False, adapt_mass=False, num_adaptation_steps=0, target_accept_prob=0.75,
                       running_variance=None):
        dp = ctypes.POINTER(ctypes.c_double)
        if adapt_mass:
            cnt, mean, var = running_variance
            mean, var = (np.ascontiguousarray(x, dtype=np.float64) for x in (mean, var))
            self._h.mcmc_oracle_set_adaptation(self._c, int(bool(adapt_step)), 1, int(num_adaptation_steps),
                                               float(target_accept_prob), float(cnt), mean.ctypes.data_as(dp),
                                               var.ctypes.data_as(dp))
        else:
            self._h.mcmc_oracle_set_adaptation(self._c, int(bool(adapt_step)), 0, int(num_adaptation_steps),
                                               float(target_accept_prob), 0.0, None, None)

    def run(self, n):
        """n sweeps with nothing but C in the loop."""
        self._h.mcmc_oracle_run(self._c, int(n))

    def sweep_once(self):
        from . import seir_oracle as so
        self._h.mcmc_oracle_sweep(self._c, ctypes.byref(self._res))
        r = self._res
        out = {"hmc": dict(is_accepted=bool(r.hmc.is_accepted), target_log_prob=r.hmc.target_log_prob,
                           step_size=r.hmc.step_size, used_step_size=r.hmc.used_step_size,
                           log_accept_ratio=r.hmc.log_accept_ratio)}
        m = int(self.cfg["m"])
        for i, key in enumerate(MOVE_KEYS):
            mv = r.move[i]
            delta = np.array([[mv.proposed_delta[a][b] for b in range(MC_MMAX)] for a in range(4)], dtype=np.int64)
            out[key] = dict(is_accepted=bool(mv.is_accepted), target_log_prob=mv.target_log_prob,
                            proposed_delta=delta[:, :m], log_q_ratio=mv.log_q_ratio, valid=bool(mv.valid))
        out["theta"] = so.constrain(self.u)
        out["events"] = self.events
        return out
