"""`SeirModel`: the device-resident counterpart of the reference's
`model = CovidUK(covariates, initial_state, initial_step=0, num_steps=T)`
together with the `joint_log_prob(unconstrained_params, events)` closure built
on it (covid19uk/inference/inference.py:518-557).

All arithmetic happens in libseirhip.so (HIP, gfx950) through the C-ABI of
include/seir_hip.h; this class only marshals arrays.  PyTorch is used purely as
a device-memory container for the `*_dev` methods.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from . import model_spec as ms


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_double_p)


class SeirModel:
    def __init__(self, covariates: ms.Covariates, initial_state, max_chains: int = 1, device: int = 0,
                 constants: ms.DerivedConstants = None):
        """`constants`: pre-derived closure constants to use instead of `derive_constants(covariates)` -- a
        T-shard passes the slices of the FULL series' constants (the weekday is centred over all days,
        model_spec.py:224-225), see covid19uk_amd/tshard.py."""
        self._lib = _lib.load()
        self._ctx = ctypes.c_void_p()
        k = ms.derive_constants(covariates) if constants is None else constants
        self.M, self.T = covariates.M, covariates.T
        self.P = ms.num_params(self.M, self.T)
        self.max_chains = int(max_chains)
        self.device = int(device)
        init = np.ascontiguousarray(initial_state, dtype=np.float64)
        if init.shape != (self.M, 4):
            raise ValueError(f"initial_state must be [M,4]=({self.M},4), got {init.shape}")
        self.initial_state = init
        keep = dict(Cstar=np.ascontiguousarray(k.Cstar), N=np.ascontiguousarray(k.N),
                    W=np.ascontiguousarray(k.W), wd=np.ascontiguousarray(k.weekday_c),
                    la=np.ascontiguousarray(k.log_area_c), Q=np.ascontiguousarray(k.car_Q))
        desc = _lib.SeirDesc(
            M=self.M, T=self.T, max_chains=self.max_chains, device=self.device,
            Cstar=_dptr(keep["Cstar"]), N=_dptr(keep["N"]), W=_dptr(keep["W"]),
            weekday_c=_dptr(keep["wd"]), log_area_c=_dptr(keep["la"]), car_Q=_dptr(keep["Q"]),
            car_half_logdet=k.car_half_logdet, init_state=_dptr(init),
            nu=ms.NU, time_delta=ms.TIME_DELTA, rate_floor=ms.RATE_FLOOR)
        _lib.check(self._lib.seir_create(ctypes.byref(desc), ctypes.byref(self._ctx)))
        assert self._lib.seir_num_params(self._ctx) == self.P

    def set_initial_state(self, initial_state):
        """Replace S,E,I,R at the first day (seir_set_initial_state)."""
        init = np.ascontiguousarray(initial_state, dtype=np.float64)
        if init.shape != (self.M, 4):
            raise ValueError(f"initial_state must be [M,4]=({self.M},4), got {init.shape}")
        _lib.check(self._lib.seir_set_initial_state(self._ctx, _dptr(init)))
        self.initial_state = init

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.seir_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- host-array entry points ---------------------------------------------
    def _batch(self, u, events):
        u = np.ascontiguousarray(u, dtype=np.float64)
        ev = np.ascontiguousarray(events, dtype=np.float64)
        single = u.ndim == 1
        if single:
            u, ev = u[None], ev[None]
        B = u.shape[0]
        if u.shape != (B, self.P):
            raise ValueError(f"u must be [B,{self.P}], got {u.shape}")
        if ev.shape != (B, self.M, self.T, 3):
            raise ValueError(f"events must be [B,{self.M},{self.T},3], got {ev.shape}")
        return u, ev, B, single

    def log_prob(self, u, events):
        """joint_log_prob(u, events) (inference.py:537-557); batched over a leading axis."""
        u, ev, B, single = self._batch(u, events)
        out = np.empty(B)
        _lib.check(self._lib.seir_log_prob(self._ctx, B, _dptr(u), _dptr(ev), _dptr(out)))
        return float(out[0]) if single else out

    def log_prob_grad(self, u, events):
        """Value and d/du (the reference differentiates joint_log_prob with TF autodiff)."""
        u, ev, B, single = self._batch(u, events)
        out, g = np.empty(B), np.empty((B, self.P))
        _lib.check(self._lib.seir_log_prob_grad(self._ctx, B, _dptr(u), _dptr(ev), _dptr(out), _dptr(g)))
        return (float(out[0]), g[0]) if single else (out, g)

    # -- device-tensor entry points (torch tensors on cuda:<device>, float64) --
    @staticmethod
    def _tptr(t, shape):
        import torch
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError("expected a contiguous float64 CUDA tensor")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return ctypes.c_void_p(t.data_ptr())

    def _torch_ready(self):
        import torch
        torch.cuda.current_stream(self.device).synchronize()

    def log_prob_dev(self, u_t, events_t, logp_t, grad_t=None):
        """Asynchronous on the context stream; call sync() before reading outputs."""
        B = u_t.shape[0]
        self._torch_ready()
        _lib.check(self._lib.seir_log_prob_dev(
            self._ctx, B, self._tptr(u_t, (B, self.P)), self._tptr(events_t, (B, self.M, self.T, 3)),
            self._tptr(logp_t, (B,)), None if grad_t is None else self._tptr(grad_t, (B, self.P))))

    def prepare_events_dev(self, events_t):
        B = events_t.shape[0]
        self._torch_ready()
        _lib.check(self._lib.seir_prepare_events_dev(
            self._ctx, B, self._tptr(events_t, (B, self.M, self.T, 3))))

    def eval_prepared_dev(self, u_t, logp_t, grad_t=None):
        B = u_t.shape[0]
        _lib.check(self._lib.seir_eval_prepared_dev(
            self._ctx, B, self._tptr(u_t, (B, self.P)), self._tptr(logp_t, (B,)),
            None if grad_t is None else self._tptr(grad_t, (B, self.P))))

    def sync(self):
        _lib.check(self._lib.seir_sync(self._ctx))

    def set_option(self, debug_skew=None, xcd_affinity=None, gemm_f32=None, eval_form=None):
        """Launch options of the context (seir_set_option): the workgroup-timing test hook, the
        chain <-> XCD block mapping and the launch form of `log_prob_dev` ("fused" -- one launch for 8 or 16 chains where the GPU
        allows it, else three -- | "three-launch" | "four-launch"; none of
        them changes a result beyond summation order), and `gemm_f32`: the mobility contraction
        with fp32 operands on the fp32 matrix instruction (BASELINE config 5; ~1e-8 relative on the log-prob)."""
        if eval_form is not None:
            form = {"fused": 0, "four-launch": 1, "three-launch": 2}[eval_form]
            _lib.check(self._lib.seir_set_option(self._ctx, _lib.OPT_EVAL_FORM, form))
        if gemm_f32 is not None:
            _lib.check(self._lib.seir_set_option(self._ctx, _lib.OPT_GEMM_F32, int(bool(gemm_f32))))
        if debug_skew is not None:
            _lib.check(self._lib.seir_set_option(self._ctx, _lib.OPT_DEBUG_SKEW, int(debug_skew)))
        if xcd_affinity is not None:
            _lib.check(self._lib.seir_set_option(self._ctx, _lib.OPT_XCD_AFFINITY, int(xcd_affinity)))

    # -- timing (HIP events on the context stream) ----------------------------
    def timer_start(self):
        _lib.check(self._lib.seir_timer_start(self._ctx))

    def timer_stop(self) -> float:
        ms_ = ctypes.c_float()
        _lib.check(self._lib.seir_timer_stop(self._ctx, ctypes.byref(ms_)))
        return float(ms_.value)

    KERNELS = {"scan": 0, "gemm": 1, "se_value": 2, "se_grad": 3, "finish": 4,
               "state": 5, "tiles_value": 6, "tiles_grad": 7, "finish_fused": 8}

    def time_kernel(self, which: str, B: int, iters: int = 50) -> float:
        """Mean launch duration (ms) of one kernel of the last evaluation."""
        ms_ = ctypes.c_float()
        _lib.check(self._lib.seir_time_kernel(self._ctx, self.KERNELS[which], B, iters, ctypes.byref(ms_)))
        return float(ms_.value)

    def selftest_math(self, x):
        """Device values of log(1-exp(-x)), 1/expm1(x), log Gamma(floor(x)+1) (csrc/device_math.h)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        L, inv, lf = np.empty_like(x), np.empty_like(x), np.empty_like(x)
        _lib.check(self._lib.seir_selftest_math(self._ctx, x.size, _dptr(x), _dptr(L), _dptr(inv), _dptr(lf)))
        return L, inv, lf

    def selftest_math_wide(self, x):
        """Device values of log(1-exp(-x)), 1/expm1(x) from the 8-term series (l1me_inv_wide)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        L, inv = np.empty_like(x), np.empty_like(x)
        _lib.check(self._lib.seir_selftest_math_wide(self._ctx, x.size, _dptr(x), _dptr(L), _dptr(inv)))
        return L, inv

    def within_between(self, psi, I_last, W):
        """(within, between) fractions [n,M] of the infection pressure of the last state
        (covid19uk/posterior/within_between.py:13-57).  psi [n], I_last [n,M], W scalar."""
        psi = np.ascontiguousarray(psi, dtype=np.float64).reshape(-1)
        I = np.ascontiguousarray(I_last, dtype=np.float64)
        n = psi.shape[0]
        if I.shape != (n, self.M):
            raise ValueError(f"need psi [n] and I_last [n,{self.M}]")
        wi, be = np.empty((n, self.M)), np.empty((n, self.M))
        _lib.check(self._lib.seir_within_between(self._ctx, n, _dptr(psi), _dptr(I), float(W), _dptr(wi), _dptr(be)))
        return wi, be

    def simulate(self, par, log_baseline, spatial, W, weekday_c, init_state, seed=0, first_draw_id=0):
        """Chain-binomial forward simulation of n draws (DiscreteTimeStateTransitionModel.sample as
        used by covid19uk/posterior/predict.py:50-70): events [n,M,S,3].

        par [n,5] = psi, sigma_space, beta_area, gamma0, gamma1; log_baseline [n,S] = a_t of each
        simulated day; spatial [n,M]; W, weekday_c [S]; init_state [n,M,4]."""
        par = np.ascontiguousarray(par, dtype=np.float64)
        ap = np.ascontiguousarray(log_baseline, dtype=np.float64)
        sp = np.ascontiguousarray(spatial, dtype=np.float64)
        W = np.ascontiguousarray(W, dtype=np.float64)
        wd = np.ascontiguousarray(weekday_c, dtype=np.float64)
        init = np.ascontiguousarray(init_state, dtype=np.float64)
        n = par.shape[0]
        S = ap.shape[1] if ap.ndim == 2 else -1
        if par.shape != (n, 5) or ap.shape != (n, S) or sp.shape != (n, self.M) or W.shape != (S,) \
                or wd.shape != (S,) or init.shape != (n, self.M, 4):
            raise ValueError(f"need par [n,5], log_baseline [n,S], spatial [n,{self.M}], W [S], weekday_c [S], "
                             f"init_state [n,{self.M},4]")
        out = np.empty((n, self.M, S, 3))
        desc = _lib.SeirSimDesc(num_draws=n, num_steps=S, first_draw_id=int(first_draw_id), reserved=0,
                                seed=int(seed) & 0xFFFFFFFFFFFFFFFF, par=_dptr(par), log_baseline=_dptr(ap),
                                spatial=_dptr(sp), W=_dptr(W), weekday_c=_dptr(wd), init_state=_dptr(init),
                                events=_dptr(out))
        _lib.check(self._lib.seir_simulate(self._ctx, ctypes.byref(desc)))
        return out

    def selftest_binomial(self, n, p, seed=0):
        """Device Binomial(n_i, p_i) variates from the simulator's sampler (csrc/sim_kernels.h)."""
        n = np.ascontiguousarray(n, dtype=np.int32)
        p = np.ascontiguousarray(p, dtype=np.float64)
        out = np.empty(n.size, dtype=np.int32)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.check(self._lib.seir_selftest_binomial(self._ctx, n.size, n.ctypes.data_as(ip), _dptr(p),
                                                    int(seed) & 0xFFFFFFFFFFFFFFFF, out.ctypes.data_as(ip)))
        return out

    def reproduction_number(self, theta, events):
        """R_it [n,T,M] for n posterior draws: column sums of the next-generation matrix
        (covid19uk/posterior/reproduction_number.py:13-44, model_spec.py:302-368).
        theta [n,P] constrained draws, events [n,M,T,3]."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        ev = np.ascontiguousarray(events, dtype=np.float64)
        n = th.shape[0]
        if th.shape != (n, self.P) or ev.shape != (n, self.M, self.T, 3):
            raise ValueError(f"need theta [n,{self.P}] and events [n,{self.M},{self.T},3]")
        out = np.empty((n, self.T, self.M))
        _lib.check(self._lib.seir_reproduction_number(self._ctx, n, _dptr(th), _dptr(ev), _dptr(out)))
        return out
