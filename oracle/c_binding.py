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
