"""ctypes binding of libseirhip.so (C-ABI: include/seir_hip.h).

There is no CPU fallback: if the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C
covid19uk_amd/csrc`) or no HIP device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libseirhip.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class SeirDesc(ctypes.Structure):
    """Mirror of `seir_desc` (include/seir_hip.h)."""
    _fields_ = [
        ("M", ctypes.c_int32), ("T", ctypes.c_int32),
        ("max_chains", ctypes.c_int32), ("device", ctypes.c_int32),
        ("Cstar", c_double_p), ("N", c_double_p), ("W", c_double_p),
        ("weekday_c", c_double_p), ("log_area_c", c_double_p),
        ("car_Q", c_double_p), ("car_half_logdet", ctypes.c_double),
        ("init_state", c_double_p),
        ("nu", ctypes.c_double), ("time_delta", ctypes.c_double),
        ("rate_floor", ctypes.c_double),
    ]


class SeirSamplerDesc(ctypes.Structure):
    """Mirror of `seir_sampler_desc` (include/seir_hip.h)."""
    _fields_ = [
        ("num_chains", ctypes.c_int32),
        ("dmax", ctypes.c_int32), ("nmax", ctypes.c_int32), ("m", ctypes.c_int32),
        ("occult_nmax", ctypes.c_int32), ("num_event_time_updates", ctypes.c_int32),
        ("t_range_lo", ctypes.c_int32), ("t_range_hi", ctypes.c_int32),
        ("num_leapfrog_steps", ctypes.c_int32), ("trace_capacity", ctypes.c_int32),
        ("first_chain_id", ctypes.c_int32), ("record_events", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        # ABI v2: launch form and test hooks, all-zero = defaults
        ("moves_mode", ctypes.c_int32), ("hmc_mode", ctypes.c_int32), ("use_graph", ctypes.c_int32),
        ("chain_groups", ctypes.c_int32), ("disable_mask", ctypes.c_int32), ("debug_pair", ctypes.c_int32),
        ("leap_rows", ctypes.c_int32), ("reserved", ctypes.c_int32 * 1),
    ]


class SeirSimDesc(ctypes.Structure):
    """Mirror of `seir_sim_desc` (include/seir_hip.h)."""
    _fields_ = [
        ("num_draws", ctypes.c_int32), ("num_steps", ctypes.c_int32),
        ("first_draw_id", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        ("par", ctypes.POINTER(ctypes.c_double)), ("log_baseline", ctypes.POINTER(ctypes.c_double)),
        ("spatial", ctypes.POINTER(ctypes.c_double)), ("W", ctypes.POINTER(ctypes.c_double)),
        ("weekday_c", ctypes.POINTER(ctypes.c_double)), ("init_state", ctypes.POINTER(ctypes.c_double)),
        ("events", ctypes.POINTER(ctypes.c_double)),
    ]


ABI_VERSION = 3               # SEIR_ABI_VERSION
OPT_DEBUG_SKEW, OPT_XCD_AFFINITY, OPT_GEMM_F32, OPT_EVAL_FORM = 0, 1, 2, 3
MMAX = 4                      # SEIR_MMAX
MOVE_TRACE = 2 + 4 * MMAX     # SEIR_MOVE_TRACE


class SeirError(RuntimeError):
    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


class HandoffTimeout(SeirError):
    """A wait inside one of the persistent launches timed out (SEIR_ERR_STATE from a read of the trace): its
    workgroups were not all resident -- something else holds part of the GPU.  `ChainSampler` recovers from it by
    itself (snapshot, restore, per-step launch forms)."""


# name -> (restype, argtypes); every symbol include/seir_hip.h declares
_SIGNATURES = {
    "seir_abi_version": (ctypes.c_int, []),
    "seir_last_error": (ctypes.c_char_p, []),
    "seir_create": (ctypes.c_int, [ctypes.POINTER(SeirDesc), c_void_pp]),
    "seir_destroy": (None, [ctypes.c_void_p]),
    "seir_num_params": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_set_initial_state": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "seir_log_prob": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, c_double_p, c_double_p, c_double_p]),
    "seir_log_prob_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, c_double_p, c_double_p,
                                          c_double_p, c_double_p]),
    "seir_log_prob_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 4),
    "seir_prepare_events_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "seir_eval_prepared_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 3),
    "seir_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "seir_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]),
    "seir_malloc": (ctypes.c_int, [c_void_pp, ctypes.c_uint64]),
    "seir_free": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_memcpy_h2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "seir_memcpy_d2h": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "seir_timer_start": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_timer_stop": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "seir_time_kernel": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.POINTER(ctypes.c_float)]),
    "seir_selftest_math": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32] + [c_double_p] * 4),
    "seir_selftest_math_wide": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32] + [c_double_p] * 3),
    "seir_reproduction_number": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, c_double_p, c_double_p, c_double_p]),
    "seir_within_between": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, c_double_p, c_double_p, ctypes.c_double,
                                           c_double_p, c_double_p]),
    "seir_simulate": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(SeirSimDesc)]),
    "seir_selftest_binomial": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                              c_double_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int32)]),
    "seir_sampler_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(SeirSamplerDesc), c_void_pp]),
    "seir_sampler_destroy": (None, [ctypes.c_void_p]),
    "seir_sampler_set_state": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    "seir_sampler_get_state": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p]),
    "seir_sampler_set_kernel": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    "seir_sampler_get_kernel": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    "seir_sampler_set_adaptation": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.c_int32, ctypes.c_double, c_double_p, c_double_p,
                                                   c_double_p]),
    "seir_sampler_refresh": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_sampler_reset_trace": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_sampler_reset_trace_at": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "seir_sampler_read_trace_async": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, c_double_p,
                                                     ctypes.c_void_p, c_double_p, c_double_p]),
    "seir_sampler_trace_wait": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_host_alloc": (ctypes.c_int, [c_void_pp, ctypes.c_uint64]),
    "seir_host_free": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_sampler_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "seir_sampler_read_trace": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, c_double_p,
                                               ctypes.c_void_p, c_double_p, c_double_p]),
    "seir_sampler_time_grad_kernel": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32,
                                                     ctypes.POINTER(ctypes.c_float)]),
    "seir_sampler_time_leapfrog": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_float),
                                                  ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "seir_sampler_pair_timeouts": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]),
    "seir_sampler_xcd_local": (ctypes.c_int, [ctypes.c_void_p]),
    "seir_sampler_snapshot": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "seir_sampler_restore": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "seir_sampler_debug_fail_handoff": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "seir_sampler_set_launch_form": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]),
    "seir_sampler_launch_form": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                                                ctypes.POINTER(ctypes.c_int32)]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load libseirhip.so (once).  Raises SeirError if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SeirError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). "
                "This package has no CPU fallback.")
        # PyTorch bundles its own libamdhip64; two HIP runtimes in one process break whichever
        # initialises second (torch.cuda.is_available() turns False).  Importing torch first makes
        # libseirhip's libamdhip64 dependency resolve to the runtime torch already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:                           # pragma: no cover - torch-free hosts
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if lib.seir_abi_version() != ABI_VERSION:
            raise SeirError(f"{LIB_PATH} has ABI {lib.seir_abi_version()}, this binding needs {ABI_VERSION}: rebuild it")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().seir_last_error()
        text = msg.decode() if msg else "?"
        cls = HandoffTimeout if (rc == -3 and "hand-off(s) timed out" in text) else SeirError
        raise cls(f"libseirhip call failed ({rc}): {text}", rc)
