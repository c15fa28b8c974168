"""MCMC kernel configuration.

The reference's `covid19uk/inference/mcmc_kernel_factory.py` builds TFP / gemlib
kernel objects (make_hmc_base_kernel :14-29, make_hmc_fast_adapt_kernel :32-44,
make_hmc_slow_adapt_kernel :47-60, make_partially_observed_step :63-86,
make_occults_step :89-113, make_event_multiscan_gibbs_step :116-168).  On the
MI355X path those kernels are HIP code selected by a handful of integers; this
module validates the same configuration keys and hands them over.
"""
from __future__ import annotations

EVENT_KEYS = ("dmax", "nmax", "m", "occult_nmax", "num_event_time_updates")


def hmc_kernel_kwargs_default():
    """hmc_kernel_kwargs of inference.py:324-329."""
    return {"step_size": 0.1, "num_leapfrog_steps": 16}


def event_kernel_config(config: dict) -> dict:
    """The keys make_partially_observed_step / make_occults_step /
    make_event_multiscan_gibbs_step read from config["Mcmc"] (:79-81, :106, :123)."""
    missing = [k for k in EVENT_KEYS if k not in config]
    if missing:
        raise KeyError(f"config['Mcmc'] lacks {missing}")
    out = {k: int(config[k]) for k in EVENT_KEYS}
    if out["m"] < 1 or out["dmax"] < 1 or out["nmax"] < 0 or out["occult_nmax"] < 0:
        raise ValueError(f"invalid event-kernel configuration {out}")
    return out
