"""One chain over several GPUs by sharding the T days (last paragraph).

The reference runs one un-batched chain in one process (covid19uk/inference/inference.py:563-576) and has
no counterpart; this is the build's answer to the one size where a single chain is worth several GPUs
(BASELINE config 5: 2048 regions x 730 days).  The per-day likelihood terms of
DiscreteTimeStateTransitionModel.log_prob (call site covid19uk/model_spec.py:278-285) are independent given
the state at the start of each day, so rank r evaluates the days [t0_r, t1_r) on its own GPU through the
unchanged C-ABI, with two small exchanges (one process per GPU, `torch.distributed`: RCCL over xGMI under
"nccl", gloo on CPU):

  1. all_gather of each shard's summed event increments [world, M, 3]: the state at a shard's first day is
     the initial state plus the increments of the shards before it (gemlib compute_state,
     inference.py:500-510);
  2. all_reduce of 1 + P float64: the log-likelihood and its gradient.

A shard is an ordinary `SeirModel` over its own days whose parameter vector is the chain's, re-based: its
alpha_0 is a_{t0} = alpha_0 + sum_{j<t0} alpha_t[j] (model_spec.py:242-256) and its alpha_t the slice
alpha_t[t0 : t1-1].  The joint log-prob such a context returns includes priors and the bijector Jacobian
of those pseudo-parameters; both are closed forms (model_spec.py:140-198, inference.py:525-557), evaluated
here on the host, subtracted per shard and added once for the real parameters after the reduction.
Everything that touches the M x T cells runs in the HIP kernels; this file is host glue.
"""
from __future__ import annotations

import math

import numpy as np

from . import model_spec as ms
from .seir import SeirModel

_EPS = np.finfo(np.float64).eps
_LOG2PI = math.log(2.0 * math.pi)


def shard_days(T: int, world: int, rank: int):
    """Contiguous block of days [t0, t1) of `rank`; blocks differ by at most one day."""
    if not (0 <= rank < world) or T < world:
        raise ValueError(f"cannot shard T={T} days over world={world} (rank {rank})")
    base, extra = divmod(T, world)
    t0 = rank * base + min(rank, extra)
    return t0, t0 + base + (1 if rank < extra else 0)


def prior_and_jacobian(u, T, car_Q, car_half_logdet):
    """Summed prior log-densities of model_spec.py:140-198 plus the inverse log-det-Jacobian of
    inference.py:525-557 at the unconstrained vector u[6 + (T-1) + M], and the gradient w.r.t. u."""
    u = np.asarray(u, dtype=np.float64)
    M = u.shape[0] - 6 - (T - 1)
    sp = np.logaddexp(0.0, u[:2])                      # softplus
    psi, sig = sp[0] + _EPS, sp[1] + _EPS
    beta, g0, g1, a0 = u[2], u[3], u[4], u[5]
    at, s = u[6:6 + T - 1], u[6 + T - 1:]
    Qs = car_Q @ s
    lsig = u[:2] - sp                                  # log sigmoid(u) = u - softplus(u)
    lp = (-math.log(10.0) - 0.5 * _LOG2PI - a0 * a0 / 200.0
          - 0.5 * _LOG2PI - 0.5 * beta * beta
          + 3.0 * math.log(10.0) - math.lgamma(3.0) + 2.0 * math.log(psi) - 10.0 * psi
          - (T - 1) * (math.log(0.005) + 0.5 * _LOG2PI) - 0.5 * float(at @ at) / 0.005 ** 2
          + 0.5 * math.log(2.0 / math.pi) - math.log(0.1) - sig * sig / 0.02
          - 0.5 * float(s @ Qs) + car_half_logdet - 0.5 * M * _LOG2PI
          + 2.0 * (-math.log(100.0) - 0.5 * _LOG2PI) - 0.5 * (g0 * g0 + g1 * g1) / 1.0e4
          + float(lsig.sum()))
    g = np.empty_like(u)
    sg = np.exp(lsig)                                  # d softplus / du
    g[0] = (2.0 / psi - 10.0) * sg[0] + (1.0 - sg[0])
    g[1] = (-sig / 0.01) * sg[1] + (1.0 - sg[1])
    g[2], g[3], g[4], g[5] = -beta, -g0 / 1.0e4, -g1 / 1.0e4, -a0 / 100.0
    g[6:6 + T - 1] = -at / 0.005 ** 2
    g[6 + T - 1:] = -Qs
    return lp, g


class TShard:
    """The days [t0, t1) of one chain on one GPU."""

    def __init__(self, covariates: ms.Covariates, initial_state, world: int, rank: int, device: int = 0):
        self.k = ms.derive_constants(covariates)               # of the FULL series
        self.M, self.T = covariates.M, covariates.T
        self.P = ms.num_params(self.M, self.T)
        self.world, self.rank = int(world), int(rank)
        self.t0, self.t1 = shard_days(self.T, self.world, self.rank)
        self.Ts = self.t1 - self.t0
        self.init_full = np.asarray(initial_state, dtype=np.float64)
        k = self.k
        ks = ms.DerivedConstants(Cstar=k.Cstar, N=k.N, W=np.ascontiguousarray(k.W[self.t0:self.t1]),
                                 weekday_c=np.ascontiguousarray(k.weekday_c[self.t0:self.t1]),
                                 log_area_c=k.log_area_c, car_Q=k.car_Q, car_half_logdet=k.car_half_logdet)
        cs = ms.Covariates(C=covariates.C, W=ks.W, N=covariates.N, adjacency=covariates.adjacency,
                           weekday=ks.weekday_c, area=covariates.area)
        self.model = SeirModel(cs, self.init_full, max_chains=1, device=device, constants=ks)

    def close(self):
        self.model.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- the three local pieces (the collectives sit between them) ---------------------------------
    def increments(self, events_slice):
        """Summed events of this shard's days [M,3] -- the payload of exchange 1."""
        ev = np.asarray(events_slice, dtype=np.float64)
        if ev.shape != (self.M, self.Ts, 3):
            raise ValueError(f"events slice must be [{self.M},{self.Ts},3], got {ev.shape}")
        return ev.sum(axis=1)

    def rebase(self, u):
        """The chain's u[P] as this shard's parameter vector u_s[6 + (Ts-1) + M]."""
        u = np.asarray(u, dtype=np.float64)
        T, t0, Ts = self.T, self.t0, self.Ts
        us = np.empty(6 + (Ts - 1) + self.M)
        us[:5] = u[:5]
        us[5] = u[5] + u[6:6 + t0].sum()                 # a_{t0}
        us[6:6 + Ts - 1] = u[6 + t0:6 + t0 + Ts - 1]
        us[6 + Ts - 1:] = u[6 + T - 1:]
        return us

    def evaluate(self, u, events_slice, increments_before):
        """Log-likelihood of this shard's days and its gradient scattered to the chain's parameters
        (the payload of exchange 2): (lik, G[P]).  increments_before [r, M, 3]: exchange 1's rows of the
        shards before this one."""
        before = np.asarray(increments_before, dtype=np.float64).reshape(-1, self.M, 3).sum(axis=0)
        state0 = self.init_full + before @ ms.STOICHIOMETRY
        if state0.min() < 0:
            return -math.inf, np.zeros(self.P)
        self.model.set_initial_state(state0)
        us = self.rebase(u)
        lp, g = self.model.log_prob_grad(us, np.ascontiguousarray(events_slice, dtype=np.float64))
        pj, gpj = prior_and_jacobian(us, self.Ts, self.k.car_Q, self.k.car_half_logdet)
        lik, gl = lp - pj, g - gpj
        T, t0, Ts = self.T, self.t0, self.Ts
        G = np.zeros(self.P)
        G[:6] = gl[:6]
        G[6:6 + t0] = gl[5]                              # a_{t0} depends on alpha_t[j], j < t0
        G[6 + t0:6 + t0 + Ts - 1] = gl[6:6 + Ts - 1]
        G[6 + T - 1:] = gl[6 + Ts - 1:]
        return float(lik), G

    def finish(self, u, lik_sum, G_sum):
        """Reduced likelihood + the chain's priors and Jacobian: joint_log_prob(u, events) and d/du."""
        pj, gpj = prior_and_jacobian(u, self.T, self.k.car_Q, self.k.car_half_logdet)
        return float(lik_sum + pj), np.asarray(G_sum, dtype=np.float64) + gpj

    # -- with torch.distributed ------------------------------------------------------------------------
    def log_prob_grad(self, u, events_slice):
        """joint_log_prob(u, events) and its gradient, every rank holding the days [t0, t1) of `events`."""
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", self.model.device) if dist.get_backend() == "nccl" else torch.device("cpu")
        inc = torch.from_numpy(self.increments(events_slice)).to(dev)
        incs = [torch.empty_like(inc) for _ in range(self.world)]
        dist.all_gather(incs, inc)                       # exchange 1: [world, M, 3]
        before = np.stack([x.cpu().numpy() for x in incs[:self.rank]]) if self.rank else np.zeros((0, self.M, 3))
        lik, G = self.evaluate(u, events_slice, before)
        buf = torch.from_numpy(np.concatenate([[lik], G])).to(dev)
        dist.all_reduce(buf)                             # exchange 2: 1 + P
        out = buf.cpu().numpy()
        return self.finish(u, out[0], out[1:])


def evaluate_in_process(shards, u, events):
    """All shards of one chain driven from one process (tests, single-GPU rehearsal): the same three
    pieces with the two exchanges done by hand.  events [M,T,3] full."""
    ev = np.asarray(events, dtype=np.float64)
    incs = [sh.increments(ev[:, sh.t0:sh.t1]) for sh in shards]
    lik, G = 0.0, 0.0
    for r, sh in enumerate(shards):
        l, g = sh.evaluate(u, ev[:, sh.t0:sh.t1], np.stack(incs[:r]) if r else np.zeros((0, sh.M, 3)))
        lik, G = lik + l, G + g
    return shards[0].finish(u, lik, G)
