"""Single-chain T-sharding (last paragraph; covid19uk_amd/tshard.py): host pieces on the CPU,
the sharded evaluation against the unsharded one on the GPU, and a two-process job with real collectives."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from covid19uk_amd import tshard
from oracle import seir_oracle as so
from tests import helpers as H


def test_shard_days_partitions_the_series():
    for T in (2, 7, 365, 730):
        for world in (1, 2, 3, 8):
            if T < world:
                continue
            blocks = [tshard.shard_days(T, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == T
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(b - a for a, b in blocks) - min(b - a for a, b in blocks) <= 1
    with pytest.raises(ValueError):
        tshard.shard_days(3, 4, 0)


def test_host_prior_and_jacobian_match_the_oracle():
    """What is subtracted per shard and added once for the chain: priors (model_spec.py:140-198) + Jacobian
    (inference.py:555-557), value and gradient, against the oracle's joint minus its likelihood."""
    case = H.build_case("micro_5x24", 3, alpha_t_sd=0.005)
    k = case["k"]
    rng = np.random.default_rng(3)
    u = case["u"] + 0.1 * rng.normal(size=k.P)
    u[6:6 + k.T - 1] = 0.005 * rng.normal(size=k.T - 1)
    ev = case["events"]
    lp, g = so.joint_log_prob_and_grad(u, ev, k)
    par = so.unpack(so.constrain(u), k.M, k.T)
    lik = so.seir_log_prob(par, k, ev, "stable")
    from covid19uk_amd import model_spec as ms
    dk = ms.derive_constants(case["cov"])
    pj, gpj = tshard.prior_and_jacobian(u, k.T, dk.car_Q, dk.car_half_logdet)
    assert abs(pj - (lp - lik)) <= 1e-10 * abs(lp)
    # gradient: finite differences of the host function itself
    for i in (0, 1, 2, 3, 5, 6, 6 + k.T - 1, k.P - 1):
        h = 1e-6 * max(1.0, abs(u[i]))
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        fd = (tshard.prior_and_jacobian(up, k.T, dk.car_Q, dk.car_half_logdet)[0]
              - tshard.prior_and_jacobian(um, k.T, dk.car_Q, dk.car_half_logdet)[0]) / (2 * h)
        assert abs(fd - gpj[i]) <= 1e-6 * max(1.0, abs(gpj[i])), i


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("micro_17x70", 3), ("uk380", 4), ("micro_5x24", 8)])
def test_sharded_evaluation_equals_the_unsharded_one(name, world):
    """`world` shards of one chain (driven from one process, the exchanges done by hand) against the plain
    evaluation of the whole series: log-prob 1e-9, gradient 1e-6 -- every parameter block, including the
    alpha_t entries that only later shards depend on."""
    import torch
    assert torch.cuda.is_available()
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    case = H.build_case(name, 21, alpha_t_sd=0.005)
    k = case["k"]
    rng = np.random.default_rng(21)
    u = case["u"] + 0.05 * rng.normal(size=k.P)
    u[6:6 + k.T - 1] = 0.005 * rng.normal(size=k.T - 1)
    with SeirModel(case["cov"], case["init"]) as model:
        want, gw = model.log_prob_grad(u, case["events"])
    shards = [tshard.TShard(case["cov"], case["init"], world, r) for r in range(world)]
    try:
        got, g = tshard.evaluate_in_process(shards, u, case["events"])
    finally:
        for sh in shards:
            sh.close()
    assert abs(got - want) <= 1e-9 * abs(want), (got, want)
    scale = np.maximum(np.abs(gw), 1e-6 * np.abs(gw).max())
    assert np.max(np.abs(g - gw) / scale) < 1e-6


@pytest.mark.gpu
def test_two_rank_sharded_job_with_collectives(tmp_path):
    """Two processes (gloo; both on this box's one GPU), each holding half of the days of the events:
    all_gather of the increments + all_reduce of 1 + P must give both ranks the unsharded answer."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=H.ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(H.ROOT, "tests", "tshard_worker.py"), str(tmp_path)],
                                      env=env, cwd=H.ROOT, stderr=subprocess.PIPE, text=True))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r in res:
        assert abs(r["lp"] - r["want"]) <= 1e-9 * abs(r["want"])
        assert r["grad_err"] < 1e-6
    assert res[0]["lp"] == res[1]["lp"]
