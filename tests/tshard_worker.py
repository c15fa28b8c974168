"""Worker for tests/test_tshard.py: one rank of a two-process T-sharded evaluation (gloo collectives)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as dist
    from covid19uk_amd import tshard
    from covid19uk_amd.seir import SeirModel
    from tests import helpers as H
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case = H.build_case("micro_17x70", 21, alpha_t_sd=0.005)
    k = case["k"]
    rng = np.random.default_rng(21)
    u = case["u"] + 0.05 * rng.normal(size=k.P)
    u[6:6 + k.T - 1] = 0.005 * rng.normal(size=k.T - 1)
    with tshard.TShard(case["cov"], case["init"], world, rank, device=0) as sh:
        lp, g = sh.log_prob_grad(u, case["events"][:, sh.t0:sh.t1])      # this rank only ever sees its own days
    with SeirModel(case["cov"], case["init"]) as model:
        want, gw = model.log_prob_grad(u, case["events"])
    scale = np.maximum(np.abs(gw), 1e-6 * np.abs(gw).max())
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(dict(lp=lp, want=float(want), grad_err=float(np.max(np.abs(g - gw) / scale))), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
