"""Worker for tests/test_distributed.py: one rank of a gloo job (CPU)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as dist
    from covid19uk_amd import distributed as D
    out_dir, total = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, ws = D.world()
    first, count = D.shard_chains(total, ws, rank)
    # the product driver's view of the same job (covid19uk_amd.inference.inference.mcmc): environment only
    from covid19uk_amd.inference import inference as inf
    lay = inf.job_layout(4)
    assert (lay["rank"], lay["world"], lay["first_chain_id"]) == (rank, ws, 4 * rank)
    assert inf.chain_file_name("p.hd5", lay["first_chain_id"] + 1, 4 * ws) == f"p_chain{4 * rank + 1}.hd5"
    local_eps = np.array([1e-3 * (c + 1) for c in range(first, first + count)])
    gathered = D.gather_chain_values(local_eps)
    pooled = D.pool_step_sizes(local_eps)
    tmax = D.max_over_ranks(0.5 + rank)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(dict(rank=rank, ws=ws, first=first, count=count, gathered=gathered.tolist(),
                       pooled=pooled, tmax=tmax), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
