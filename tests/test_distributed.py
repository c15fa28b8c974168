"""CPU, world_size 2 (gloo): chain sharding, step-size pooling and the max-over-ranks timing
used by bench.py and the multi-GPU driver."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from covid19uk_amd import distributed as D
from tests import helpers as H


def test_shard_chains_partitions_exactly():
    for total in (0, 1, 7, 8, 64, 65):
        for ws in (1, 2, 3, 8):
            blocks = [D.shard_chains(total, ws, r) for r in range(ws)]
            ids = [c for first, n in blocks for c in range(first, first + n)]
            assert ids == list(range(total))
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
    assert D.shard_chains(64, 8, 3) == (24, 8)
    with pytest.raises(ValueError):
        D.shard_chains(8, 2, 2)


def test_single_process_fallbacks():
    assert D.world() == (0, 1)
    assert np.array_equal(D.gather_chain_values([1.0, 2.0]), [1.0, 2.0])
    assert abs(D.pool_step_sizes([1e-2, 1e-4]) - 1e-3) < 1e-15
    assert D.max_over_ranks(3.5) == 3.5


@pytest.mark.parametrize("total", [8, 5])
def test_two_rank_gloo_job(tmp_path, total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=H.ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(H.ROOT, "tests", "dist_worker.py"),
                                       str(tmp_path), str(total)], env=env, cwd=H.ROOT))
    for p in procs:
        assert p.wait(timeout=120) == 0
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    want = [1e-3 * (c + 1) for c in range(total)]
    for r in res:
        assert r["ws"] == 2
        assert np.allclose(r["gathered"], want)            # global chain order, uneven shards too
        assert abs(r["pooled"] - np.exp(np.mean(np.log(want)))) < 1e-15
        assert r["tmax"] == 1.5
    assert res[0]["first"] == 0 and res[1]["first"] == res[0]["count"]
    assert res[0]["count"] + res[1]["count"] == total
