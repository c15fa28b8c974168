"""GPU end-to-end: the drop-in CLI surface (`python -m covid19uk.inference.inference`) on the
11-LAD example size: data file in, posterior.hd5 out with the reference's layout, sane
acceptance rates and a posterior that recovers the generating parameters."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

from covid19uk_amd import hdf5io, synth
from covid19uk_amd.inference import inference as inf
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_cli_end_to_end_ni11(tmp_path):
    cov = synth.make_covariates("ni11")
    events, init, truth = synth.simulate_epidemic(cov)
    data = str(tmp_path / "inferencedata.nc")
    dates = [str(np.datetime64("2021-01-01") + np.timedelta64(i, "D")) for i in range(cov.T)]
    inf.write_inference_data(data, cov, events[..., 2], dates)
    cfg = {"Mcmc": dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5,
                        num_bursts=3, num_burst_samples=100, thin=1, num_adaptation_iterations=1000)}
    cfg_path = str(tmp_path / "config.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    out = str(tmp_path / "posterior.hd5")
    env = dict(os.environ, PYTHONPATH=H.ROOT)
    r = subprocess.run([sys.executable, "-m", "covid19uk.inference.inference", "-c", cfg_path, "-o", out,
                        "--seed", "4", data], cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Initial logpi" in r.stdout and "Acceptance theta" in r.stdout
    n = 1825 + 300
    with hdf5io.File(out, "r") as f:
        assert f.shape("/samples/seir") == (n, cov.M, cov.T, 3)
        assert f.shape("/samples/alpha_t") == (n, cov.T - 1)
        assert f.shape("/results/move/E->I/proposed_delta") == (n, 4, 2)
        assert f.shape("/initial_state") == (cov.M, 4)
        assert [x.decode() for x in f.read("/time")] == dates
        lp = f.read("/results/hmc/target_log_prob")
        assert np.isfinite(lp).all()
        acc = f.read("/results/hmc/is_accepted")[1825:].mean()
        assert 0.4 < acc <= 1.0, acc
        for key in ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I"):
            a = f.read(f"/results/{key}/is_accepted").mean()
            # with T=32 and dmax=84 most time-moves land outside [0,T) and are auto-rejected
            assert 0.001 < a < 0.99, (key, a)
        seir = f.read("/samples/seir")
        assert np.array_equal(seir[-1][..., 2], events[..., 2])      # removals are data
        g0 = f.read("/samples/gamma0")[1825:]
        assert abs(g0.mean() - truth["gamma0"]) < 0.35, g0.mean()     # I->R rate is well identified
        psi = f.read("/samples/psi")
        assert (psi > 0).all()


def test_two_rank_job_equals_one_process_with_all_chains(tmp_path):
    """SURVEY.md 8e through the product driver: chains {0..3} run as 2 + 2 in two processes (RANK /
    WORLD_SIZE as torchrun sets them; both ranks share this box's one GPU, the step-size exchange goes over
    gloo) must write the same posterior_chain{c}.hd5 files, bit for bit, as one process running all four --
    the draws of a chain depend on its global id only, and the pooled step size on all chains of the job."""
    import socket
    cov = synth.make_covariates("ni11")
    events, init, truth = synth.simulate_epidemic(cov)
    data = str(tmp_path / "inferencedata.nc")
    inf.write_inference_data(data, cov, events[..., 2])
    cfg = {"Mcmc": dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5,
                        num_bursts=2, num_burst_samples=40, thin=1)}
    cfg_path = str(tmp_path / "config.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    base = [sys.executable, "-m", "covid19uk.inference.inference", "-c", cfg_path, "--seed", "11",
            "--pool-step-size", "--device", "0"]
    env = dict(os.environ, PYTHONPATH=H.ROOT)
    one = tmp_path / "one"
    one.mkdir()
    # (the two ranks below are given the same --device: mcmc() then runs the per-step launch forms, two persistent
    # whole-chip launches cannot share a GPU; the reference run is told to use the same forms -- same code, same bits)
    r = subprocess.run(base + ["--chains", "4", "--hmc", "chunk-launch", "--moves", "paired-launch",
                               "-o", str(one / "posterior.hd5"), data], cwd=H.ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ranks of this job share" not in r.stderr
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = tmp_path / "two"
    two.mkdir()
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), SEIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(base + ["--chains", "2", "-o", str(two / "posterior.hd5"), data], cwd=H.ROOT,
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, err[-2000:]
        assert "Pooled step size" in err
        assert "ranks of this job share device 0" in err, err[-2000:]
    n = 1825 + 80
    for c in range(4):
        with hdf5io.File(str(one / f"posterior_chain{c}.hd5"), "r") as fa, \
                hdf5io.File(str(two / f"posterior_chain{c}.hd5"), "r") as fb:
            for name in ("/samples/psi", "/samples/alpha_t", "/samples/spatial_effect", "/samples/seir",
                         "/results/hmc/target_log_prob", "/results/hmc/step_size", "/results/move/E->I/proposed_delta",
                         "/results/occult/S->E/is_accepted"):
                a, b = fa.read(name), fb.read(name)
                assert a.shape[0] == n and np.array_equal(a, b), (c, name)
    # the chains are different chains
    with hdf5io.File(str(two / "posterior_chain0.hd5"), "r") as fa, hdf5io.File(str(two / "posterior_chain3.hd5"), "r") as fb:
        assert not np.array_equal(fa.read("/samples/psi"), fb.read("/samples/psi"))


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment -- the way the driver calls it -- must start its N
    ranks itself (a torch.distributed.run child), relay rank 0's JSON line and pass the exit code on.  Rehearsed with two
    ranks that share this box's one GPU and meet over gloo (BENCH_SHARE_GPU / BENCH_DIST_BACKEND: the collectives of the
    N > 1 path on CPU tensors; RCCL itself needs the driver's multi-GPU node)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=H.ROOT, BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--workload", "ni11", "--chains-per-gpu", "3", "--adapt-sweeps", "5", "--spinup-seconds", "0.05",
                        "--no-cpu-baseline", "--no-cli", "--no-chains-scaling", "--no-egress"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2
    assert out["config"]["chains_total"] == 6 and out["value"] > 0 and out["all_log_probs_finite"]
    assert "launching:" in r.stderr
    # a wrong rank count under an external launcher is refused
    bad = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                         cwd=str(tmp_path), env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode == 2
