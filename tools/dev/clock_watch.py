#!/usr/bin/env python3
"""Developer probe: the shader clock the driver reports (sysfs / rocm-smi) while idle, during 1-sweep bursts with a
synchronisation after each, and during one long run."""
import glob, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as entry
entry.build()
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel


def sclk():
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if "*" in line:
                    out.append(line.strip())
        except OSError as e:
            out.append(f"{f}: {e}")
    if not out:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20)
            out = [l.strip() for l in r.stdout.splitlines() if "sclk" in l.lower()][:2]
        except Exception as e:  # noqa: BLE001
            out = [repr(e)]
    return out


cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
B = 8
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
print("idle:", sclk())
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=100, record_events=False) as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        samples = []
        stop = False

        def watch():
            while not stop:
                samples.append((time.perf_counter(), sclk()))
                time.sleep(0.05)
        th = threading.Thread(target=watch); th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            s.reset_trace(); s.run(1); model.sync()
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 1.0:
            s.reset_trace(); s.run(400); model.sync()
        t2 = time.perf_counter()
        stop = True; th.join()
        import re
        def mhz(v):
            return [int(re.search(r"(\d+)Mhz", x).group(1)) if re.search(r"(\d+)Mhz", x) else -1 for x in v]
        rows = np.array([mhz(v) for t, v in samples if len(v) == len(samples[0][1])])
        ts = np.array([t for t, v in samples if len(v) == len(samples[0][1])])
        # the card whose reading differs most between the two phases is the one this process runs on
        a, b = rows[ts < t1], rows[ts >= t1]
        col = int(np.argmax(np.abs(np.median(a, axis=0) - np.median(b, axis=0))))
        print("column", col, "of", rows.shape[1], "(this process's GPU, by its behaviour)")
        print("1-sweep bursts with a synchronisation after each, MHz every 50 ms:", a[:, col].tolist())
        print("runs of 400 sweeps, MHz every 50 ms:", b[:, col].tolist())
