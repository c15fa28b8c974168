"""CPU-only: the C-ABI library builds, loads and exports every symbol that
include/seir_hip.h declares; the host-side model spec matches the oracle's
derived constants.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry
from covid19uk_amd import _lib, model_spec as ms, synth
from oracle import seir_oracle as so
from tests import helpers as H

ROOT = H.ROOT


@pytest.fixture(scope="module")
def lib():
    entry.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "seir_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(seir_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in seir_hip.h but not exported"
    assert sorted(declared) == _lib.exported_symbols()


def test_abi_version(lib):
    text = open(os.path.join(ROOT, "include", "seir_hip.h")).read()
    declared = int(re.search(r"#define\s+SEIR_ABI_VERSION\s+(\d+)", text).group(1))
    assert lib.seir_abi_version() == declared == _lib.ABI_VERSION == 3


def test_sampler_desc_struct_layout_matches_header():
    # 12 int32, uint64 seed, then the ABI v2 tail: 6 int32 + 2 reserved
    assert ctypes.sizeof(_lib.SeirSamplerDesc) == 12 * 4 + 8 + 8 * 4
    assert _lib.SeirSamplerDesc.seed.offset == 48 and _lib.SeirSamplerDesc.moves_mode.offset == 56


def test_desc_struct_layout_matches_header():
    # 4 int32, 6 pointers, double, pointer, 3 doubles
    assert ctypes.sizeof(_lib.SeirDesc) == 16 + 6 * 8 + 8 + 8 + 3 * 8


def test_create_rejects_bad_arguments_without_touching_the_gpu(lib):
    ctx = ctypes.c_void_p()
    desc = _lib.SeirDesc(M=0, T=5, max_chains=1)
    assert lib.seir_create(ctypes.byref(desc), ctypes.byref(ctx)) == -1
    assert b"must be >= 1" in lib.seir_last_error()
    assert not ctx


def test_missing_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from covid19uk_amd.seir import SeirModel
    cov = synth.make_covariates("ni11")
    _, init, _ = synth.simulate_epidemic(cov)
    with pytest.raises(_lib.SeirError):
        SeirModel(cov, init)


@pytest.mark.parametrize("name", ["ni11", "uk380"])
def test_host_constants_match_oracle(name):
    cov = synth.make_covariates(name)
    _, init, _ = synth.simulate_epidemic(cov)
    k = ms.derive_constants(cov)
    o = so.make_constants(cov.C, cov.N, cov.W, cov.weekday, cov.area, cov.adjacency, init)
    assert np.array_equal(k.Cstar, o.Cstar)
    assert np.array_equal(k.weekday_c, o.weekday_c)
    assert np.allclose(k.log_area_c, o.log_area_c, rtol=0, atol=1e-15)
    assert np.array_equal(k.car_Q, o.Q)
    assert abs(k.car_half_logdet - o.half_logdet_Q) < 1e-10 * abs(o.half_logdet_Q)
    assert ms.num_params(cov.M, cov.T) == o.P


def test_workload_shapes():
    for name, (M, T) in synth.WORKLOADS.items():
        if name == "syn2048":
            continue
        cov = synth.make_covariates(name)
        assert (cov.M, cov.T) == (M, T)
        ev, init, _ = synth.simulate_epidemic(cov)
        st = ms.compute_state(init, ev)
        assert st.min() >= 0 and ev.min() >= 0
        assert np.all(ev[..., 0] <= st[..., 0]) and np.all(ev[..., 1] <= st[..., 1])
        assert np.all(ev[..., 2] <= st[..., 2])


def test_series_constants():
    """The small-rate series in csrc/device_math.h: literals equal the exact Taylor
    coefficients and the truncated series meet 3e-16 on (0, 1/8] (mpmath reference)."""
    import mpmath as mp
    text = open(os.path.join(ROOT, "covid19uk_amd", "csrc", "device_math.h")).read()
    lits = sorted(set(float(x) for x in re.findall(r"\d\.\d{10,}e-\d+", text)))
    exact = sorted([1 / 24, 1 / 2880, 1 / 181440, 1 / 9676800, 1 / 12, 1 / 720, 1 / 30240, 1 / 1209600,
                    1 / 12, 1 / 360, 1 / 1260, 1 / 1680, 1 / 3, 1 / 6, 1 / 7, 2.3190468138462996e-17] +
                   # Bernoulli terms 5..8 of l1me_inv_wide: |B_2n| / (2n (2n)!) and |B_2n| / (2n)!
                   [float(abs(mp.bernoulli(2 * n)) / (2 * n * mp.factorial(2 * n))) for n in range(5, 9)] +
                   [float(abs(mp.bernoulli(2 * n)) / mp.factorial(2 * n)) for n in range(5, 9)])
    for v in lits:
        assert min(abs(v - e) / e for e in exact) < 1e-15, v
    m = re.search(r"L1ME_SERIES_MAX = ([0-9.]+);", text)
    rmax = float(m.group(1))
    mp.mp.dps = 40
    for r in list(np.logspace(-12, np.log10(rmax), 60)) + [rmax]:
        r2 = r * r
        L = np.log(r) + r * (-0.5 + r * (1 / 24 - r2 * (1 / 2880 - r2 * (1 / 181440 - r2 / 9676800))))
        inv = 1 / r - 0.5 + r * (1 / 12 - r2 * (1 / 720 - r2 * (1 / 30240 - r2 / 1209600)))
        Lt = mp.log(1 - mp.e ** (-mp.mpf(r)))
        It = 1 / (mp.e ** mp.mpf(r) - 1)
        assert abs((mp.mpf(L) - Lt) / Lt) < 3e-16 and abs((mp.mpf(inv) - It) / It) < 3e-16
