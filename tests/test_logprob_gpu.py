"""GPU parity: libseirhip's joint log-probability and gradient, called through
the C-ABI, against the CPU oracle on the same seeded inputs.

Tolerances (fp64): |dlogp| <= 1e-9 |logp| (measured ~1e-13),
gradient rtol 1e-6 against the analytic oracle gradient (measured ~1e-11)."""
import numpy as np
import pytest

from covid19uk_amd import synth
from oracle import seir_oracle as so
from tests import helpers as H

pytestmark = pytest.mark.gpu

RTOL_LOGP = 1e-9
RTOL_GRAD = 1e-6


@pytest.fixture(scope="module")
def Model():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd.seir import SeirModel
    return SeirModel


def _batch(case, B, seed):
    T = case["k"].T
    u = synth.jitter_params(case["u"], B, seed=seed, T=T)
    rng = np.random.default_rng(seed)
    u[:, 6:6 + T - 1] = 0.005 * rng.normal(size=(B, T - 1))
    ev = np.stack([case["events"]] * B)
    return u, ev


def _check(case, model, u, ev, grad=True, use_c=False):
    k = case["k"]
    if grad:
        lp, g = model.log_prob_grad(u, ev)
    else:
        lp, g = model.log_prob(u, ev), None
    for b in range(u.shape[0]):
        if use_c:
            want, gw = H.c_oracle_eval(k, u[b], ev[b], stable=1, want_grad=True)
        else:
            want, gw = so.joint_log_prob_and_grad(u[b], ev[b], k)
        assert abs(lp[b] - want) <= RTOL_LOGP * abs(want), (b, lp[b], want)
        if grad:
            scale = np.maximum(np.abs(gw), 1e-6 * np.abs(gw).max())
            err = np.max(np.abs(g[b] - gw) / scale)
            assert err < RTOL_GRAD, (b, err)
    return lp


@pytest.mark.parametrize("name,seed,B", [("micro_1x1", 1, 1), ("micro_2x3", 2, 3), ("micro_3x5", 3, 2),
                                         ("micro_17x70", 4, 2), ("ni11", 5, 16)])
def test_small_cases_match_numpy_oracle(Model, name, seed, B):
    case = H.build_case(name, seed, alpha_t_sd=0.005)
    u, ev = _batch(case, B, seed)
    with Model(case["cov"], case["init"], max_chains=B) as model:
        lp = _check(case, model, u, ev, grad=True)
        lp2 = model.log_prob(u, ev)
        assert np.array_equal(lp, lp2), "value-only and value+grad paths disagree"


def test_value_only_matches_reference_formulation(Model):
    # the oracle's 'reference' form evaluates log(1-(1-e^-r)) naively as TFP does
    case = H.build_case("ni11", 6)
    u, ev = _batch(case, 4, 6)
    with Model(case["cov"], case["init"], max_chains=4) as model:
        lp = model.log_prob(u, ev)
    for b in range(4):
        want = so.joint_log_prob(u[b], ev[b], case["k"], "reference")
        assert abs(lp[b] - want) <= RTOL_LOGP * abs(want)


def test_distinct_events_per_chain(Model):
    case = H.build_case("ni11", 7)
    u, ev = _batch(case, 3, 7)
    other = H.build_case("ni11", 8)
    ev[1] = other["events"]          # feasible under its own init only -> may be -inf; use own init
    ev[2, 3, 10, 0] += 2             # perturb one cell
    with Model(case["cov"], case["init"], max_chains=3) as model:
        lp = model.log_prob(u, ev)
    for b in range(3):
        want = so.joint_log_prob(u[b], ev[b], case["k"], "stable")
        if np.isfinite(want):
            assert abs(lp[b] - want) <= RTOL_LOGP * abs(want)
        else:
            assert lp[b] == want or (np.isnan(lp[b]) and np.isnan(want))


@pytest.mark.parametrize("form", ["fused", "four-launch"])
def test_large_hazards_take_the_libm_branch(Model, form):
    """Rates beyond the small-rate series of log(1 - exp(-r)) (r > 1/8): in the fused form those cells are left out
    of the branch-free pass and redone by the cold pass of the tile epilogue; some waves have them, some do not."""
    case = H.build_case("ni11", 21, alpha_t_sd=0.005)
    k = case["k"]
    u, ev = _batch(case, 4, 21)
    u[:, 5] += np.array([5.5, 6.5, 7.5, 0.0])               # alpha_0: hazards x245, x665, x1800, x1
    with Model(case["cov"], case["init"], max_chains=4) as model:
        model.set_option(eval_form=form)
        lp = _check(case, model, u, ev, grad=True)
        _check(case, model, u, ev, grad=False)
    assert np.isfinite(lp).all()
    # the point of the test: the shifted chains have cells beyond the series' range, the last one has none
    state = so.compute_state(case["init"], ev[0])
    over = []
    for b in range(4):
        lam, _, _ = so.transition_rates(so.unpack(so.constrain(u[b]), k.M, k.T), k, state)
        over.append(int((lam > 0.125).sum()))
    assert over[2] > 10 and over[3] == 0, over


def test_infeasible_events_are_minus_inf(Model):
    case = H.build_case("micro_2x3", 9)
    ev = case["events"].copy()
    ev[0, 1, 1] = 1e6
    with Model(case["cov"], case["init"]) as model:
        assert model.log_prob(case["u"], ev) == -np.inf


def test_negative_rate_semantics(Model):
    # lam<0 with k_se=0 stays finite (multiply_no_nan), with k_se>0 is NaN -> MH rejects
    from covid19uk_amd import model_spec as ms
    cov = ms.Covariates(C=np.array([[0.0, 50.0], [50.0, 0.0]]), W=np.ones(1), N=np.array([100.0, 100.0]),
                        adjacency=np.array([[0.0, 1.0], [1.0, 0.0]]), weekday=np.ones(1),
                        area=np.array([1e8, 1e8]))
    init = np.array([[90.0, 0.0, 10.0, 0.0], [100.0, 0.0, 0.0, 0.0]])
    k = H.oracle_constants(cov, init)
    theta = np.array([5.0, 0.1, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0])
    u = synth.unconstrain(theta)
    ev = np.zeros((2, 1, 3))
    with Model(cov, init) as model:
        got = model.log_prob(u, ev)
        want = so.joint_log_prob(u, ev, k)
        assert np.isfinite(want) and abs(got - want) <= RTOL_LOGP * abs(want)
        ev[0, 0, 0] = 1.0
        assert np.isnan(model.log_prob(u, ev))


@pytest.mark.parametrize("form", ["fused", "three-launch", "four-launch"])
def test_uk380_batch_matches_c_oracle(Model, form):
    case = H.build_case("uk380", 10, alpha_t_sd=0.005)
    u, ev = _batch(case, 8, 10)
    with Model(case["cov"], case["init"], max_chains=8) as model:
        model.set_option(eval_form=form)
        _check(case, model, u, ev, grad=True, use_c=True)
        _check(case, model, u, ev, grad=False, use_c=True)


@pytest.mark.parametrize("name,seed,B", [("micro_2x3", 2, 3), ("micro_17x70", 4, 2), ("ni11", 5, 16)])
def test_four_launch_form_matches_numpy_oracle(Model, name, seed, B):
    """The default (fused) form is what every other test exercises; the four-launch form stays as the cross-check."""
    case = H.build_case(name, seed, alpha_t_sd=0.005)
    u, ev = _batch(case, B, seed)
    with Model(case["cov"], case["init"], max_chains=B) as model:
        model.set_option(eval_form="four-launch")
        _check(case, model, u, ev, grad=True)


def test_prepared_path_and_linearity_properties_at_full_size(Model):
    """Size-independent properties at the BASELINE size (UK-380 x 365, 8 chains):
    (1) prepare+eval == one-shot; (2) chains are independent: permuting the batch
    permutes the outputs bit-for-bit; (3) the log-prob of relabelled LADs is invariant."""
    import torch
    case = H.build_case("uk380", 11)
    B = 8
    u, ev = _batch(case, B, 11)
    dev = torch.device("cuda:0")
    ut, evt = torch.tensor(u, device=dev), torch.tensor(ev, device=dev)
    lp1 = torch.empty(B, dtype=torch.float64, device=dev)
    lp2 = torch.empty_like(lp1)
    g1 = torch.empty(B, u.shape[1], dtype=torch.float64, device=dev)
    g2 = torch.empty_like(g1)
    with Model(case["cov"], case["init"], max_chains=B) as model:
        model.set_option(eval_form="four-launch")          # the same kernels as prepare + eval: same bits
        model.log_prob_dev(ut, evt, lp1, g1)
        model.sync()
        model.prepare_events_dev(evt)
        model.eval_prepared_dev(ut, lp2, g2)
        model.sync()
        assert torch.equal(lp1, lp2) and torch.equal(g1, g2)
        model.set_option(eval_form="fused")                # default: other tiles, other summation order
        model.log_prob_dev(ut, evt, lp1, g1)
        model.sync()
        assert torch.allclose(lp1, lp2, rtol=1e-13, atol=0.0)
        gs = torch.maximum(g2.abs(), 1e-6 * g2.abs().amax(dim=1, keepdim=True))
        assert float(((g1 - g2).abs() / gs).max()) < 1e-9
        model.eval_prepared_dev(ut, lp2, g2)               # F and the tables left by the fused form serve the prepared path
        model.sync()
        assert torch.allclose(lp1, lp2, rtol=1e-13, atol=0.0)
        perm = torch.tensor([3, 1, 7, 0, 2, 6, 5, 4], device=dev)
        model.log_prob_dev(ut[perm].contiguous(), evt[perm].contiguous(), lp2, g2)
        model.sync()
        assert torch.equal(lp1[perm], lp2) and torch.equal(g1[perm], g2)
    # relabelling invariance through a second context
    cov, k = case["cov"], case["k"]
    rp = np.random.default_rng(0).permutation(k.M)
    from covid19uk_amd import model_spec as ms
    cov2 = ms.Covariates(C=cov.C[np.ix_(rp, rp)], W=cov.W, N=cov.N[rp], adjacency=cov.adjacency[np.ix_(rp, rp)],
                         weekday=cov.weekday, area=cov.area[rp])
    u2 = u[:1].copy()
    u2[0, 6 + k.T - 1:] = u[0, 6 + k.T - 1:][rp]
    with Model(cov2, case["init"][rp]) as model2:
        b = model2.log_prob(u2, ev[:1, rp])
    a = float(lp1[0].cpu())
    assert abs(a - b[0]) <= 1e-11 * abs(a)


def test_argument_errors(Model):
    from covid19uk_amd import _lib
    case = H.build_case("micro_2x3", 12)
    with Model(case["cov"], case["init"], max_chains=1) as model:
        with pytest.raises(ValueError):
            model.log_prob(case["u"][:-1], case["events"])
        with pytest.raises(_lib.SeirError):
            model.log_prob(np.stack([case["u"]] * 2), np.stack([case["events"]] * 2))   # B > max_chains


def test_device_math_against_mpmath(Model):
    """The table log / Newton reciprocal / series behind every cell: 1e-15 relative against
    50-digit references over the rates and counts the model produces."""
    import mpmath as mp
    mp.mp.dps = 40
    case = H.build_case("micro_2x3", 13)
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-9), np.log(5.0), 4000)),
                        [1e-9, 0.125, 0.1250001, 1.0, 0.28, 63.5, 64.0, 65.2, 1e3, 1.2e6]])
    with Model(case["cov"], case["init"]) as model:
        L, inv, lf = model.selftest_math(x)
    for i in range(0, len(x), 7):
        xi = mp.mpf(float(x[i]))
        Lt, it = mp.log(1 - mp.e ** (-xi)), 1 / (mp.e ** xi - 1)
        assert abs((mp.mpf(float(L[i])) - Lt) / Lt) < 2e-15 or abs(mp.mpf(float(L[i])) - Lt) < 2e-16, (x[i], L[i])
        assert abs((mp.mpf(float(inv[i])) - it) / it) < 2e-15, (x[i], inv[i])
        lt = mp.loggamma(mp.floor(xi) + 1)
        assert abs(mp.mpf(float(lf[i])) - lt) <= 2e-15 * max(1, abs(lt)), (x[i], lf[i])
    # the 8-term variant of the I->R terms (series up to 3/4, libm beyond)
    xw = np.concatenate([np.linspace(1e-6, 0.75, 400), [0.125, 0.25, 0.2857, 0.75, 0.7500001, 2.0]])
    with Model(case["cov"], case["init"]) as model:
        Lw, iw = model.selftest_math_wide(xw)
    for i in range(len(xw)):
        xi = mp.mpf(float(xw[i]))
        Lt, it = mp.log(1 - mp.e ** (-xi)), 1 / (mp.e ** xi - 1)
        assert abs((mp.mpf(float(Lw[i])) - Lt) / Lt) < 2e-15 or abs(mp.mpf(float(Lw[i])) - Lt) < 2e-16, (xw[i], Lw[i])
        assert abs((mp.mpf(float(iw[i])) - it) / it) < 2e-15, (xw[i], iw[i])


def test_evaluation_does_not_depend_on_workgroup_timing(Model):
    """The debug-skew option starts a third of the workgroups of every launch ~30 us late: the stateless
    evaluation (scan, partial folds, MFMA contraction, S->E tiles) must give the same bits."""
    case = H.build_case("uk380", 15)
    u, ev = _batch(case, 3, 15)
    ref = None
    for skew in (0, 1, 2, 3):
        with Model(case["cov"], case["init"], max_chains=3) as model:
            model.set_option(debug_skew=skew)
            got = model.log_prob_grad(u, ev)
        if ref is None:
            ref = got
        else:
            assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]), skew


def test_syn2048_matches_c_oracle(Model):
    """BASELINE config 5 shape (2048 regions x 730 days, dense mobility): the K-chunked fp64 MFMA
    contraction, the multi-chunk scans and the 2048-row reductions against the C oracle."""
    from oracle import c_binding
    case = H.build_case("syn2048", 14)
    c_binding.set_threads(8)
    u, ev = _batch(case, 2, 14)
    u[:, :6] = case["u"][:6] + 0.01 * np.random.default_rng(14).normal(size=(2, 6))
    with Model(case["cov"], case["init"], max_chains=2) as model:
        lp, g = model.log_prob_grad(u, ev)
    for b in range(2):
        want, gw = c_binding.evaluate(case["k"], u[b], ev[b], 1, want_grad=True)
        assert abs(lp[b] - want) <= RTOL_LOGP * abs(want), (lp[b], want)
        scale = np.maximum(np.abs(gw), 1e-6 * np.abs(gw).max())
        assert np.max(np.abs(g[b] - gw) / scale) < RTOL_GRAD


def test_fp32_mfma_contraction_at_syn2048(Model):
    """BASELINE config 5 ("fp32 MFMA mobility matvec", 2048 regions x 730 days): the contraction F = Cstar . I/N
    with fp32 operands on v_mfma_f32_32x32x2_f32 (option gemm_f32) against the C oracle.  fp32 products and
    accumulation put ~1e-7 relative error into F; what reaches the log-prob was measured at 3e-10 .. 1.2e-9,
    the gradient at 2e-7 of its largest component -- stated here as 1e-8 / 2e-6, outside the 1e-9 / 1e-6
    of the fp64 path, which is why the option is off by default."""
    from oracle import c_binding
    case = H.build_case("syn2048", 14)
    c_binding.set_threads(8)
    u, ev = _batch(case, 2, 14)
    u[:, :6] = case["u"][:6] + 0.01 * np.random.default_rng(14).normal(size=(2, 6))
    with Model(case["cov"], case["init"], max_chains=2) as model:
        exact = model.log_prob_grad(u, ev)
        model.set_option(gemm_f32=True)
        lp, g = model.log_prob_grad(u, ev)
        t32 = model.time_kernel("gemm", 2, 5)
        model.set_option(gemm_f32=False)
        again = model.log_prob_grad(u, ev)
        t64 = model.time_kernel("gemm", 2, 5)
    assert np.array_equal(exact[0], again[0]) and np.array_equal(exact[1], again[1])     # the option switches back cleanly
    assert not np.array_equal(lp, exact[0])                                            # ... and it did switch
    for b in range(2):
        want, gw = c_binding.evaluate(case["k"], u[b], ev[b], 1, want_grad=True)
        assert abs(lp[b] - want) <= 1e-8 * abs(want), (lp[b], want)
        assert np.max(np.abs(g[b] - gw)) < 2e-6 * np.abs(gw).max()
    assert t32 < 0.8 * t64, (t32, t64)     # 2 chains: 192 workgroups; at 8 chains 461 us against 986 us


def test_fp32_contraction_needs_128_tiles(Model):
    from covid19uk_amd import _lib
    case = H.build_case("ni11", 3)
    with Model(case["cov"], case["init"], max_chains=1) as model:
        with pytest.raises(_lib.SeirError):
            model.set_option(gemm_f32=True)          # ceil64(11) = 64 is not a multiple of 128
        lp = model.log_prob(case["u"], case["events"])
        assert abs(lp - so.joint_log_prob(case["u"], case["events"], case["k"], "stable")) <= RTOL_LOGP * abs(lp)


@pytest.mark.parametrize("name,seed", [("ni11", 23), ("uk380", 24)])
@pytest.mark.parametrize("B", [8, 16])
def test_one_launch_and_three_launch_evaluations_give_the_same_bits(Model, name, seed, B):
    """With a multiple of 8 chains whose tile workgroups all fit the chip (UK-380: 8 and 16; chain b on XCD b mod 8) the
    default form runs state scan, contraction tiles and reduction as ONE launch (k_eval_all:
    counters in the chain's own L2 line, consumers behind producers in block order) where the GPU places block ids
    congruent mod 8 on one XCD each; "three-launch" forces the separate launches.  Same code, same order: same bits,
    call after call (the counters run on), with workgroup skew, and for value-only calls."""
    case = H.build_case(name, seed, alpha_t_sd=0.005)
    u, ev = _batch(case, B, seed)
    res = {}
    for form, skew in (("three-launch", 0), ("fused", 0), ("fused", 2)):
        with Model(case["cov"], case["init"], max_chains=B) as model:
            model.set_option(eval_form=form, debug_skew=skew)
            out = []
            for rep in range(3):
                lp, g = model.log_prob_grad(u, ev)
                out.append((lp.copy(), g.copy(), model.log_prob(u, ev).copy()))
            res[(form, skew)] = out
    ref = res[("three-launch", 0)]
    for key in (("fused", 0), ("fused", 2)):
        for (a, b, c), (x, y, z) in zip(ref, res[key]):
            assert np.array_equal(a, x) and np.array_equal(b, y) and np.array_equal(c, z), key


def test_one_context_serves_batches_of_different_sizes(Model):
    """The one-launch form keeps a counter pair per chain and one running target per context: after a batch of 8 the
    chains 8..15 of a batch of 16 must not be left one launch behind the target (they would time out and evaluate
    without their producers' data).  8, 16, 8, 16 and a partial batch on ONE context, each against a context of its
    own size, bit for bit, and no hand-off time-out."""
    case = H.build_case("uk380", 25, alpha_t_sd=0.005)
    u, ev = _batch(case, 16, 25)
    ref = {}
    for B in (8, 16, 5):
        with Model(case["cov"], case["init"], max_chains=B) as model:
            ref[B] = model.log_prob_grad(u[:B], ev[:B])
    with Model(case["cov"], case["init"], max_chains=16) as model:
        for B in (8, 16, 8, 5, 16, 16):
            lp, g = model.log_prob_grad(u[:B], ev[:B])        # raises on a hand-off time-out
            assert np.array_equal(lp, ref[B][0]) and np.array_equal(g, ref[B][1]), B
            assert np.array_equal(model.log_prob(u[:B], ev[:B]), ref[B][0]), B
