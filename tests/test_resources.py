"""The compiler's account of the kernels (`hipcc -Rpass-analysis=kernel-resource-usage`, kept by `__graft_entry__.build()`
as covid19uk_amd/kernel_resources.json; a round's copy is committed as profiles/rNN_kernel_resources.json): the instances
that a BASELINE.json configuration launches on its hot path must not acquire scratch -- a spill inside a persistent launch's
step loop is paid in every step -- and must keep the occupancy their launches' residency arithmetic relies on."""
import json
import os

import pytest

import __graft_entry__ as entry

# instance -> (max VGPRs, min waves per SIMD) the host's launch logic assumes; scratch must be 0 for all of them
HOT_PATH = {
    # UK-380 x 8 / x 1 chains (BASELINE configs 3, 4): the two persistent launches of a sweep
    "k_leap<1,6,1,6>": (128, 4),       # 24-row tiles: 96 + 12 workgroups per XCD need four waves per SIMD
    "k_leap<1,6,2,4>": (168, 3),       # 32-row tiles (the fall-back shape)
    "k_move_pairs<6>": (256, 2),
    # NI-11 x 16 chains (config 2)
    "k_leap<1,1,2,4>": (168, 3),
    # the per-step forms (16+ chains per GPU, a shared GPU, SYN-2048: config 5)
    "k_se_chunk<1,6>": (128, 4), "k_se_chunk<1,1>": (128, 4), "k_se_chunk<2,12>": (168, 3),
    "k_move_pair<6>": (168, 3), "k_move_pair<12>": (168, 3), "k_move_delta<false>": (128, 4),
    "k_hmc_step<0,1,1>": (256, 2), "k_hmc_step<2,1,1>": (256, 2), "k_hmc_step<0,2,4>": (256, 2), "k_hmc_step<2,2,4>": (256, 2),
    "k_se<true,1,0>": (128, 4), "k_se<true,1,1>": (128, 4), "k_se<true,1,2>": (128, 4), "k_record": (128, 4),
    "k_gemm_f32": (168, 3),
    # the stateless evaluation (seir_log_prob_dev) at UK-380 / SYN-2048 (96-day tiles) and NI-11 (64)
    "k_eval_all<true,96>": (128, 4), "k_eval_all<false,96>": (128, 4), "k_eval_tiles<true,96>": (128, 4),
    "k_eval_tiles<false,96>": (128, 4), "k_state_params": (128, 4), "k_finish<true>": (128, 4), "k_finish<false>": (128, 4),
}


@pytest.fixture(scope="module")
def resources():
    entry.build()
    assert os.path.exists(entry.RESOURCES), "build() keeps the compiler's resource remarks next to the library"
    return json.load(open(entry.RESOURCES))


def test_every_hot_path_instance_is_compiled(resources):
    missing = [k for k in HOT_PATH if k not in resources]
    assert not missing, missing


@pytest.mark.parametrize("kernel", sorted(HOT_PATH))
def test_hot_path_instances_have_no_scratch_and_keep_their_occupancy(resources, kernel):
    r = resources[kernel]
    vmax, occ = HOT_PATH[kernel]
    assert r["scratch_bytes_per_lane"] == 0 and r["vgpr_spill"] == 0, (kernel, r)
    assert r["vgpr"] <= vmax and r["occupancy_waves_per_simd"] >= occ, (kernel, r)


def test_the_committed_copy_is_this_rounds(resources):
    """profiles/rNN_kernel_resources.json is the tracked copy the docs cite: it must list the same kernels, and agree with a
    fresh build on what the hot path is held to (scratch) -- register counts may move by a few with the compiler's mood."""
    import glob
    files = sorted(glob.glob(os.path.join(entry.ROOT, "profiles", "r*_kernel_resources.json")))
    assert files, "no committed profiles/rNN_kernel_resources.json"
    doc = json.load(open(files[-1]))
    assert set(doc) == set(resources)
    for k in HOT_PATH:
        assert doc[k]["scratch_bytes_per_lane"] == resources[k]["scratch_bytes_per_lane"], k
