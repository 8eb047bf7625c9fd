"""Committed golden digests (tests/golden/state_digests.json, made by make_golden.py):
the oracle must keep reproducing them (CPU), the HIP engine must match them (-m gpu)."""
import json
import os

import pytest

from tests.golden.make_golden import CASES, oracle_factory, run_case

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_digests.json")))["cases"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(oracle_lib, name):
    assert run_case(oracle_factory, CASES[name]) == GOLD[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_golden(hip_api, name):
    from dynslam_amd.engine import EngineCore, default_settings, make_calib

    def hip_factory(settings, calib_args):
        return EngineCore(default_settings(**settings), make_calib(*calib_args))
    got = run_case(hip_factory, CASES[name])
    diff = {k: (got[k], GOLD[name][k]) for k in got if got[k] != GOLD[name][k]}
    assert not diff, f"fields differing from the golden fixture: {sorted(diff)}"


def test_fullsize_fixture_covers_every_benched_case():
    """tests/golden/fullsize_digests.json (made offline by make_golden_fullsize.py; replayed on the GPU by
    tests/test_gpu_fullsize_golden.py) holds one per-frame record for every frame of every case."""
    from tests.golden.make_golden_fullsize import CASES as FULL
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize_digests.json")))["cases"]
    for name, case in FULL.items():
        assert name in doc and len(doc[name]["frames"]) == case["frames"]
        last = doc[name]["frames"][-1]
        assert "voxels_in_use" in last and "raycast_result" in last and len(last["hash_table"]) == 64
    assert doc["cfg5_4mm_gc_swap"]["frames"][-1]["decayed_block_count"] > 0
    assert doc["cfg5_4mm_gc_swap"]["frames"][-1]["swap_stored_count"] > 0
    assert len(doc["cfg2_instances"]["frames"][-1]["instances"]) == 4
    assert doc["seq06_5cm_50frames"]["frames"][-1]["decayed_block_count"] > 0
