"""-m gpu: the BENCHED configurations at BASELINE.json's full size against committed oracle digests
(tests/golden/fullsize_digests.json, made offline by tests/golden/make_golden_fullsize.py with the
CPU oracle) — frame by frame, without running the oracle on the GPU box:

  bench_5mm         bench.py's default workload with its exact table sizes, frames 0..24
                    (= the driver's `--steps 20 --warmup 5` run)
  cfg5_4mm_gc_swap  the `4mm` preset with voxel GC + host swapping (BASELINE configs[4])
  cfg2_instances    static map + 4 instance volumes, masks split on the device (configs[2])

Bit-exact: every digest is a SHA-256 over the raw arrays (hash table, visible list, voxels,
range image, raycast points, ICP maps, swap state, host store)."""
import json
import os

import pytest

from tests.golden.make_golden_fullsize import CASES, case_frames, run_case

pytestmark = pytest.mark.gpu
_PATH = os.path.join(os.path.dirname(__file__), "golden", "fullsize_digests.json")
GOLD = json.load(open(_PATH))["cases"] if os.path.exists(_PATH) else {}


def _diff(got, want, path=""):
    out = []
    if isinstance(want, dict):
        for k in want:
            out += _diff(got.get(k) if isinstance(got, dict) else None, want[k], f"{path}/{k}")
    elif isinstance(want, list):
        for i, w in enumerate(want):
            out += _diff(got[i] if isinstance(got, list) and i < len(got) else None, w, f"{path}[{i}]")
    elif got != want:
        out.append(path)
    return out


@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_fullsize_golden(hip_api, name):
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    assert name in GOLD, f"{name} missing from fullsize_digests.json: run tests/golden/make_golden_fullsize.py"

    def hip_factory(settings, calib_args):
        return EngineCore(default_settings(**settings), make_calib(*calib_args))
    got = run_case(hip_factory, CASES[name], case_frames(CASES[name]))
    bad = _diff(got, GOLD[name])
    assert not bad, f"{len(bad)} digests differ from the oracle's, first: {bad[:8]}"
