"""The ray march of k_raycast (dynslam_amd/csrc/k_raycast.h cast_ray) against the oracle — on the CPU.

cast_ray is written against a small `Ops` policy (float -> int conversion, "any ray of the wave"), so the very function the
kernel runs can be compiled for the host with a one-ray Ops (tests/hostsim/raycast_host.hip) and driven over the oracle's
table, voxels and range image: the lookup rounds, the look-ahead slot, the block map (with maps small enough to be mostly
conflicted, too) and the trilinear reads must reproduce the oracle's raycast bit for bit.  Here, without a GPU; the same
comparison runs on the device in the -m gpu suite.
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.common import SMALL

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "raycast_host.hip")
LIB = os.path.join(HERE, "hostsim", "_build", "libraycast_host.so")
CSRC = os.path.join(os.path.dirname(HERE), "dynslam_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _lib():
    deps = [SRC] + [os.path.join(CSRC, h) for h in ("k_raycast.h", "dsr_device.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
            pytest.skip("hipcc not available to build the host stand-in")
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-fno-fast-math", "-Wno-unused-function", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    lib.rr_cast_all.restype = C.c_int
    return lib


def _oracle_scene(n_frames, **kw):
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import OracleEngine, oracle_settings
    W, H = 320, 96
    settings = dict(SMALL)
    settings.update(kw)
    sc = StreetScene(W, H)
    o = OracleEngine(oracle_settings(**settings), make_calib(*sc.intrinsics(), W, H))
    for i in range(n_frames):
        rgba, d, T, _ = sc.frame(i)
        o.update_view(rgba, d)
        o.set_pose_inv_m(T)
        o.process_frame()
        o.prepare()
    return sc, o, settings


def _cast(lib, sc, o, settings, occ_entries):
    rs = o.dump_render_state()
    table = o.dump_hash_table()
    vox = o.dump_voxel_blocks()
    vba = np.zeros((o.no_blocks, 4096), np.uint8)  # the library's block layout: the sdf plane comes first
    vba[:, :1024] = np.ascontiguousarray(vox["sdf"]).view(np.uint8).reshape(o.no_blocks, 1024)
    _, inv_m = o.get_pose()
    inv_m = np.ascontiguousarray(inv_m.T.astype(np.float32)).ravel()  # column-major, as the engine holds it
    proj = np.array(sc.intrinsics(), np.float32)
    out = np.zeros((o.H, o.W, 4), np.float32)
    stats = np.zeros(2, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.rr_cast_all(p(inv_m), p(proj), C.c_float(settings["voxel_size"]), C.c_float(settings["mu"]), o.W, o.H,
                         settings["hash_bucket_num"], o.no_total_entries, p(table), p(vba), p(np.ascontiguousarray(rs["minmax"])),
                         occ_entries, p(out), p(stats))
    assert rc == 0
    return out, rs["raycast_result"], stats


@pytest.mark.parametrize("occ_entries", [0, 1 << 17, 2048, 64])
def test_march_equals_oracle(occ_entries):
    lib = _lib()
    sc, o, settings = _oracle_scene(4)
    got, want, stats = _cast(lib, sc, o, settings, occ_entries)
    assert (want[..., 3] > 0).sum() > 0.3 * want[..., 3].size, "the scene must be hit by a good part of the rays"
    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        bad = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1))
        raise AssertionError(f"{len(bad)} rays differ, first {bad[0]}: {got[tuple(bad[0])]} vs {want[tuple(bad[0])]}")
    if occ_entries == 64:
        assert stats[1] > 32, "the tiny map must be mostly conflicted (its rays ask the table)"
    if occ_entries == 1 << 17:
        assert stats[1] < 0.1 * stats[0]
    o.close()


def test_march_with_long_chains():
    """256 buckets for thousands of blocks: nearly every lookup walks the excess list."""
    lib = _lib()
    sc, o, settings = _oracle_scene(3, hash_bucket_num=256, excess_list_size=0x8000)
    for occ_entries in (0, 1 << 16):
        got, want, _ = _cast(lib, sc, o, settings, occ_entries)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    o.close()
