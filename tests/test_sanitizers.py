"""Memory-safety of the host-side C++ that sits above the C ABI — shim/ITMLib.h (header-only ITMLib facade), our two hosts
(shim/host_bench.cpp incl. the instance mode, shim/example_host.cpp) — and of the CPU oracle underneath them, under
AddressSanitizer + UndefinedBehaviorSanitizer with leak detection.  The hosts are linked against a sanitised build of the
oracle (dsr_* renamed to orc_* at compile time, as in tests/test_reference_compiles.py); nothing here needs a GPU."""
import os
import subprocess

import pytest

from dynslam_amd import _capi
from tests.test_reference_compiles import _instances_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.fixture(scope="module")
def sanitised(tmp_path_factory):
    d = tmp_path_factory.mktemp("asan")
    probe = subprocess.run(["g++", "-fsanitize=address,undefined", "-x", "c++", "-", "-o", str(d / "probe")], input="int main(){return 0;}",
                           capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("this toolchain has no sanitizer runtime")
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared"] + SAN +
                          ["-o", str(d / "liboracle.so"), os.path.join(ROOT, "oracle", "dsr_oracle.cpp")])
    rename = d / "dsr_to_orc.h"
    rename.write_text("".join(f"#define dsr_{name} orc_{name}\n" for name in _capi.SIGNATURES))
    exes = {}
    for name in ("host_bench", "example_host"):
        exes[name] = str(d / name)
        subprocess.check_call(["g++", "-std=c++17"] + SAN + ["-include", str(rename), "-I", os.path.join(ROOT, "shim"),
                                                             os.path.join(ROOT, "shim", name + ".cpp"), "-o", exes[name], "-L", str(d), "-loracle",
                                                             f"-Wl,-rpath,{d}", "-fopenmp"])
    exes["dir"] = str(d)
    return exes


def _run_clean(args, asan="detect_leaks=1:halt_on_error=1", cwd=None):
    env = dict(os.environ, ASAN_OPTIONS=asan, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="4")
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=900, cwd=cwd)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout


def test_shim_hosts_are_clean_under_asan_and_ubsan(sanitised, tmp_path):
    W, H, n, n_inst = 256, 80, 4, 2
    sc, frames, path, mpath, _ = _instances_case(tmp_path, W, H, n, n_inst)
    fx, fy, cx, cy = sc.intrinsics()
    base = [sanitised["host_bench"], str(path), str(W), str(H), str(fx), str(fy), str(cx), str(cy), str(n), "1", "0.05", "0.2", "40000", "65536", "16384"]
    assert "frames=4" in _run_clean(base)                                    # static map
    assert "saved_bytes=" in _run_clean(base + ["2", "1"])                    # + voxel GC (max weight 2, min age 1)
    assert "instances=2" in _run_clean(base + ["--masks", str(mpath), str(n_inst)])  # + instance volumes, view split
    assert "hash=" in _run_clean([sanitised["example_host"]])


def test_reference_pipeline_is_clean_under_asan_and_ubsan(sanitised, tmp_path):
    """The reference's whole per-frame pipeline (tests/test_reference_pipeline.py) on the sanitised oracle: the reference's
    host keeps views, engines and tracks alive in its own order (a track's views outlive its engine, engines are created
    mid-sequence, reaped, destroyed) — the shim must stay memory-safe under exactly that usage.  The reference's own
    new[] / delete mismatches and leaks (VelodyneIO.h:52, Track pointers) are not ours to fix: those two checks are off."""
    from tests.refhost import build_pipeline as bp
    from tests.refhost.make_dataset import write_dataset
    if not bp.have_reference():
        pytest.skip("/root/reference is not on this machine")
    exe = bp.build("oracle", str(tmp_path / "ref_dynslam_host_asan"), str(tmp_path / "obj"), extra_flags=SAN, oracle_dir=sanitised["dir"])
    root = tmp_path / "kitti_like"
    root.mkdir()
    write_dataset(str(root), 5, 1242, 375)
    out = _run_clean([exe, str(root), "5", str(tmp_path / "out.bin"), "0.05", "1", "1"],
                     asan="detect_leaks=0:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0:halt_on_error=1", cwd=str(root))
    assert "frames=5" in out and "pruned_view_detached=1" in out
