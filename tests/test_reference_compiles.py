"""The drop-in claim, checked against the reference's OWN host code.

CPU (here, where /root/reference exists):
  * the unmodified src/DynSLAM/InfiniTamDriver.cpp (which pulls InfiniTamDriver.h, Input.h, Utils.h,
    DepthProvider.h, Defines.h, PreviewType.h, VoxelDecayParams.h), InstRecLib/InstanceReconstructor.cpp (which
    pulls DynSlam.h, InstanceTracker.h, Track.h, InstanceView.h, ...), InstanceTracker.cpp, Track.cpp, InstanceView.cpp
    and the host units around them (REFERENCE_UNITS: 24 translation units) compile against shim/ITMLib.h through
    the forwarding headers under shim/InfiniTAM/ — the only other headers are the functional stand-ins for
    OpenCV / Eigen / Pangolin / gflags / libviso2 under tests/stubs/ (none is installed or vendored);
  * they LINK with shim/host_bench.cpp (-DDSR_HOST_REFERENCE_DRIVER) and libdsr_hip.so into
    tests/refhost/_build/ref_driver_host: every ITMLib symbol the reference's driver needs resolves.
GPU (the prebuilt binary travels with the snapshot; /root/reference does not exist there):
  * the reference's `dynslam::drivers::InfiniTamDriver` — UpdateView(cv::Mat3b, cv::Mat1s), SetPose(Eigen),
    Integrate, PrepareNextStep, Decay, GetImage / GetFloatImage(pangolin::OpenGlMatrix) — runs on the HIP
    engine and produces bit for bit what the Python mirror and our own C++ HostDriver produce.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/DynSLAM"
LIB_DIR = os.path.join(ROOT, "dynslam_amd", "csrc")
REF_EXE = os.path.join(ROOT, "tests", "refhost", "_build", "ref_driver_host")
SHIM_EXE = os.path.join(ROOT, "shim", "host_bench")
LINK = ["-L", LIB_DIR, "-ldsr_hip", f"-Wl,-rpath,{LIB_DIR}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
REF_INC = ["-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "tests", "stubs", "DynSLAM"),
           "-I", os.path.join(ROOT, "tests", "stubs", "DynSLAM", "InstRecLib"),
           "-I", os.path.join(ROOT, "shim", "DynSLAM"), "-I", os.path.join(ROOT, "shim", "DynSLAM", "InstRecLib"),
           "-I", REF, "-I", os.path.join(REF, "InstRecLib")]
# every translation unit of the reference that reaches the engines (ITMLib names) — plus the host-side units around
# them that need nothing beyond the stand-in third-party headers.  Not in the list: DynSLAMGUI.cpp, DSHandler3D.cpp and
# Evaluation/ErrorVisualizationCallback.cpp (Pangolin GUI / OpenGL), the Direct/ image-alignment library (unused:
# InstanceReconstructor.cpp:590-606 throws "Deprecated").
REFERENCE_UNITS = [
    "InfiniTamDriver.cpp", "Utils.cpp",
    "InstRecLib/InstanceReconstructor.cpp",  # ITMView, SetView, GetScene, ITMMeshingEngine / ITMMesh, GetImage per instance
    "InstRecLib/InstanceTracker.cpp", "InstRecLib/Track.cpp", "InstRecLib/InstanceView.cpp",
    "InstRecLib/InstanceSegmentationResult.cpp", "InstRecLib/SegmentationDataset.cpp", "InstRecLib/SparseSFProvider.cpp",
    "InstRecLib/Utils/BoundingBox.cpp", "Evaluation/CsvWriter.cpp", "Evaluation/Tracklets.cpp",
    "PrecomputedDepthProvider.cpp",  # cv::FileStorage / pfmLib ReadFilePFM stand-ins call dsr_read_depth_xml / dsr_read_pfm
    "DynSlam.cpp",  # the per-frame orchestrator: ITMSafeCall(cudaDeviceSynchronize()) -> dsr_device_synchronize
    "Input.cpp", "InstRecLib/Utils/Mask.cpp", "InstRecLib/PrecomputedSegmentationProvider.cpp",
    "InstRecLib/VisoSparseSFProvider.cpp",
    "Evaluation/Evaluation.cpp", "Evaluation/VelodyneIO.cpp", "Evaluation/EvaluationCallback.cpp",  # readers of the a14 depth renders
    "Evaluation/SegmentedCallback.cpp", "Evaluation/SegmentedEvaluationCallback.cpp", "Evaluation/SegmentedVisualizationCallback.cpp",
]

have_ref = os.path.isdir(REF)


def build_shim_host():
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "shim"),
                           os.path.join(ROOT, "shim", "host_bench.cpp"), "-o", SHIM_EXE] + LINK)
    return SHIM_EXE


def build_ref_host():
    os.makedirs(os.path.dirname(REF_EXE), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-DNDEBUG", "-DDSR_HOST_REFERENCE_DRIVER"] + REF_INC +
                          [os.path.join(ROOT, "shim", "host_bench.cpp"), os.path.join(REF, "InfiniTamDriver.cpp"),
                           os.path.join(REF, "Utils.cpp"), "-o", REF_EXE] + LINK)
    return REF_EXE


@pytest.mark.skipif(not have_ref, reason="/root/reference is not on this machine")
@pytest.mark.parametrize("src", REFERENCE_UNITS)
def test_reference_sources_compile_unmodified_against_the_shim(src):
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wno-unused-variable"] + REF_INC + [os.path.join(REF, src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.skipif(not have_ref, reason="/root/reference is not on this machine")
def test_reference_driver_links_with_the_hip_library():
    exe = build_ref_host()
    und = subprocess.check_output(["nm", "-u", exe]).decode()
    used = {l.split()[-1] for l in und.splitlines() if "dsr_" in l}
    # what the reference's InfiniTamDriver reaches through the shim
    for sym in ("dsr_engine_create", "dsr_update_view", "dsr_set_pose_inv_m", "dsr_process_frame", "dsr_prepare",
                "dsr_decay", "dsr_get_image", "dsr_get_stats", "dsr_set_fusion_weight_params"):
        assert sym in used, sym
    defined = subprocess.check_output(["nm", "-C", "--defined-only", exe]).decode()
    assert "dynslam::drivers::InfiniTamDriver::UpdateView" in defined  # the reference's own translation unit is in there
    assert "dynslam::drivers::CreateItmCalib" in defined


def test_shim_host_bench_builds():
    assert os.path.exists(build_shim_host())


def write_frames_file(path, sc, n):
    """frames.bin of shim/host_bench.cpp: BGR u8, depth int16 mm, pose float[16] row-major per frame + render pose."""
    with open(path, "wb") as f:
        for i in range(n):
            rgba, d, T, _ = sc.frame(i)
            f.write(np.ascontiguousarray(rgba[..., 2::-1]).tobytes())
            f.write(np.ascontiguousarray(d, np.int16).tobytes())
            f.write(np.ascontiguousarray(T, np.float32).tobytes())
        M = np.linalg.inv(sc.pose(n - 1).astype(np.float64)).astype(np.float32)
        f.write(M.tobytes())
    return M


def fnv(data, h=1469598103934665603):
    a = np.frombuffer(bytes(data), np.uint8)
    for b in a.tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


KW = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
          sdf_local_block_num=20000, hash_bucket_num=0x8000, excess_list_size=0x2000)


def _host_args(path, sc, n, W, H):
    fx, fy, cx, cy = sc.intrinsics()
    return [str(path), str(W), str(H), repr(fx), repr(fy), repr(cx), repr(cy), str(n), "1", repr(KW["voxel_size"]), repr(KW["mu"]),
            str(KW["sdf_local_block_num"]), str(KW["hash_bucket_num"]), str(KW["excess_list_size"]), "1", "1"]


def _mirror_digest(make_engine, api, sc, n, W, H, M):
    """The host_bench call sequence through the Python mirror -> (stats, FNV digest of what host_bench digests)."""
    import ctypes as C
    from dynslam_amd import _capi
    fx, fy, cx, cy = sc.intrinsics()
    e = make_engine()
    for i in range(n):
        rgba, d, T, _ = sc.frame(i)
        rgba = rgba.copy(); rgba[..., 3] = 255  # CvToItm sets alpha to 255 (InfiniTamDriver.cpp:94)
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        e.process_frame()
        e.prepare()
        e.decay(1, 1, False)
    st = e.get_stats()
    col, _ = e.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, intrinsics=[fx, fy, cx, cy])
    _, dep = e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=M, intrinsics=[fx, fy, cx, cy], want_rgba=False, want_depth=True)
    vdepth = e.get_view()[1]
    mm = np.empty(W * H, np.int16)
    bgr = np.empty((W * H, 3), np.uint8)
    assert api.depth_m_to_mm(dep.ctypes.data_as(C.c_void_p), mm.ctypes.data_as(C.c_void_p), W * H) == 0
    assert api.rgba_to_bgr(col.ctypes.data_as(C.c_void_p), bgr.ctypes.data_as(C.c_void_p), W * H) == 0
    h = fnv(bgr.tobytes(), fnv(mm.tobytes(), fnv(vdepth.tobytes(), fnv(dep.tobytes(), fnv(col.tobytes())))))
    e.close()
    return st, h, dep


@pytest.mark.parametrize("driver", ["reference", "shim"])
def test_cpp_drivers_run_on_the_cpu_oracle(oracle_lib, tmp_path, driver):
    """The same host — the reference's unmodified InfiniTamDriver over shim/ITMLib.h — with every dsr_* entry point
    renamed to the CPU oracle's orc_* at compile time (a generated -include header: TEST-ONLY, the shipped shim binds
    libdsr_hip.so): the reference's driver logic and the shim run in the CPU suite too, and must reproduce what the
    oracle gives when driven directly."""
    from dynslam_amd import _capi
    from dynslam_amd.engine import make_calib
    from dynslam_amd.synth import StreetScene
    from oracle.oracle import LIB_PATH, OracleEngine, oracle_settings
    if driver == "reference" and not have_ref:
        pytest.skip("/root/reference is not on this machine")
    rename = tmp_path / "dsr_to_orc.h"
    rename.write_text("".join(f"#define dsr_{name} orc_{name}\n" for name in _capi.SIGNATURES))
    exe = tmp_path / "ref_driver_host_cpu"
    odir = os.path.dirname(LIB_PATH)
    if driver == "reference":
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-DNDEBUG", "-DDSR_HOST_REFERENCE_DRIVER", "-include", str(rename)] + REF_INC +
                              [os.path.join(ROOT, "shim", "host_bench.cpp"), os.path.join(REF, "InfiniTamDriver.cpp"),
                               os.path.join(REF, "Utils.cpp"), "-o", str(exe), "-L", odir, "-loracle", f"-Wl,-rpath,{odir}"])
    else:  # our own HostDriver (shim/host_bench.cpp) — the host of bench.py's through_shim leg
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-include", str(rename), "-I", os.path.join(ROOT, "shim"),
                               os.path.join(ROOT, "shim", "host_bench.cpp"), "-o", str(exe), "-L", odir, "-loracle", f"-Wl,-rpath,{odir}"])
    W, H, n = 128, 48, 4
    sc = StreetScene(W, H)
    path = tmp_path / "frames.bin"
    M = write_frames_file(path, sc, n)
    fx, fy, cx, cy = sc.intrinsics()
    st, h, dep = _mirror_digest(lambda: OracleEngine(oracle_settings(**KW), make_calib(fx, fy, cx, cy, W, H)), oracle_lib, sc, n, W, H, M)
    out = subprocess.check_output([str(exe)] + _host_args(path, sc, n, W, H)).decode().strip()
    got = dict(kv.split("=") for kv in out.split())
    assert got["driver"] == driver and (dep > 0).mean() > 0.05
    assert int(got["used_bytes"]) == 8 * 512 * (st.num_allocated_voxel_blocks - st.last_free_block_id), out
    assert int(got["saved_bytes"]) == st.decayed_block_count * 4096, out
    assert got["hash"] == f"{h:016x}", out


def write_masks_file(path, frames):
    """masks.bin of shim/host_bench.cpp --masks: per frame n, then (k, x0, y0, bw, bh, rel[16] row-major, mask bytes)."""
    import struct
    with open(path, "wb") as f:
        for _, _, _, masks in frames:
            f.write(struct.pack("<i", len(masks)))
            for k, x0, y0, mask, rel in masks:
                f.write(struct.pack("<5i", k, x0, y0, mask.shape[1], mask.shape[0]))
                f.write(np.ascontiguousarray(rel, np.float32).tobytes())
                f.write(np.ascontiguousarray(mask, np.uint8).tobytes())


INST_KW = dict(voxel_size=0.035, mu=1.0, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
               sdf_local_block_num=7142, hash_bucket_num=0x100000, excess_list_size=0x20000)


def _instances_case(tmp_path, W, H, n, n_inst):
    from bench import _gen_frame
    from dynslam_amd.synth import StreetScene
    sc = StreetScene(W, H, n_instances=n_inst)
    frames = [_gen_frame((W, H, i, n_inst)) for i in range(n)]
    path, mpath = tmp_path / "frames.bin", tmp_path / "masks.bin"
    M = np.linalg.inv(np.asarray(frames[-1][2], np.float64)).astype(np.float32)
    with open(path, "wb") as f:
        for rgba, d, T, _ in frames:
            f.write(np.ascontiguousarray(rgba[..., 2::-1]).tobytes())
            f.write(np.ascontiguousarray(d, np.int16).tobytes())
            f.write(np.ascontiguousarray(T, np.float32).tobytes())
        f.write(M.tobytes())
    write_masks_file(mpath, frames)
    return sc, frames, path, mpath, M


def _instances_mirror(make_engine, api, sc, frames, W, H, M, n_inst):
    """configs[2] through the Python mirror: bench.py's step (GPU view split + one volume per instance)."""
    import ctypes as C
    from dynslam_amd import _capi
    fx, fy, cx, cy = sc.intrinsics()
    e = make_engine(KW)
    inst = [make_engine(INST_KW) for _ in range(n_inst)]
    for rgba, d, T, masks in frames:
        rgba = rgba.copy(); rgba[..., 3] = 255
        e.update_view(rgba, d)
        for k, x0, y0, mask, rel in masks:
            e.extract_silhouette(inst[k], mask, x0, y0)
            e.remove_silhouette(mask, x0, y0)
            inst[k].set_pose_inv_m(rel)
            inst[k].process_frame()
            inst[k].prepare()
        e.set_pose_inv_m(T)
        e.process_frame()
        e.prepare()
    col, _ = e.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, intrinsics=[fx, fy, cx, cy])
    _, dep = e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=M, intrinsics=[fx, fy, cx, cy], want_rgba=False, want_depth=True)
    vdepth = e.get_view()[1]
    mm = np.empty(W * H, np.int16)
    bgr = np.empty((W * H, 3), np.uint8)
    assert api.depth_m_to_mm(dep.ctypes.data_as(C.c_void_p), mm.ctypes.data_as(C.c_void_p), W * H) == 0
    assert api.rgba_to_bgr(col.ctypes.data_as(C.c_void_p), bgr.ctypes.data_as(C.c_void_p), W * H) == 0
    h = fnv(bgr.tobytes(), fnv(mm.tobytes(), fnv(vdepth.tobytes(), fnv(dep.tobytes(), fnv(col.tobytes())))))
    inst_used = 0
    for ie in inst:
        st = ie.get_stats()
        inst_used += 8 * 512 * (st.num_allocated_voxel_blocks - st.last_free_block_id)
        ie.close()
    st = e.get_stats()
    e.close()
    return st, h, inst_used


def _instances_args(path, mpath, sc, n, W, H, n_inst):
    return _host_args(path, sc, n, W, H)[:14] + ["--masks", str(mpath), str(n_inst)]


def test_cpp_host_with_instance_volumes_runs_on_the_cpu_oracle(oracle_lib, tmp_path):
    """configs[2] through the C++ host (shim/host_bench.cpp --masks: GPU view split + one HostDriver per instance
    volume), oracle-backed like the test above: same digest and the same instance-volume sizes as the Python mirror."""
    from dynslam_amd import _capi
    from dynslam_amd.engine import make_calib
    from oracle.oracle import LIB_PATH, OracleEngine, oracle_settings
    rename = tmp_path / "dsr_to_orc.h"
    rename.write_text("".join(f"#define dsr_{name} orc_{name}\n" for name in _capi.SIGNATURES))
    exe = tmp_path / "host_bench_cpu"
    odir = os.path.dirname(LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-include", str(rename), "-I", os.path.join(ROOT, "shim"),
                           os.path.join(ROOT, "shim", "host_bench.cpp"), "-o", str(exe), "-L", odir, "-loracle", f"-Wl,-rpath,{odir}"])
    W, H, n, n_inst = 160, 56, 3, 2
    sc, frames, path, mpath, M = _instances_case(tmp_path, W, H, n, n_inst)
    assert sum(len(f[3]) for f in frames) >= n  # the silhouettes are really there
    calib = make_calib(*sc.intrinsics(), W, H)
    st, h, inst_used = _instances_mirror(lambda kw: OracleEngine(oracle_settings(**kw), calib), oracle_lib, sc, frames, W, H, M, n_inst)
    out = subprocess.check_output([str(exe)] + _instances_args(path, mpath, sc, n, W, H, n_inst)).decode().strip()
    got = dict(kv.split("=") for kv in out.split())
    assert int(got["instances"]) == n_inst and int(got["inst_used_bytes"]) == inst_used > 2 * 4096, out
    assert int(got["used_bytes"]) == 8 * 512 * (st.num_allocated_voxel_blocks - st.last_free_block_id), out
    assert got["hash"] == f"{h:016x}", out
    # one volume per GPU through the C ABI (--devices: ITMLibSettings::deviceIndex + dsr_exchange_*): the fused preview served by
    # the exchange equals the reference's own flow (every volume's render to the host, composited there) — with one rank, with
    # two ranks "on one device" and with the preview inside the frame loop
    assert int(got["composite_hash"], 16) != 0
    for extra, ranks in (["--devices", "0"], 1), (["--devices", "0,0"], 2), (["--devices", "0,0,0", "--preview"], 3), (["--preview"], 1):
        o2 = dict(kv.split("=") for kv in subprocess.check_output([str(exe)] + _instances_args(path, mpath, sc, n, W, H, n_inst) + extra).decode().split())
        assert o2["composite_hash"] == got["composite_hash"] and o2["hash"] == got["hash"] and int(o2["ranks"]) == ranks, (extra, o2)


def build_oracle_host(tmp_path):
    """shim/host_bench.cpp with every dsr_* call renamed to the CPU oracle's orc_* (TEST-ONLY build: the checker of the -m gpu tests)"""
    from dynslam_amd import _capi
    from oracle.oracle import LIB_PATH
    rename = tmp_path / "dsr_to_orc.h"
    rename.write_text("".join(f"#define dsr_{name} orc_{name}\n" for name in _capi.SIGNATURES))
    exe = tmp_path / "host_bench_cpu"
    odir = os.path.dirname(LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-include", str(rename), "-I", os.path.join(ROOT, "shim"),
                           os.path.join(ROOT, "shim", "host_bench.cpp"), "-o", str(exe), "-L", odir, "-loracle", f"-Wl,-rpath,{odir}"])
    return str(exe)


@pytest.mark.gpu
def test_cpp_host_one_volume_per_gpu_through_the_exchange(hip_api, oracle_lib, tmp_path):
    """`shim/host_bench --masks --devices N` — the volume-per-GPU split (ITMLibSettings::deviceIndex), the
    cross-GPU view split and the fused preview through dsr_exchange_* (RCCL called by the library) from a C++ host.  On a
    one-GPU box: devices {0,0} (two ranks, one GPU: the exchange degenerates to layers in place), the same with the cross-GPU
    code paths forced (peer copy of the cut-out; a 1-rank RCCL communicator and its all-gather) — each equals the
    single-engine run (no exchange: renders to the host + dsr_composite_instances) and the SAME host on the CPU oracle."""
    W, H, n, n_inst = 320, 96, 4, 3
    sc, frames, path, mpath, M = _instances_case(tmp_path, W, H, n, n_inst)
    assert sum(len(f[3]) for f in frames) >= n
    base = _instances_args(path, mpath, sc, n, W, H, n_inst)
    ref = dict(kv.split("=") for kv in subprocess.check_output([build_oracle_host(tmp_path)] + base).decode().split())
    assert int(ref["composite_hash"], 16) != 0
    hip = build_shim_host()
    runs = [([], {}), (["--devices", "0,0"], {}), (["--devices", "0,0", "--preview"], {}),
            (["--devices", "0,0,0"], {"DSR_FORCE_PEER_PATH": "1"}), (["--devices", "0"], {"DSR_EXCHANGE_FORCE_RCCL": "1"}),
            (["--devices", "0,0"], {"DSR_EXCHANGE_FORCE_RCCL": "1", "DSR_FORCE_PEER_PATH": "1"})]
    for extra, env in runs:
        out = subprocess.check_output([hip] + base + extra, env=dict(os.environ, **env), timeout=300).decode()
        lines = [ln for ln in out.splitlines() if ln.startswith("driver=")]
        assert len(lines) == 1 and out.strip().startswith("driver="), out  # nothing but the host's own line on stdout (no RCCL banner)
        got = dict(kv.split("=") for kv in lines[0].split())
        assert got["composite_hash"] == ref["composite_hash"] and got["hash"] == ref["hash"], (extra, env, got, ref)
        assert got["inst_used_bytes"] == ref["inst_used_bytes"] and got["used_bytes"] == ref["used_bytes"]


@pytest.mark.gpu
def test_cpp_host_with_instance_volumes_on_the_hip_engine(hip_api, tmp_path):
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    W, H, n, n_inst = 160, 56, 3, 2
    sc, frames, path, mpath, M = _instances_case(tmp_path, W, H, n, n_inst)
    calib = make_calib(*sc.intrinsics(), W, H)
    st, h, inst_used = _instances_mirror(lambda kw: EngineCore(default_settings(**kw), calib), hip_api, sc, frames, W, H, M, n_inst)
    out = subprocess.check_output([build_shim_host()] + _instances_args(path, mpath, sc, n, W, H, n_inst)).decode().strip()
    got = dict(kv.split("=") for kv in out.split())
    assert int(got["inst_used_bytes"]) == inst_used > 2 * 4096 and got["hash"] == f"{h:016x}", out


@pytest.mark.gpu
def test_reference_driver_runs_on_the_hip_engine(hip_api, tmp_path):
    import ctypes as C
    from dynslam_amd import _capi
    from dynslam_amd.engine import EngineCore, default_settings, make_calib
    from dynslam_amd.synth import StreetScene
    W, H, n = 128, 48, 4
    sc = StreetScene(W, H)
    path = tmp_path / "frames.bin"
    M = write_frames_file(path, sc, n)
    fx, fy, cx, cy = sc.intrinsics()
    args = [str(path), str(W), str(H), repr(fx), repr(fy), repr(cx), repr(cy), str(n), "1", repr(KW["voxel_size"]), repr(KW["mu"]),
            str(KW["sdf_local_block_num"]), str(KW["hash_bucket_num"]), str(KW["excess_list_size"]), "1", "1"]
    exes = [build_shim_host()]
    if have_ref:
        build_ref_host()
    if os.path.exists(REF_EXE):  # prebuilt where /root/reference exists; shipped to the GPU box with the snapshot
        exes.append(REF_EXE)
    else:
        pytest.fail("tests/refhost/_build/ref_driver_host is missing: __graft_entry__.build() makes it where /root/reference exists")
    # the same call sequence through the Python mirror
    e = EngineCore(default_settings(**KW), make_calib(fx, fy, cx, cy, W, H))
    for i in range(n):
        rgba, d, T, _ = sc.frame(i)
        rgba = rgba.copy(); rgba[..., 3] = 255  # CvToItm sets alpha to 255 (InfiniTamDriver.cpp:94)
        e.update_view(rgba, d)
        e.set_pose_inv_m(T)
        e.process_frame()
        e.prepare()
        e.decay(1, 1, False)
    st = e.get_stats()
    col, _ = e.get_image(_capi.IMAGE_FREECAMERA_COLOUR_FROM_VOLUME, pose_m=M, intrinsics=[fx, fy, cx, cy])
    _, dep = e.get_image(_capi.IMAGE_FREECAMERA_DEPTH, pose_m=M, intrinsics=[fx, fy, cx, cy], want_rgba=False, want_depth=True)
    vdepth = e.get_view()[1]
    mm = np.empty(W * H, np.int16)
    bgr = np.empty((W * H, 3), np.uint8)
    assert e.api.depth_m_to_mm(dep.ctypes.data_as(C.c_void_p), mm.ctypes.data_as(C.c_void_p), W * H) == 0
    assert e.api.rgba_to_bgr(col.ctypes.data_as(C.c_void_p), bgr.ctypes.data_as(C.c_void_p), W * H) == 0
    h = fnv(bgr.tobytes(), fnv(mm.tobytes(), fnv(vdepth.tobytes(), fnv(dep.tobytes(), fnv(col.tobytes())))))
    assert (dep > 0).mean() > 0.05 and st.decayed_block_count > 0
    for exe in exes:
        out = subprocess.check_output([exe] + args).decode().strip()
        got = dict(kv.split("=") for kv in out.split())
        assert int(got["used_bytes"]) == 8 * 512 * (st.num_allocated_voxel_blocks - st.last_free_block_id), out
        assert int(got["saved_bytes"]) == st.decayed_block_count * 4096, out
        assert got["hash"] == f"{h:016x}", out
    assert "driver=reference" in out
