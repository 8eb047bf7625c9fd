"""The edges of the path against the REFERENCE'S OWN CODE (oracle/_ref/libref_edges.so = the reference's unmodified
InstanceReconstructor.cpp / InfiniTamDriver.cpp / DepthProvider.h compiled from where they lie, oracle/ref_edges.cpp):

  ProcessSilhouette_CPU / RemoveSilhouette_CPU   InstanceReconstructor.cpp:59-170   dsr_view_extract/remove_silhouette
  CompositeColor / CompositeDepth                InstanceReconstructor.cpp:851-908  dsr_composite_instances
  DepthProvider::DepthFromDisparityMap<float>    DepthProvider.h:94-137             dsr_depth_from_disparity
  CvToItm / ItmToCv / FloatDepthmapToShort       InfiniTamDriver.cpp:81-139         dsr_bgr_to_rgba / rgba_to_bgr / depth_m_to_mm

CPU: the oracle's restatements == the reference's functions (so these rows of the oracle are pinned by the reference
itself, not by a reading of it).  GPU (-m gpu): the HIP kernels == the reference's functions.  The library is built
where /root/reference exists (`make -C oracle _ref`, also done by __graft_entry__.build()) and travels to the GPU box."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from dynslam_amd.engine import EngineCore, default_settings, make_calib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_edges.so")
HAVE_REF_SRC = os.path.isdir("/root/reference/src/DynSLAM")
vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


@pytest.fixture(scope="module")
def ref():
    if HAVE_REF_SRC:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(REF_LIB):
        msg = ("oracle/_ref/libref_edges.so is missing: __graft_entry__.build() makes it where /root/reference exists "
               "and it travels to the GPU box with the snapshot")
        import torch
        if torch.cuda.is_available():  # on a GPU box these are the only tests pinned by the reference's own code
            pytest.fail(msg)
        pytest.skip(msg)
    from dynslam_amd import _capi
    _capi.preload_hip_runtime()
    return C.CDLL(REF_LIB)


W, H = 96, 40
KW = dict(voxel_size=0.05, mu=0.2, max_w=100, view_frustum_min=0.2, view_frustum_max=30.0,
          sdf_local_block_num=2048, hash_bucket_num=1024, excess_list_size=256)


def frame(seed):
    rng = np.random.default_rng(seed)
    rgba = rng.integers(0, 256, (H, W, 4)).astype(np.uint8)
    depth = rng.uniform(0.5, 20.0, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.15] = -1.0  # invalid after conversion (InstanceReconstructor.cpp:97)
    return rgba, depth


def boxes(seed):
    """bbox-local masks incl. ones sticking out of the frame on every side."""
    rng = np.random.default_rng(seed)
    out = []
    for x0, y0, bw, bh in [(10, 5, 30, 20), (-7, -3, 25, 15), (80, 30, 40, 25), (0, 0, W, H), (50, 10, 1, 1)]:
        out.append((x0, y0, (rng.random((bh, bw)) < 0.6).astype(np.uint8)))
    # boxes entirely outside the frame / much taller than it, and masks holding values other than 0 / 1 (only 1 selects)
    for x0, y0, bw, bh in [(-50, -50, 20, 20), (200, 5, 10, 10), (5, -100, 10, 300), (20, 8, 40, 20)]:
        out.append((x0, y0, rng.choice(np.array([0, 1, 2, 255], np.uint8), (bh, bw))))
    return out


def check_silhouettes(api, ref):
    calib = make_calib(50.0, 50.0, W / 2, H / 2, W, H)
    main = EngineCore(default_settings(api=api, **KW), calib, api=api)
    inst = EngineCore(default_settings(api=api, **KW), calib, api=api)
    rgba, depth = frame(1)
    main.set_view_float(rgba, depth)
    cur_rgba, cur_depth = rgba.copy(), depth.copy()  # the reference's main view
    for x0, y0, mask in boxes(2):
        bh, bw = mask.shape
        main.extract_silhouette(inst, mask, x0, y0)
        want_rgba, want_depth = np.empty_like(rgba), np.empty_like(depth)
        assert ref.ref_process_silhouette(vp(cur_rgba), vp(cur_depth), vp(want_rgba), vp(want_depth), W, H, vp(mask), x0, y0, bw, bh) == 0
        got_rgba, got_depth = inst.get_view()
        assert np.array_equal(got_depth, want_depth) and np.array_equal(got_rgba, want_rgba), (x0, y0)
        main.remove_silhouette(mask, x0, y0)
        assert ref.ref_remove_silhouette(vp(cur_rgba), vp(cur_depth), W, H, vp(mask), x0, y0, bw, bh) == 0
        got_rgba, got_depth = main.get_view()
        assert np.array_equal(got_depth, cur_depth) and np.array_equal(got_rgba, cur_rgba), (x0, y0)
    main.close(); inst.close()


def check_composite(api, ref):
    rng = np.random.default_rng(4)
    P = W * H
    bg_c = rng.integers(0, 256, (P, 4)).astype(np.uint8)
    bg_d = rng.uniform(1, 9, P).astype(np.float32); bg_d[rng.random(P) < 0.3] = 0
    layers_c = rng.integers(0, 256, (4, P, 4)).astype(np.uint8)
    layers_d = rng.uniform(1, 9, (4, P)).astype(np.float32); layers_d[rng.random((4, P)) < 0.5] = 0
    layers_d[1, :50] = bg_d[:50]  # exact ties: the strict `t > s` must keep the target
    ids = np.array([3, 7, 12, 31], np.int32)
    for tint_strength in (1.0, 0.5, 0.0):
        t_c, t_d = bg_c.copy(), bg_d.copy()
        assert api.composite_instances(vp(t_c), vp(t_d), vp(layers_c), vp(layers_d), vp(ids), 4, P, tint_strength, 0) == 0
        w_c, w_d = bg_c.copy(), bg_d.copy()
        for k in range(4):  # the host's loop over tracks in ascending id (InstanceReconstructor.cpp:960-990)
            assert ref.ref_composite_color(vp(w_c), vp(w_d), vp(layers_c[k]), vp(layers_d[k]), W, H, int(ids[k]), C.c_float(tint_strength)) == 0
        assert np.array_equal(t_d, w_d) and np.array_equal(t_c, w_c), tint_strength
    # depth only: CompositeDepth == the composite without colour buffers
    t_d = bg_d.copy()
    assert api.composite_instances(None, vp(t_d), None, vp(layers_d), vp(ids), 4, P, 1.0, 0) == 0
    w_d = bg_d.copy()
    for k in range(4):
        assert ref.ref_composite_depth(vp(w_d), vp(layers_d[k]), W, H) == 0
    assert np.array_equal(t_d, w_d)


def check_disparity_and_conversions(api, ref):
    rng = np.random.default_rng(6)
    P = W * H
    disp = rng.uniform(-2.0, 130.0, P).astype(np.float32)
    disp[:40] = [0.0, 1e-6, -1e-6, 1e-5, 2e-5, 0.5, 19.0, 19.1, 757.0, 760.0] * 4
    for scale, lo, hi in ((1.0, 0.5, 20.0), (0.75, 0.3, 12.5), (1.0, 0.0, 32.0)):
        got, want = np.empty(P, np.int16), np.empty(P, np.int16)
        assert api.depth_from_disparity(vp(disp), vp(got), P, 0.537150654273, 707.0912, scale, lo, hi) == 0
        assert ref.ref_depth_from_disparity(vp(disp), vp(want), W, H, C.c_float(0.537150654273), C.c_float(707.0912), C.c_float(scale),
                                            C.c_float(lo), C.c_float(hi)) == 0
        assert np.array_equal(got, want), (scale, lo, hi)
    bgr = rng.integers(0, 256, (P, 3)).astype(np.uint8)
    got, want = np.empty((P, 4), np.uint8), np.empty((P, 4), np.uint8)
    assert api.bgr_to_rgba(vp(bgr), vp(got), P) == 0 and ref.ref_cv_to_itm(vp(bgr), vp(want), W, H) == 0
    assert np.array_equal(got, want)
    back, wback = np.empty((P, 3), np.uint8), np.empty((P, 3), np.uint8)
    assert api.rgba_to_bgr(vp(got), vp(back), P) == 0 and ref.ref_itm_to_cv(vp(got), vp(wback), W, H) == 0
    assert np.array_equal(back, wback) and np.array_equal(back, bgr)
    depth_m = rng.uniform(0.0, 32.7, P).astype(np.float32)  # within int16 mm: the cast is defined
    depth_m[:5] = [0.0, 0.0005, 1.0, 19.9999, 32.767]
    mm, wmm = np.empty(P, np.int16), np.empty(P, np.int16)
    assert api.depth_m_to_mm(vp(depth_m), vp(mm), P) == 0 and ref.ref_float_depthmap_to_short(vp(depth_m), vp(wmm), W, H) == 0
    assert np.array_equal(mm, wmm)


def check_precomputed_provider(api, ref, tmp_path):
    """The reference's PrecomputedDepthProvider::GetDepth (its own ReadPrecomputed logic, clamp and disparity loops; the
    absent cv::FileStorage / pfmLib underneath replaced by stand-ins over dsr_read_depth_xml / dsr_read_pfm) == the
    Python mirror dynslam_amd/depth_io.py over `api`."""
    from dynslam_amd.depth_io import DepthIOError, PrecomputedDepthProvider
    from tests.test_depth_io import write_cv_xml, write_pfm
    rng = np.random.default_rng(11)
    depth = rng.integers(0, 32767, (H, W)).astype(np.int16)
    depth[rng.random((H, W)) < 0.1] = 0
    write_cv_xml(tmp_path / "0007.xml", depth)
    disp = rng.uniform(0.0, 120.0, (H, W)).astype(np.float32)
    disp[rng.random((H, W)) < 0.05] = 0.0
    write_pfm(tmp_path / "000007.pfm", disp, little=False)
    err = C.create_string_buffer(256)
    for fmt, is_depth in (("%04d.xml", 1), ("%06d.pfm", 0)):
        want = np.empty((H, W), np.int16)
        st = ref.ref_precomputed_get_depth(str(tmp_path).encode(), fmt.encode(), is_depth, C.c_float(0.5), C.c_float(20.0), 7,
                                           C.c_float(0.537150654273), C.c_float(707.0912), C.c_float(1.0), vp(want), W, H, err, 256)
        assert st == 0, err.value
        got = PrecomputedDepthProvider(tmp_path, fmt, bool(is_depth), 0.5, 20.0, api=api).GetDepth(7, 0.537150654273, 707.0912)
        assert np.array_equal(got, want) and (got > 0).mean() > 0.3, fmt
    # a missing file: std::runtime_error in the reference, DepthIOError in the mirror
    assert ref.ref_precomputed_get_depth(str(tmp_path).encode(), b"%04d.xml", 1, C.c_float(0.5), C.c_float(20.0), 8, C.c_float(0.5),
                                         C.c_float(700.0), C.c_float(1.0), vp(np.empty((H, W), np.int16)), W, H, err, 256) == 1
    assert b"precomputed depth" in err.value
    with pytest.raises(DepthIOError):
        PrecomputedDepthProvider(tmp_path, "%04d.xml", True, 0.5, 20.0, api=api).GetDepth(8, 0.5, 700.0)


def test_oracle_precomputed_depth_provider_equals_reference_code(oracle_lib, ref, tmp_path):
    check_precomputed_provider(oracle_lib, ref, tmp_path)


@pytest.mark.gpu
def test_hip_precomputed_depth_provider_equals_reference_code(hip_api, ref, tmp_path):
    check_precomputed_provider(hip_api, ref, tmp_path)


def test_oracle_silhouettes_equal_reference_code(oracle_lib, ref):
    check_silhouettes(oracle_lib, ref)


def test_oracle_composite_equals_reference_code(oracle_lib, ref):
    check_composite(oracle_lib, ref)


def test_oracle_disparity_and_conversions_equal_reference_code(oracle_lib, ref):
    check_disparity_and_conversions(oracle_lib, ref)


@pytest.mark.gpu
def test_hip_silhouettes_equal_reference_code(hip_api, ref):
    check_silhouettes(hip_api, ref)


@pytest.mark.gpu
def test_hip_composite_equals_reference_code(hip_api, ref):
    check_composite(hip_api, ref)


@pytest.mark.gpu
def test_hip_disparity_and_conversions_equal_reference_code(hip_api, ref):
    check_disparity_and_conversions(hip_api, ref)


# --- special values: NaN, infinities, signed zeros, denormals, exact ties -----------------------------------------------------

_SPECIAL_DEPTH = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, -1.0, 1e-40, -1e-40, 3.0, 3.0000002, 1e30], np.float32)
_SPECIAL_DISP = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-5, 9.9e-6, -9.9e-6, 1e-6, -3.0, 0.5, 19.0, 37.98, 759.0, 1e-38, 1e20], np.float32)


def check_special_values(api, ref, trials):
    """Compositing and disparity conversion on buffers drawn from the awkward floats: the reference's comparisons
    (`s != 0 && (t == 0 || t > s)`, `abs(disp) < 1e-5`, the int casts) decide every case, bit for bit."""
    P = W * H
    rng = np.random.default_rng(21)
    for _ in range(trials):
        bg_c = rng.integers(0, 256, (P, 4)).astype(np.uint8)
        bg_d = rng.choice(_SPECIAL_DEPTH, P).astype(np.float32)
        L = 3
        lc = rng.integers(0, 256, (L, P, 4)).astype(np.uint8)
        ld = rng.choice(_SPECIAL_DEPTH, (L, P)).astype(np.float32)
        ids = np.sort(rng.choice(60, L, replace=False)).astype(np.int32)
        ts = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
        t_c, t_d = bg_c.copy(), bg_d.copy()
        assert api.composite_instances(vp(t_c), vp(t_d), vp(lc), vp(ld), vp(ids), L, P, ts, 0) == 0
        w_c, w_d = bg_c.copy(), bg_d.copy()
        for k in range(L):
            assert ref.ref_composite_color(vp(w_c), vp(w_d), vp(lc[k]), vp(ld[k]), W, H, int(ids[k]), C.c_float(ts)) == 0
        assert np.array_equal(t_d.view(np.uint32), w_d.view(np.uint32)) and np.array_equal(t_c, w_c)
        disp = rng.choice(_SPECIAL_DISP, P).astype(np.float32)
        scale, lo, hi = float(rng.choice([1.0, 0.75, 0.5])), float(rng.choice([0.0, 0.5])), float(rng.choice([20.0, 32.0]))
        got, want = np.empty(P, np.int16), np.empty(P, np.int16)
        assert api.depth_from_disparity(vp(disp), vp(got), P, 0.537150654273, 707.0912, scale, lo, hi) == 0
        assert ref.ref_depth_from_disparity(vp(disp), vp(want), W, H, C.c_float(0.537150654273), C.c_float(707.0912), C.c_float(scale),
                                            C.c_float(lo), C.c_float(hi)) == 0
        assert np.array_equal(got, want), (scale, lo, hi)


def test_oracle_edges_equal_reference_code_on_special_values(oracle_lib, ref):
    check_special_values(oracle_lib, ref, 60)


@pytest.mark.gpu
def test_hip_edges_equal_reference_code_on_special_values(hip_api, ref):
    check_special_values(hip_api, ref, 20)


# --- the HIP kernels' per-element device functions, compiled for the CPU (tests/hostsim), against the reference's own code ----
# (k_edges.h / k_composite.h: one thread per element around a __host__ __device__ function; here those functions run on the
#  host — the same source the GPU runs, pinned by the reference itself without a GPU.)

class _DeviceFunctionsOnHost:
    """The five buffer-level entry points the checks above call, bound to the host stand-in library."""

    def __init__(self, lib):
        F, I, P = C.c_float, C.c_int, C.c_void_p
        sig = {"depth_from_disparity": [P, P, I, F, F, F, F, F], "bgr_to_rgba": [P, P, I], "rgba_to_bgr": [P, P, I],
               "depth_m_to_mm": [P, P, I], "composite_instances": [P, P, P, P, P, I, I, F, I]}
        for name, args in sig.items():
            fn = getattr(lib, "hs_" + name)
            fn.restype, fn.argtypes = I, args
            setattr(self, name, fn)
        self.lib = lib


@pytest.fixture(scope="module")
def devfn_host():
    from tests.test_device_functions_host import _lib
    return _DeviceFunctionsOnHost(_lib())


def test_device_functions_composite_equals_reference_code(devfn_host, ref):
    check_composite(devfn_host, ref)


def test_device_functions_disparity_and_conversions_equal_reference_code(devfn_host, ref):
    check_disparity_and_conversions(devfn_host, ref)


def test_device_functions_equal_reference_code_on_special_values(devfn_host, ref):
    check_special_values(devfn_host, ref, 60)


def test_device_functions_silhouettes_equal_reference_code(devfn_host, ref):
    """k_extract_silhouette / k_remove_silhouette's per-pixel functions on plain buffers == ProcessSilhouette_CPU /
    RemoveSilhouette_CPU, over boxes sticking out of the frame on every side and masks with values other than 0 / 1."""
    lib = devfn_host.lib
    rgba, depth = frame(1)
    cur_rgba, cur_depth = rgba.copy(), depth.copy()      # the reference's main view
    our_rgba, our_depth = rgba.copy(), depth.copy()      # ours
    for x0, y0, mask in boxes(2):
        bh, bw = mask.shape
        got_rgba, got_depth = np.empty_like(rgba), np.empty_like(depth)
        assert lib.hs_extract_silhouette(vp(our_rgba), vp(our_depth), vp(got_rgba), vp(got_depth), W, H, vp(mask), x0, y0, bw, bh) == 0
        want_rgba, want_depth = np.empty_like(rgba), np.empty_like(depth)
        assert ref.ref_process_silhouette(vp(cur_rgba), vp(cur_depth), vp(want_rgba), vp(want_depth), W, H, vp(mask), x0, y0, bw, bh) == 0
        assert np.array_equal(got_depth, want_depth) and np.array_equal(got_rgba, want_rgba), (x0, y0)
        assert lib.hs_remove_silhouette(vp(our_rgba), vp(our_depth), W, H, vp(mask), x0, y0, bw, bh) == 0
        assert ref.ref_remove_silhouette(vp(cur_rgba), vp(cur_depth), W, H, vp(mask), x0, y0, bw, bh) == 0
        assert np.array_equal(our_depth, cur_depth) and np.array_equal(our_rgba, cur_rgba), (x0, y0)

