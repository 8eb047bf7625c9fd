"""The C-ABI libraries load and export every symbol include/dsr.h declares; struct layouts
in dynslam_amd/_capi.py equal the C ones.  No compute calls (runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from dynslam_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dsr.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_bindings_agree():
    names = declared_functions()
    assert names, "no functions parsed from dsr.h"
    assert sorted("dsr_" + k for k in _capi.SIGNATURES) == names


def test_hip_library_exports_every_symbol():
    path = os.path.join(ROOT, "dynslam_amd", "csrc", "libdsr_hip.so")
    assert os.path.exists(path), "libdsr_hip.so not built: run __graft_entry__.build()"
    _capi.preload_hip_runtime()
    lib = C.CDLL(path)
    api = _capi.bind(lib, "dsr_")  # AttributeError if a symbol is missing
    assert api.abi_version() == _capi.ABI_VERSION
    # ... and the header's: a struct-layout change must bump all three together
    assert int(re.search(r"#define\s+DSR_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1)) == _capi.ABI_VERSION
    s = _capi.Settings()
    api.default_settings(C.byref(s))
    # upstream ITMLibSettings defaults
    assert abs(s.voxel_size - 0.005) < 1e-9 and abs(s.mu - 0.02) < 1e-9 and s.max_w == 100
    assert s.hash_bucket_num == 0x100000 and s.excess_list_size == 0x20000 and s.sdf_local_block_num == 0x40000


def test_oracle_library_exports_every_symbol(oracle_lib):
    assert oracle_lib.abi_version() == _capi.ABI_VERSION
    for name in _capi.SIGNATURES:
        assert hasattr(oracle_lib.lib, "orc_" + name)


def test_struct_layouts_match_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dsr.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(dsr_hash_entry),sizeof(dsr_voxel),sizeof(dsr_settings),sizeof(dsr_intrinsics),sizeof(dsr_calib),"
                   "sizeof(dsr_stats),sizeof(dsr_kernel_time),offsetof(dsr_stats,decayed_block_count),offsetof(dsr_settings,device));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_capi.HashEntry), C.sizeof(_capi.Voxel), C.sizeof(_capi.Settings), C.sizeof(_capi.Intrinsics),
            C.sizeof(_capi.Calib), C.sizeof(_capi.Stats), C.sizeof(_capi.KernelTime),
            _capi.Stats.decayed_block_count.offset, _capi.Settings.device.offset]
    assert got == want
    assert got[0] == 16 and got[1] == 8


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under dynslam_amd/, include/ or shim/ may
    reference it."""
    bad = []
    for base in ("dynslam_amd", "include", "shim"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|#include\s+\"[^\"]*oracle", txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_hip_library_fails_loudly(monkeypatch):
    import dynslam_amd.engine as eng
    monkeypatch.setattr(eng, "HIP_LIB_PATH", "/nonexistent/libdsr_hip.so")
    monkeypatch.setattr(eng, "_hip_api", None)
    with pytest.raises(ImportError):
        eng.load_hip_api()
