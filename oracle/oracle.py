"""ctypes binding of the CPU oracle (oracle/liboracle.so, prefix `orc_`).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product (dynslam_amd/).
"""
import ctypes as C
import os
import subprocess

from dynslam_amd import _capi
from dynslam_amd.engine import EngineCore, InfiniTamDriver, default_settings  # noqa: F401 (re-export)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_api = None


def build(force=False):
    src = os.path.join(_HERE, "dsr_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "dsr.h")
    stale = (not os.path.exists(LIB_PATH)
             or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr),
                                                  os.path.getmtime(os.path.join(_HERE, "mc_tables.h"))))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)


def load_api():
    global _api
    if _api is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        _api = _capi.bind(lib, "orc_")
        _api.set_threads = lib.orc_set_threads
        _api.set_threads.restype = C.c_int
        _api.set_threads.argtypes = [C.c_void_p, C.c_int]
    return _api


def oracle_settings(**overrides):
    return default_settings(api=load_api(), **overrides)


class OracleEngine(EngineCore):
    def __init__(self, settings, calib, threads=1):
        super().__init__(settings, calib, api=load_api())
        if threads != 1:
            self.set_threads(threads)

    def set_threads(self, n):
        self._check(self.api.set_threads(self._h, int(n)))


def oracle_driver(settings, calib, voxel_decay_params=None, use_depth_weighting=False):
    """An InfiniTamDriver mirror running on the oracle."""
    return InfiniTamDriver(settings, calib, voxel_decay_params, use_depth_weighting, api=load_api())
