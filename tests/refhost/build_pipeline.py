"""Builds tests/refhost/ref_dynslam_host.cpp together with the reference's own, unmodified DynSLAM sources
(compiled from where they lie under /root/reference, never copied) into one executable:

  kind "hip"     links dynslam_amd/csrc/libdsr_hip.so (the product);
  kind "oracle"  the same objects compiled with every dsr_* name redirected to the CPU oracle's orc_* by a
                 generated -include header (TEST-ONLY: the checker leg of tests/test_reference_pipeline.py).

Used by the tests and by __graft_entry__.build_hosts() (which leaves both executables under
tests/refhost/_build/ so that they travel to the GPU box, where /root/reference does not exist)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src/DynSLAM"
HOST = os.path.join(ROOT, "tests", "refhost", "ref_dynslam_host.cpp")
# the reference units of the per-frame pipeline (everything DynSlam::ProcessFrame reaches, GUI excluded)
PIPELINE_UNITS = [
    "DynSlam.cpp", "InfiniTamDriver.cpp", "Utils.cpp", "Input.cpp", "PrecomputedDepthProvider.cpp",
    "InstRecLib/InstanceReconstructor.cpp", "InstRecLib/InstanceTracker.cpp", "InstRecLib/Track.cpp", "InstRecLib/InstanceView.cpp",
    "InstRecLib/InstanceSegmentationResult.cpp", "InstRecLib/SegmentationDataset.cpp", "InstRecLib/SparseSFProvider.cpp",
    "InstRecLib/VisoSparseSFProvider.cpp", "InstRecLib/PrecomputedSegmentationProvider.cpp", "InstRecLib/Utils/BoundingBox.cpp",
    "InstRecLib/Utils/Mask.cpp", "Evaluation/Evaluation.cpp", "Evaluation/CsvWriter.cpp", "Evaluation/Tracklets.cpp",
    "Evaluation/VelodyneIO.cpp", "Evaluation/EvaluationCallback.cpp", "Evaluation/SegmentedCallback.cpp",
    "Evaluation/SegmentedEvaluationCallback.cpp",
]
INC = ["-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "tests", "stubs", "DynSLAM"),
       "-I", os.path.join(ROOT, "tests", "stubs", "DynSLAM", "InstRecLib"),
       "-I", os.path.join(ROOT, "shim", "DynSLAM"), "-I", os.path.join(ROOT, "shim", "DynSLAM", "InstRecLib"),
       "-I", REF, "-I", os.path.join(REF, "InstRecLib")]


def have_reference():
    return os.path.isdir(REF)


def inputs():
    return [HOST, os.path.join(ROOT, "shim", "ITMLib.h"), os.path.join(ROOT, "include", "dsr.h")]


def build(kind, exe, workdir, extra_flags=(), oracle_dir=None):
    """-> exe.  `workdir` receives the object files (and the rename header for kind "oracle").  extra_flags: e.g. sanitizer
    options (compile and link); oracle_dir: where liboracle.so is taken from (a sanitised build for those)."""
    assert kind in ("hip", "oracle")
    os.makedirs(workdir, exist_ok=True)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    flags = ["-std=c++14", "-O1", "-DNDEBUG"] + list(extra_flags)
    if kind == "oracle":
        import sys
        sys.path.insert(0, ROOT)
        from dynslam_amd import _capi
        rename = os.path.join(workdir, "dsr_to_orc.h")
        with open(rename, "w") as f:
            f.write("".join(f"#define dsr_{name} orc_{name}\n" for name in _capi.SIGNATURES))
        flags += ["-include", rename]
        lib_dir = oracle_dir or os.path.join(ROOT, "oracle")
        link = ["-L", lib_dir, "-loracle", f"-Wl,-rpath,{lib_dir}", "-fopenmp"] + list(extra_flags)
    else:
        lib_dir = os.path.join(ROOT, "dynslam_amd", "csrc")
        link = ["-L", lib_dir, "-ldsr_hip", f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    jobs = [(os.path.join(REF, u), os.path.join(workdir, u.replace("/", "_") + ".o")) for u in PIPELINE_UNITS]
    jobs.append((HOST, os.path.join(workdir, "ref_dynslam_host.o")))

    def cc(job):
        r = subprocess.run(["g++"] + flags + INC + ["-c", job[0], "-o", job[1]], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{job[0]}:\n{r.stderr[-3000:]}")
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(cc, jobs))
    subprocess.check_call(["g++"] + [o for _, o in jobs] + ["-o", exe, "-lpthread"] + link)
    return exe
