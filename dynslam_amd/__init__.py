"""dynslam_amd — MI355X-native voxel-hashed TSDF integrate + raycast engines for DynSLAM.

Only what the hot path needs: `csrc/` (HIP kernels + the C ABI of include/dsr.h),
`engine` (host-side mirror of the reference's InfiniTamDriver boundary), `synth`
(deterministic KITTI-like input generator used by tests and bench) and
`multigpu` (one-volume-per-GPU sharding + RCCL composite).
"""
__version__ = "0.1.0"
