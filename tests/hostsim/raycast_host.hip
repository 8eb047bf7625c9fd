// Host stand-in for ONE lane of k_raycast: the device function cast_ray<Ops> (dynslam_amd/csrc/k_raycast.h) compiled for
// the CPU with a one-ray Ops, so that tests/test_raycast_host.py can check the march — its lookup rounds, its look-ahead slot,
// its use of the block map, the trilinear reads — against the oracle's raycast WITHOUT a GPU.
// Test infrastructure: built by the test with hipcc (host code only is run), never part of libdsr_hip.so.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../dynslam_amd/csrc/k_raycast.h"

namespace {
struct HostOps {
  static int f2i(float f) {  // v_cvt_i32_f32: toward zero, saturating, NaN -> 0
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
  }
  static bool any(bool b) { return b; }
  static float sqrt(float f) { return sqrtf(f); }
  static float floor(float f) { return floorf(f); }
};
}  // namespace

// table: noTotalEntries entries; vba: noBlocks * 4096 bytes in the library's block layout (only the sdf plane is read);
// occ_entries: 0 = no block map, else a power of two: the map is built here from the table the way the kernels maintain it
extern "C" int rr_cast_all(const float *invM, const float *proj, float voxelSize, float mu, int W, int H, int noBuckets, int noTotalEntries,
                           const dsr_hash_entry *table, const unsigned char *vba, const float *minmax, int occ_entries, float *out,
                           long long *occ_stats) {
  using namespace dsr;
  FrameP p;
  std::memset(&p, 0, sizeof p);
  std::memcpy(p.invM.m, invM, sizeof p.invM.m);
  p.proj = make_float4(proj[0], proj[1], proj[2], proj[3]);
  p.voxelSize = voxelSize; p.mu = mu; p.W = W; p.H = H;
  p.noBuckets = noBuckets; p.noTotalEntries = noTotalEntries; p.hashMask = (uint32_t)noBuckets - 1u;
  SceneP s;
  std::memset(&s, 0, sizeof s);
  s.table = const_cast<dsr_hash_entry *>(table);
  s.vba = const_cast<uint8_t *>(vba);
  std::vector<uint4> occ;
  if (occ_entries > 0) {
    occ.assign((size_t)occ_entries, make_uint4(0u, 0u, 0u, 0u));
    s.occ = occ.data(); s.occMask = (uint32_t)occ_entries - 1u;
    long long conflicts = 0, used = 0;
    for (int t = 0; t < noTotalEntries; ++t) {  // occ_set (dsr_device.h), sequentially
      if (table[t].ptr < 0) continue;
      const int bx = table[t].pos[0], by = table[t].pos[1], bz = table[t].pos[2];
      uint4 &e = occ[occ_index(bx, by, bz, s.occMask)];
      const uint32_t kxy = occ_key_xy(bx, by), kz = occ_key_z(bz);
      if (e.x == 0u && e.y == 0u) { e.x = kxy; e.y = kz; ++used; }
      if (e.x == kxy && (e.y & ~kOccConflict) == kz) e.z = (uint32_t)table[t].ptr;
      else { if (!(e.y & kOccConflict)) ++conflicts; e.y |= kOccConflict; }
    }
    if (occ_stats) { occ_stats[0] = used; occ_stats[1] = conflicts; }
  }
  const int mw = (W + kMinmaxSubsample - 1) / kMinmaxSubsample;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const float *mmp = minmax + 2 * ((x >> 3) + (y >> 3) * mw);
      const float4 r = cast_ray<HostOps>(p, s, x, y, make_float2(mmp[0], mmp[1]));
      float *o = out + 4 * ((size_t)x + (size_t)y * W);
      o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
  return 0;
}
