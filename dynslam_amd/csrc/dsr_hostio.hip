// dsr_hostio.hip — host-side I/O at the boundary: the precomputed depth / disparity maps DynSLAM reads from disk
// (PrecomputedDepthProvider.cpp:22-75: OpenCV FileStorage XML with int16 millimetres, pfmLib .pfm with float disparities) and the
// page-locking of the host's persistent frame / preview buffers.  No kernels here: parsing files is not GPU work.
#include <sched.h>

#include <cctype>

#include "dsr_internal.h"

namespace {
std::mutex g_pinMutex;
std::map<uintptr_t, size_t> g_pinned;           // host ranges the caller page-locked through dsr_pin_host_buffer
}  // namespace
bool dsr_internal::host_range_pinned(const void *p, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_pinMutex);
  if (g_pinned.empty()) return false;
  auto it = g_pinned.upper_bound((uintptr_t)p);
  if (it == g_pinned.begin()) return false;
  --it;
  return (uintptr_t)p + bytes <= it->first + it->second;
}
extern "C" {

int dsr_pin_host_buffer(void *ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(DSR_E_ARG, "null buffer");
  if (host_range_pinned(ptr, bytes)) return DSR_OK;
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  std::lock_guard<std::mutex> lock(g_pinMutex);
  g_pinned[(uintptr_t)ptr] = bytes;
  return DSR_OK;
}
int dsr_unpin_host_buffer(void *ptr) {
  if (!ptr) return fail(DSR_E_ARG, "null buffer");
  {
    std::lock_guard<std::mutex> lock(g_pinMutex);
    auto it = g_pinned.find((uintptr_t)ptr);
    if (it == g_pinned.end()) return fail(DSR_E_ARG, "not a buffer pinned through dsr_pin_host_buffer");
    g_pinned.erase(it);
  }
  HIP_TRY(hipHostUnregister(ptr));
  return DSR_OK;
}

// ---- precomputed depth / disparity maps on disk (PrecomputedDepthProvider.cpp:22-75) -------------------
// the text between <tag ...> and </tag> of the first such element at or after `from` (FileStorage XML is
// flat enough for this: the node "depth-frame" holds <rows>, <cols>, <dt>, <data>)
static bool xml_element(const std::string &doc, const char *tag, size_t from, size_t *begin, size_t *end) {
  const std::string open = std::string("<") + tag;
  size_t p0 = doc.find(open, from);
  while (p0 != std::string::npos) {
    const char c = p0 + open.size() < doc.size() ? doc[p0 + open.size()] : 0;
    if (c == '>' || c == ' ' || c == '\t' || c == '\n' || c == '\r') break;
    p0 = doc.find(open, p0 + 1);
  }
  if (p0 == std::string::npos) return false;
  const size_t gt = doc.find('>', p0);
  if (gt == std::string::npos) return false;
  const size_t close = doc.find(std::string("</") + tag + ">", gt);
  if (close == std::string::npos) return false;
  *begin = gt + 1; *end = close;
  return true;
}
static bool read_whole_file(const char *path, std::string *out) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}

// The calling thread onto the CPUs next to the GPU (the PCI device's local_cpulist): DynSLAM's host thread copies ~7.5 MB of
// frames and previews per frame to and from pinned memory and polls status words the GPU writes — on a two-socket box both are
// cheaper from the GPU's own NUMA node.  Nothing happens (DSR_OK) where the kernel does not publish the list.
int dsr_pin_host_thread(int device) {
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(DSR_E_DEVICE, "no current HIP device");
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) return fail(DSR_E_DEVICE, "hipDeviceGetPCIBusId failed");
  for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
  std::string list;
  if (!read_whole_file(path.c_str(), &list)) return DSR_OK;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n = 0;
  for (const char *q = list.c_str(); *q;) {  // "0-63,128-191"
    char *end = nullptr;
    const long a = strtol(q, &end, 10);
    if (end == q) break;
    long b2 = a;
    q = end;
    if (*q == '-') { b2 = strtol(q + 1, &end, 10); q = end; }
    for (long c = a; c <= b2 && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++n; }
    if (*q == ',') ++q; else break;
  }
  if (n > 0 && sched_setaffinity(0, sizeof set, &set) != 0) return fail(DSR_E_DEVICE, "sched_setaffinity failed");
  return DSR_OK;
}

static int read_depth_xml_impl(const char *path, int16_t *depth_mm_out, int capacity, int *width, int *height) {
  if (!path || !width || !height) return fail(DSR_E_ARG, "bad arguments");
  std::string doc;
  if (!read_whole_file(path, &doc)) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  size_t nb, ne, b, e2;
  if (!xml_element(doc, "depth-frame", 0, &nb, &ne)) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  const std::string node = doc.substr(nb, ne - nb);
  int rows = 0, cols = 0;
  if (!xml_element(node, "rows", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <rows>");
  rows = atoi(node.substr(b, e2 - b).c_str());
  if (!xml_element(node, "cols", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <cols>");
  cols = atoi(node.substr(b, e2 - b).c_str());
  if (!xml_element(node, "dt", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <dt>");
  std::string dt = node.substr(b, e2 - b);
  dt.erase(std::remove_if(dt.begin(), dt.end(), [](char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }), dt.end());
  if (dt != "s") return fail(DSR_E_IO, "Precomputed depth map had the wrong format.");  // :42-44: CV_16SC1 only
  // a size no camera produces is a malformed file, not something to allocate for (the size query hands it to the caller)
  if ((long long)rows * cols > (1ll << 28) || rows > (1 << 20) || cols > (1 << 20)) return fail(DSR_E_IO, "depth-frame: implausible rows x cols");
  *width = cols; *height = rows;
  if (rows <= 0 || cols <= 0) return fail(DSR_E_IO, "Could not read precomputed depth map: empty matrix");
  if (!depth_mm_out || (long long)rows * cols > capacity) return fail(DSR_E_ARG, "depth map larger than the buffer");
  if (!xml_element(node, "data", 0, &b, &e2)) return fail(DSR_E_IO, "depth-frame without <data>");
  const char *p = node.c_str() + b, *end = node.c_str() + e2;
  const long long n = (long long)rows * cols;
  long long i = 0;
  while (i < n) {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
    if (p >= end) break;
    char *next = nullptr;
    const long v = strtol(p, &next, 10);
    if (next == p) return fail(DSR_E_IO, "malformed <data> in depth-frame");
    depth_mm_out[i++] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));  // cv::saturate_cast<short>
    p = next;
  }
  if (i != n) return fail(DSR_E_IO, "depth-frame <data> holds fewer values than rows x cols");
  return DSR_OK;
}

static int read_pfm_impl(const char *path, float *out, int capacity, int *width, int *height) {
  if (!path || !width || !height) return fail(DSR_E_ARG, "bad arguments");
  FILE *f = fopen(path, "rb");
  if (!f) return fail(DSR_E_IO, "Could not read precomputed depth map.");
  char magic[3] = {0, 0, 0};
  int w = 0, h = 0;
  float scale = 0.0f;
  // "Pf" <ws> width <ws> height <ws> scale <single whitespace byte> raster
  if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "Pf") != 0 || fscanf(f, "%d %d %f", &w, &h, &scale) != 3) {
    fclose(f);
    return fail(DSR_E_IO, "not a single-channel PFM (\"Pf\") file");
  }
  (void)fgetc(f);
  if ((long long)w * h > (1ll << 28) || w > (1 << 20) || h > (1 << 20)) { fclose(f); return fail(DSR_E_IO, "PFM: implausible width x height"); }
  *width = w; *height = h;
  if (w <= 0 || h <= 0) { fclose(f); return fail(DSR_E_IO, "Could not read precomputed depth map: empty image"); }
  if (!out || (long long)w * h > capacity) { fclose(f); return fail(DSR_E_ARG, "PFM image larger than the buffer"); }
  const bool fileLittle = scale < 0.0f;
  const uint16_t probe = 1;
  const bool hostLittle = *reinterpret_cast<const uint8_t *>(&probe) == 1;
  for (int r = h - 1; r >= 0; --r) {  // the file's first row is the image's bottom row
    float *row = out + (size_t)r * w;
    if (fread(row, 4, (size_t)w, f) != (size_t)w) { fclose(f); return fail(DSR_E_IO, "PFM raster shorter than width x height"); }
    if (fileLittle != hostLittle)
      for (int c = 0; c < w; ++c) {
        uint32_t v; memcpy(&v, row + c, 4);
        v = (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
        memcpy(row + c, &v, 4);
      }
  }
  fclose(f);
  return DSR_OK;
}

// The size is reported whenever the header could be read (DSR_OK, and DSR_E_ARG for a buffer that is too small: the
// size query of a caller that allocates afterwards); after DSR_E_IO it is 0 x 0, never a half-parsed value.
int dsr_read_depth_xml(const char *path, int16_t *depth_mm_out, int capacity, int *width, int *height) {
  const int st = read_depth_xml_impl(path, depth_mm_out, capacity, width, height);
  if (st == DSR_E_IO && width && height) *width = *height = 0;
  return st;
}
int dsr_read_pfm(const char *path, float *out, int capacity, int *width, int *height) {
  const int st = read_pfm_impl(path, out, capacity, width, height);
  if (st == DSR_E_IO && width && height) *width = *height = 0;
  return st;
}

}  // extern "C"
