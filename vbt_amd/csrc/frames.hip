// Frame plumbing in front of the detector (gfx950): assembling a detector batch from frames that live in different
// places of device memory.  HBM-bound byte moves: 16 bytes per lane, consecutive lanes on consecutive addresses.
#include <dlfcn.h>

#include "common.h"

namespace vbt {

constexpr int GATHER_FRAMES = 64;
struct GatherMeta {
  const uint8_t* src[GATHER_FRAMES];
};

// grid = (chunks, frames): workgroup (x, y) copies 256 x 16-byte pieces x `per` of frame y
__global__ __launch_bounds__(256) void gather_frames_kernel(uint4* __restrict__ dst, GatherMeta meta, size_t frame_vec, int per) {
  const uint4* __restrict__ s = (const uint4*)meta.src[blockIdx.y];
  uint4* __restrict__ d = dst + (size_t)blockIdx.y * frame_vec;
  size_t i = ((size_t)blockIdx.x * per) * 256 + threadIdx.x;
#pragma unroll 4
  for (int k = 0; k < per; k++, i += 256)
    if (i < frame_vec) d[i] = s[i];
}

// ---- roctx ranges (common.h): resolved lazily with dlopen, so that the library has no link-time dependency on the profiler SDK ----
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static int g_roctx_state = 0;   // 0 = not tried, 1 = on, -1 = off
static bool roctx_ready() {
  if (g_roctx_state == 0) {
    g_roctx_state = -1;
    const char* e = getenv("VBT_ROCTX");
    if (e && e[0] == '1') {
      void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) {
        g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) g_roctx_state = 1;
      }
    }
  }
  return g_roctx_state == 1;
}
RoctxRange::RoctxRange(const char* name) : on(roctx_ready()) { if (on) g_roctx_push(name); }
RoctxRange::~RoctxRange() { if (on) g_roctx_pop(); }

}  // namespace vbt

using namespace vbt;

extern "C" int vbt_gather_frames(uint8_t* dst_dev, const uint8_t* const* src_frames_host, int n_frames, size_t frame_bytes, void* stream) {
  if (!dst_dev || !src_frames_host || n_frames < 1 || frame_bytes < 16 || (frame_bytes & 15) != 0 || ((uintptr_t)dst_dev & 15) != 0) {
    set_error("vbt_gather_frames: bad argument (frame_bytes must be a positive multiple of 16, pointers 16-byte aligned)");
    return VBT_ERR_ARG;
  }
  for (int i = 0; i < n_frames; i++)
    if (!src_frames_host[i] || ((uintptr_t)src_frames_host[i] & 15) != 0) { set_error("vbt_gather_frames: source %d is NULL or misaligned", i); return VBT_ERR_ARG; }
  const size_t vec = frame_bytes / 16;
  const int per = 4;
  const unsigned chunks = (unsigned)((vec + 256 * per - 1) / (256 * per));
  for (int f0 = 0; f0 < n_frames; f0 += GATHER_FRAMES) {
    const int nb = std::min(GATHER_FRAMES, n_frames - f0);
    GatherMeta meta;
    for (int i = 0; i < nb; i++) meta.src[i] = src_frames_host[f0 + i];
    for (int i = nb; i < GATHER_FRAMES; i++) meta.src[i] = nullptr;
    gather_frames_kernel<<<dim3(chunks, (unsigned)nb), 256, 0, (hipStream_t)stream>>>((uint4*)(dst_dev + (size_t)f0 * frame_bytes), meta, vec, per);
  }
  VBT_HIP_CHECK(hipGetLastError());
  return VBT_OK;
}
