// Shared host-side declarations for libvbt_hip.so (product code; never includes anything from oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vbt_hip.h"
#include "../../include/vbt_hip_diag.h"
#include "container_parse.h"

namespace vbt {

void set_error(const char* fmt, ...);

// roctx ranges around the host-side enqueue of the path's three stages (detect incl. decode + NMS, track, clip close): with
// VBT_ROCTX=1 the library loads librocprofiler-sdk-roctx.so at the first use and `rocprofv3 --marker-trace` shows
// "vbt:detect" / "vbt:track" / "vbt:finish" next to the kernels they enqueue; without it every call is one branch.
struct RoctxRange {
  explicit RoctxRange(const char* name);
  ~RoctxRange();
  bool on;
};

// More than 64 KB of dynamic LDS is an opt-in per kernel AND per device (hipFuncSetAttribute acts on the current device's code object).
// Every launch site keeps its own per-device record; a refusal is an error (VBT_ERR_HIP with the text set), not a silent launch failure
// that surfaces at a later synchronisation or inside a stream capture.
struct LdsOptIn { signed char dev[64] = {0}; };
bool lds_opt_in(const void* fn, LdsOptIn* state);
#define VBT_LDS_OPT_IN(...)                                                                          \
  do {                                                                                               \
    static vbt::LdsOptIn lds_state_;                                                                 \
    if (!vbt::lds_opt_in(reinterpret_cast<const void*>(&__VA_ARGS__), &lds_state_)) return VBT_ERR_HIP; \
  } while (0)

#define VBT_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      vbt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return VBT_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

// preprocess_image (reference odt.py:10-19) on device memory (detector.hip); compact != 0: src holds only the row pairs the resize reads
int resize_frames_dev(const uint8_t* src_dev, int B, int H, int W, uint8_t* dst_dev, int h, int w, int swap_rb, int compact, hipStream_t st);

// ---- VBTM container records, reader + validator, plan-file reader: container_parse.h (pure C++, also built host-only under
// -fsanitize=address,undefined by tests/test_parser_fuzz.py) ----
// Integer parameters of an int8 ADD exactly as XNNPACK derives them (xnn_create_add_nd_qs8 +
// xnn_init_qs8_add_minmax_*_params of the XNNPACK revision tflite-runtime 2.14 builds; the reference runs that kernel
// through the default delegate, reference odt.py:58-61):  q = clamp(((bias + a*am + b*bm) >> shift) + z_out, lo, hi).
struct AddParams { int32_t bias, am, bm, shift; };
inline bool xnn_add_params(float s_a, float s_b, float s_out, int z_a, int z_b, AddParams* p) {
  const float a_os = s_a / s_out, b_os = s_b / s_out;
  if (!(a_os >= 0x1.0p-10f && a_os < 0x1.0p+8f) || !(b_os >= 0x1.0p-10f && b_os < 0x1.0p+8f)) return false;
  const float mx = a_os > b_os ? a_os : b_os;
  uint32_t bits;
  memcpy(&bits, &mx, 4);
  const int32_t max_exp = (int32_t)(bits >> 23) - 127;
  const uint32_t shift = (uint32_t)(20 - max_exp);
  if (shift < 12 || shift > 30) return false;
  auto scaled = [&](float v) {  // v * 2^shift through the exponent field, then lrintf (round to nearest even)
    uint32_t b;
    memcpy(&b, &v, 4);
    b += shift << 23;
    float f;
    memcpy(&f, &b, 4);
    return (int32_t)lrintf(f);
  };
  p->am = scaled(a_os);
  p->bm = scaled(b_os);
  p->shift = (int32_t)shift;
  p->bias = (int32_t)((1u << (shift - 1)) - (uint32_t)(p->am * z_a) - (uint32_t)(p->bm * z_b));
  return true;
}

}  // namespace vbt
