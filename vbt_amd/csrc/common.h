// Shared host-side declarations for libvbt_hip.so (product code; never includes anything from oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vbt_hip.h"

namespace vbt {

void set_error(const char* fmt, ...);

#define VBT_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      vbt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return VBT_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

// ---- VBTM container records (vbt_amd/container.py is the writer) ----
enum { OP_STEM = 1, OP_PW = 2, OP_DW = 3, OP_ADD = 4, OP_MAXPOOL = 5, OP_RESIZE_NN = 6, OP_POSTPROCESS = 7 };

#pragma pack(push, 1)
struct Header {
  char magic[8];
  int32_t arch, image_size, num_tensors, num_ops, num_anchors, max_detections;
  float nms_iou_threshold, nms_score_threshold;
  int64_t blob_offset, blob_bytes;
  int32_t input_tensor, reserved[17];
};
struct TensorRec {
  int32_t h, w, c, zero_point;
  float scale;
  int32_t pad[3];
};
struct OpRec {
  int32_t type, n_inputs, inputs[12], output, k, stride, pad_t, pad_l, act_min, act_max, level;
  int64_t w_off, b_off, m_off, aux_off, aux2_off;
  float in_mult[3];
  int32_t reserved[5];
};
#pragma pack(pop)
static_assert(sizeof(Header) == 128, "header");
static_assert(sizeof(TensorRec) == 32, "tensor");
static_assert(sizeof(OpRec) == 160, "op");

}  // namespace vbt
