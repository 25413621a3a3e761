// VBTM container: on-disk records, reader and validator.  Pure C++ (no HIP): the library (detector.hip) parses model files through
// this header, and tests/fuzz/parse_fuzz.cc builds the same code with gcc -fsanitize=address,undefined and feeds it truncated and
// bit-flipped files - a malformed container must come back as an error text, never as a crash, an exception across the C ABI or an
// index the planner / the kernels would follow out of bounds.  (vbt_amd/container.py is the writer.)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace vbt {

enum { OP_STEM = 1, OP_PW = 2, OP_DW = 3, OP_ADD = 4, OP_MAXPOOL = 5, OP_RESIZE_NN = 6, OP_POSTPROCESS = 7 };

#pragma pack(push, 1)
struct Header {
  char magic[8];
  int32_t arch, image_size, num_tensors, num_ops, num_anchors, max_detections;
  float nms_iou_threshold, nms_score_threshold;
  int64_t blob_offset, blob_bytes;
  int32_t input_tensor;
  int32_t num_classes;   // class columns per anchor of the head's class tensors (0 in files written before the field existed = 1)
  int32_t reserved[16];
};
struct TensorRec {
  int32_t h, w, c, zero_point;
  float scale;
  int32_t pad[3];
};
struct OpRec {
  int32_t type, n_inputs, inputs[12], output, k, stride, pad_t, pad_l, act_min, act_max, level;
  int64_t w_off, b_off, m_off, aux_off, aux2_off;
  float in_mult[3];      // ADD: s_a/s_out, s_b/s_out (informational)
  int32_t add_q[4];      // ADD: bias, a_multiplier, b_multiplier, shift (XNNPACK qs8-vadd)
  int32_t reserved[1];
};
#pragma pack(pop)
static_assert(sizeof(Header) == 128, "header");
static_assert(sizeof(TensorRec) == 32, "tensor");
static_assert(sizeof(OpRec) == 160, "op");
#define VBT_CONTAINER_MAGIC "VBTM0002"
#define VBT_POST_TABLE_BYTES (256 * 4 * 2 + 256 * 8 * 2 + 16)   // score f32 | box f32 | dq f64 | ex f64 | scales f32[4]

struct ContainerData {
  Header hdr;
  std::vector<TensorRec> tensors;
  std::vector<OpRec> ops;
  std::vector<unsigned char> blob;
};

namespace detail {
inline std::string fmt(const char* f, long a = 0, long b = 0, long c = 0, long d = 0) {
  char buf[256];
  snprintf(buf, sizeof(buf), f, a, b, c, d);
  return buf;
}
// [off, off + bytes) inside a blob of `total` bytes, `align`-aligned
inline bool span_ok(int64_t off, int64_t bytes, int64_t total, int align) {
  return off >= 0 && bytes >= 0 && off <= total && bytes <= total - off && (off % align) == 0;
}
inline int same_out(int in, int stride) { return (in + stride - 1) / stride; }
}  // namespace detail

// Structural validation: every index the planner and the kernels follow is in range, every blob span lies inside the blob, the shapes
// of each op relate the way its kernel assumes (SAME padding, channel counts), tensors are written before they are read.
inline bool validate_container(const ContainerData& c, std::string* err) {
  using detail::fmt;
  using detail::span_ok;
  const Header& h = c.hdr;
  const int nt = (int)c.tensors.size(), no = (int)c.ops.size();
  const int64_t nb = (int64_t)c.blob.size();
  auto fail = [&](const std::string& s) { *err = s; return false; };
  if (h.image_size < 32 || h.image_size > 4096) return fail(fmt("image size %ld outside 32..4096", h.image_size));
  if (h.num_anchors < 1 || h.num_anchors > 65535) return fail(fmt("%ld anchors (1..65535 supported)", h.num_anchors));
  if (h.max_detections < 1 || h.max_detections > 25) return fail(fmt("max_detections %ld outside 1..25", h.max_detections));
  if (h.num_classes < 0 || h.num_classes > 4) return fail(fmt("%ld class columns per anchor (1..4 supported)", h.num_classes));
  if (!(h.nms_iou_threshold >= 0.0f && h.nms_iou_threshold <= 1.0f) || !std::isfinite(h.nms_score_threshold)) return fail("NMS thresholds are not finite / in range");
  if (h.input_tensor < 0 || h.input_tensor >= nt) return fail(fmt("input tensor %ld of %ld", h.input_tensor, nt));
  for (int i = 0; i < nt; i++) {
    const TensorRec& t = c.tensors[i];
    if (t.h < 1 || t.w < 1 || t.c < 1 || t.h > 8192 || t.w > 8192 || t.c > 8192 || (int64_t)t.h * t.w * t.c > (1ll << 28)) return fail(fmt("tensor %ld: shape %ld x %ld x %ld", i, t.h, t.w, t.c));
    if (t.zero_point < -128 || t.zero_point > 127) return fail(fmt("tensor %ld: zero point %ld", i, t.zero_point));
    if (!(t.scale > 0.0f) || !std::isfinite(t.scale)) return fail(fmt("tensor %ld: scale is not a positive finite number", i));
  }
  const TensorRec& in = c.tensors[h.input_tensor];
  if (in.h != h.image_size || in.w != h.image_size || in.c != 3) return fail("the input tensor is not image_size x image_size x 3");
  std::vector<char> written((size_t)nt, 0);
  written[(size_t)h.input_tensor] = 1;
  int n_post = 0;
  for (int i = 0; i < no; i++) {
    const OpRec& op = c.ops[i];
    if (op.type < OP_STEM || op.type > OP_POSTPROCESS) return fail(fmt("op %ld: unknown type %ld", i, op.type));
    if (op.n_inputs < 1 || op.n_inputs > 12) return fail(fmt("op %ld: %ld inputs", i, op.n_inputs));
    for (int k = 0; k < op.n_inputs; k++) {
      if (op.inputs[k] < 0 || op.inputs[k] >= nt) return fail(fmt("op %ld: input %ld is tensor %ld of %ld", i, k, op.inputs[k], nt));
      if (!written[(size_t)op.inputs[k]]) return fail(fmt("op %ld reads tensor %ld before any op has written it", i, op.inputs[k]));
    }
    if (op.output < 0 || op.output >= nt || op.output == h.input_tensor) return fail(fmt("op %ld: output tensor %ld", i, op.output));
    if (written[(size_t)op.output]) return fail(fmt("op %ld: tensor %ld is written twice", i, op.output));
    for (int k = 0; k < op.n_inputs; k++)
      if (op.inputs[k] == op.output) return fail(fmt("op %ld writes its own input", i));
    if (op.act_min < -128 || op.act_max > 127 || op.act_min > op.act_max) return fail(fmt("op %ld: activation range %ld..%ld", i, op.act_min, op.act_max));
    const TensorRec& ti = c.tensors[op.inputs[0]];
    const TensorRec& to = c.tensors[op.output];
    switch (op.type) {
      case OP_STEM:
      case OP_PW:
      case OP_DW: {
        if (op.n_inputs != 1) return fail(fmt("op %ld: a conv has one input", i));
        const bool stem = op.type == OP_STEM, dw = op.type == OP_DW;
        const int k = op.type == OP_PW ? 1 : op.k, s = op.type == OP_PW ? 1 : op.stride;
        if (stem && (k != 3 || s != 2 || ti.c != 3)) return fail(fmt("op %ld: stem conv must be 3x3 stride 2 on 3 channels", i));
        if (dw && !((k == 3 || k == 5) && (s == 1 || s == 2) && to.c == ti.c)) return fail(fmt("op %ld: depthwise conv k %ld stride %ld", i, k, s));
        if (to.h != detail::same_out(ti.h, s) || to.w != detail::same_out(ti.w, s)) return fail(fmt("op %ld: output %ld x %ld is not SAME-padded stride %ld", i, to.h, to.w, s));
        if (op.type != OP_PW) {   // TF SAME padding: total = max((out - 1) s + k - in, 0), the smaller half first
          const int pt = std::max((to.h - 1) * s + k - ti.h, 0) / 2, pl = std::max((to.w - 1) * s + k - ti.w, 0) / 2;
          if (op.pad_t != pt || op.pad_l != pl) return fail(fmt("op %ld: padding %ld,%ld is not TF SAME", i, op.pad_t, op.pad_l));
        }
        const int64_t wbytes = dw ? (int64_t)k * k * to.c : (int64_t)k * k * ti.c * to.c;
        if (!span_ok(op.w_off, wbytes, nb, 1) || !span_ok(op.b_off, 4ll * to.c, nb, 4) || !span_ok(op.m_off, 4ll * to.c, nb, 4))
          return fail(fmt("op %ld: weights / bias / multipliers lie outside the blob", i));
        break;
      }
      case OP_ADD:
        if (op.n_inputs != 2) return fail(fmt("op %ld: ADD has two inputs", i));
        for (int k = 0; k < 2; k++) {
          const TensorRec& t = c.tensors[op.inputs[k]];
          if (t.h != to.h || t.w != to.w || t.c != to.c) return fail(fmt("op %ld: ADD operands differ in shape", i));
        }
        break;
      case OP_MAXPOOL:
        if (op.n_inputs != 1 || op.k != 3 || op.stride != 2 || to.c != ti.c || to.h != detail::same_out(ti.h, 2) || to.w != detail::same_out(ti.w, 2))
          return fail(fmt("op %ld: max pool must be 3x3 stride 2 SAME", i));
        break;
      case OP_RESIZE_NN:
        if (op.n_inputs != 1 || to.c != ti.c) return fail(fmt("op %ld: resize keeps the channel count", i));
        break;
      case OP_POSTPROCESS: {
        n_post++;
        if (i != no - 1) return fail("the post-process op must be the last op");
        if (op.n_inputs != 10) return fail(fmt("op %ld: post-process takes 5 class + 5 box tensors", i));
        const int nc = h.num_classes > 0 ? h.num_classes : 1;
        int64_t anchors = 0;
        for (int l = 0; l < 5; l++) {
          const TensorRec &tc = c.tensors[op.inputs[l]], &tb = c.tensors[op.inputs[5 + l]];
          if (tc.c % nc != 0 || tb.c != 4 * (tc.c / nc) || tc.h != tb.h || tc.w != tb.w) return fail(fmt("op %ld: level %ld class / box head shapes disagree", i, l));
          anchors += (int64_t)tc.h * tc.w * (tc.c / nc);
        }
        if (anchors != h.num_anchors) return fail(fmt("the head tensors hold %ld anchors, the header says %ld", anchors, h.num_anchors));
        if (!span_ok(op.aux_off, 16ll * h.num_anchors, nb, 4) || !span_ok(op.aux2_off, VBT_POST_TABLE_BYTES, nb, 8)) return fail("anchors / post-process tables lie outside the blob");
        break;
      }
    }
    written[(size_t)op.output] = 1;
  }
  if (n_post != 1) return fail("a model needs exactly one post-process op");
  return true;
}

// Reads and validates a container file.  false + *err on any problem (I/O, magic, sizes, structure).
inline bool read_container(const char* path, ContainerData* out, std::string* err) {
  using detail::fmt;
  FILE* f = fopen(path, "rb");
  if (!f) { *err = std::string("cannot open model container '") + path + "'"; return false; }
  auto fail = [&](const std::string& s) { fclose(f); *err = std::string("'") + path + "' is not a valid " VBT_CONTAINER_MAGIC " container: " + s; return false; };
  if (fseek(f, 0, SEEK_END) != 0) return fail("not seekable");
  const long fsize = ftell(f);
  if (fsize < (long)sizeof(Header) || fseek(f, 0, SEEK_SET) != 0) return fail("shorter than a header");
  Header& h = out->hdr;
  if (fread(&h, sizeof(Header), 1, f) != 1) return fail("header unreadable");
  if (memcmp(h.magic, VBT_CONTAINER_MAGIC, 8) != 0) return fail("bad magic (older containers: regenerate with tools/make_model.py)");
  if (h.num_tensors < 2 || h.num_tensors > 100000 || h.num_ops < 1 || h.num_ops > 100000) return fail(fmt("%ld tensors / %ld ops", h.num_tensors, h.num_ops));
  const int64_t recs = (int64_t)sizeof(Header) + (int64_t)h.num_tensors * (int64_t)sizeof(TensorRec) + (int64_t)h.num_ops * (int64_t)sizeof(OpRec);
  if (recs > fsize) return fail("tensor / op tables run past the end of the file");
  if (h.blob_offset < recs || h.blob_offset > fsize || h.blob_bytes < 0 || h.blob_bytes > fsize - h.blob_offset || h.blob_bytes > (1ll << 31))
    return fail(fmt("blob [%ld, +%ld) outside the file of %ld bytes", (long)h.blob_offset, (long)h.blob_bytes, fsize));
  out->tensors.resize((size_t)h.num_tensors);
  out->ops.resize((size_t)h.num_ops);
  out->blob.resize((size_t)h.blob_bytes);
  const bool ok = fread(out->tensors.data(), sizeof(TensorRec), out->tensors.size(), f) == out->tensors.size() &&
                  fread(out->ops.data(), sizeof(OpRec), out->ops.size(), f) == out->ops.size() && fseek(f, (long)h.blob_offset, SEEK_SET) == 0 &&
                  (out->blob.empty() || fread(out->blob.data(), 1, out->blob.size(), f) == out->blob.size());
  if (!ok) return fail("truncated");
  fclose(f);
  f = nullptr;
  std::string why;
  if (!validate_container(*out, &why)) { *err = std::string("'") + path + "' is not a valid " VBT_CONTAINER_MAGIC " container: " + why; return false; }
  return true;
}

// ---- plan files (VBT_PLAN_FILE): "VBTPLAN2 <groups>", then per group "<chosen> <steps> <family>:<variant> ..." (format 1: bare integers).
// The shape of the plan the library built (alternatives per group, steps per alternative, the family of every step and the variants the
// planner offers for it) is the only thing a file can select from: anything else refuses the file (the caller re-tunes).
struct PlanStepShape {
  std::string family;
  std::vector<int> variants;   // legal variant numbers of this step (empty: any value in -1..100000 is passed on to the launcher's own checks)
};
struct PlanShape {
  std::vector<std::vector<std::vector<PlanStepShape>>> groups;   // [group][alternative][step]
};
struct PlanChoice {
  int chosen = 0;
  std::vector<int> variants;
};
inline bool parse_plan_file(const char* path, const PlanShape& shape, std::vector<PlanChoice>* out, std::string* note) {
  FILE* f = fopen(path, "r");
  if (!f) { *note = "no such file"; return false; }
  auto fail = [&](const std::string& s) { fclose(f); *note = s; return false; };
  char head[32] = "";
  int ng = 0;
  bool v2 = false;
  if (fscanf(f, "%31s", head) != 1) return fail("empty file");
  if (strcmp(head, "VBTPLAN2") == 0) {
    v2 = true;
    if (fscanf(f, "%d", &ng) != 1) return fail("no group count");
  } else {
    char* end = nullptr;
    const long v = strtol(head, &end, 10);
    if (!end || *end != 0 || v < 0 || v > 1000000) return fail("neither VBTPLAN2 nor a format-1 group count");
    ng = (int)v;
  }
  if (ng != (int)shape.groups.size()) return fail(detail::fmt("%ld groups in the file, %ld in this library's plan", ng, (long)shape.groups.size()));
  out->clear();
  for (int gi = 0; gi < ng; gi++) {
    int ch = 0, ns = 0;
    if (fscanf(f, "%d %d", &ch, &ns) != 2) return fail(detail::fmt("group %ld: truncated", gi));
    if (ch < 0 || ch >= (int)shape.groups[(size_t)gi].size()) return fail(detail::fmt("group %ld: alternative %ld does not exist", gi, ch));
    const std::vector<PlanStepShape>& steps = shape.groups[(size_t)gi][(size_t)ch];
    if (ns != (int)steps.size()) return fail(detail::fmt("group %ld: %ld steps in the file, %ld in the plan", gi, ns, (long)steps.size()));
    PlanChoice pc;
    pc.chosen = ch;
    for (int i = 0; i < ns; i++) {
      long v = 0;
      if (v2) {
        char tok[96] = "";
        if (fscanf(f, "%95s", tok) != 1) return fail(detail::fmt("group %ld step %ld: truncated", gi, i));
        char* colon = strrchr(tok, ':');
        if (!colon) return fail(detail::fmt("group %ld step %ld: no ':'", gi, i));
        *colon = 0;
        if (steps[(size_t)i].family != tok)
          return fail(detail::fmt("group %ld step %ld is '", gi, i) + tok + "' in the file, '" + steps[(size_t)i].family + "' in this library");
        char* end = nullptr;
        v = strtol(colon + 1, &end, 10);
        if (end == colon + 1 || *end != 0) return fail(detail::fmt("group %ld step %ld: variant is not a number", gi, i));
      } else {
        int vi = 0;
        if (fscanf(f, "%d", &vi) != 1) return fail(detail::fmt("group %ld step %ld: truncated", gi, i));
        v = vi;
      }
      const std::vector<int>& legal = steps[(size_t)i].variants;
      bool ok = v >= -1 && v <= 100000;
      if (ok && !legal.empty()) {
        ok = false;
        for (int l : legal) ok = ok || l == (int)v;
      }
      if (!ok) return fail(detail::fmt("group %ld step %ld: variant %ld is not one the planner offers for this step", gi, i, v));
      pc.variants.push_back((int)v);
    }
    out->push_back(pc);
  }
  fclose(f);
  return true;
}

}  // namespace vbt
