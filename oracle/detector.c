/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * CPU restatement of the detector half of the reference hot path:
 *   reference odt.py:58-66  (signature_fn(images=uint8[1,H,W,3]) -> output_0 count, output_1 scores,
 *                            output_2 classes, output_3 boxes[25,4] ymin,xmin,ymax,xmax)
 * The arithmetic itself lives in an un-vendored dependency, tflite-runtime==2.14.0
 * (reference requirements.txt:381) running missing model files (.MISSING_LARGE_BLOBS), so this
 * file restates the published algorithms [EXTERNAL, SURVEY.md section 8c]:
 * tflite-runtime 2.14 on x86-64 runs the graph through the XNNPACK delegate (applied by default, signed
 * 8-bit operators enabled); ops the delegate does not take run on the TFLite builtin kernels.  Per op:
 *   - CONV_2D / DEPTHWISE_CONV_2D (XNNPACK qs8-qc8w, fp32 requantisation): int8 x int8 -> int32,
 *     acc = sum (x_q - z_x) * w_q + bias_q ; q = clamp(rne(float(acc) * M[c]) + z_y) with
 *     M[c] = (s_x * s_w[c]) / s_y in float32;
 *   - ADD (XNNPACK qs8-vadd-minmax, integer): multipliers lrintf(s_in/s_out * 2^shift) with the larger
 *     one in [2^20, 2^21), q = clamp(((bias + a*a_mult + b*b_mult) >> shift) + z_y); the 3-input BiFPN
 *     sums of the Keras graph are two chained binary ADDs with an intermediate quantisation;
 *   - MAX_POOL_2D 3x3/2 SAME (XNNPACK s8-maxpool, exact);
 *   - RESIZE_NEAREST_NEIGHBOR (TFLite builtin, align_corners = half_pixel_centers = false):
 *     src = min(floor(dst * (float)in / out), in - 1), the product evaluated in float32;
 *   - LOGISTIC (XNNPACK x8-lut): table lrintf(256 / (1 + expf(-s_x (i - z_x)))) - 128 in float32;
 *   - DEQUANTIZE: one float32 multiplication s * (q - z);
 *   - TFLite_Detection_PostProcess (TFLite builtin detection_postprocess.cc, NonMaxSuppressionMultiClassFastHelper with
 *     max_classes_per_detection = 1: the best class column of every anchor, then single-class NMS on those scores):
 *     centre-size decode evaluated in double and rounded to float once per quantity, score filter
 *     (>=), stable descending sort, greedy IoU (> threshold) suppression, top-25.
 * PARITY UNPINNED for this file: the reference holds no model, no input frame and no golden
 * tensor for the detector; structural pins only (MAC counts, 25 detections,
 * k/256 score lattice) - see tests/test_spec.py, tests/test_oracle_detector.py.
 *
 * Every float operation below is written so that it is reproducible bit-for-bit on the GPU:
 * single IEEE mul/add/fma/div per statement, round-to-nearest-even, no contraction
 * (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { OP_STEM = 1, OP_PW = 2, OP_DW = 3, OP_ADD = 4, OP_MAXPOOL = 5, OP_RESIZE_NN = 6, OP_POSTPROCESS = 7 };

#pragma pack(push, 1)
typedef struct {
  char magic[8];
  int32_t arch, image_size, num_tensors, num_ops, num_anchors, max_detections;
  float nms_iou_threshold, nms_score_threshold;
  int64_t blob_offset, blob_bytes;
  int32_t input_tensor;
  int32_t num_classes; /* class columns per anchor of the head's class tensors (0 in older files = 1) */
  int32_t reserved[16];
} hdr_t;
typedef struct { int32_t h, w, c, zero_point; float scale; int32_t pad[3]; } tens_t;
typedef struct {
  int32_t type, n_inputs, inputs[12], output, k, stride, pad_t, pad_l, act_min, act_max, level;
  int64_t w_off, b_off, m_off, aux_off, aux2_off;
  float in_mult[3];
  int32_t add_q[4]; /* bias, a_multiplier, b_multiplier, shift as written by the model tool (checked, not trusted) */
  int32_t reserved[1];
} op_t;
#pragma pack(pop)

typedef struct vbto_model {
  hdr_t hdr;
  tens_t* tensors;
  op_t* ops;
  uint8_t* blob;
  int8_t** data; /* one buffer per tensor, single frame */
} vbto_model;

vbto_model* vbto_load(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  vbto_model* m = (vbto_model*)calloc(1, sizeof(*m));
  if (fread(&m->hdr, sizeof(hdr_t), 1, f) != 1 || memcmp(m->hdr.magic, "VBTM0002", 8) != 0) { fclose(f); free(m); return NULL; }
  int nt = m->hdr.num_tensors, no = m->hdr.num_ops;
  m->tensors = (tens_t*)malloc(sizeof(tens_t) * nt);
  m->ops = (op_t*)malloc(sizeof(op_t) * no);
  m->blob = (uint8_t*)malloc(m->hdr.blob_bytes);
  int ok = fread(m->tensors, sizeof(tens_t), nt, f) == (size_t)nt && fread(m->ops, sizeof(op_t), no, f) == (size_t)no;
  fseek(f, (long)m->hdr.blob_offset, SEEK_SET);
  ok = ok && fread(m->blob, 1, m->hdr.blob_bytes, f) == (size_t)m->hdr.blob_bytes;
  fclose(f);
  if (!ok) { free(m->tensors); free(m->ops); free(m->blob); free(m); return NULL; }
  m->data = (int8_t**)calloc(nt, sizeof(int8_t*));
  for (int i = 0; i < nt; i++) {
    tens_t* t = &m->tensors[i];
    m->data[i] = (int8_t*)calloc((size_t)t->h * t->w * t->c + 64, 1);
  }
  return m;
}

void vbto_free(vbto_model* m) {
  if (!m) return;
  for (int i = 0; i < m->hdr.num_tensors; i++) free(m->data[i]);
  free(m->data); free(m->tensors); free(m->ops); free(m->blob); free(m);
}

int vbto_num_tensors(const vbto_model* m) { return m->hdr.num_tensors; }
int vbto_num_ops(const vbto_model* m) { return m->hdr.num_ops; }
int vbto_image_size(const vbto_model* m) { return m->hdr.image_size; }
int vbto_tensor_shape(const vbto_model* m, int id, int* shape) {
  if (id < 0 || id >= m->hdr.num_tensors) return -1;
  shape[0] = m->tensors[id].h; shape[1] = m->tensors[id].w; shape[2] = m->tensors[id].c;
  return 0;
}
const int8_t* vbto_tensor_data(const vbto_model* m, int id) { return m->data[id]; }

/* XNNPACK-style fp32 requantisation. */
static inline int8_t requant(int32_t acc, float mult, int32_t zp, int32_t lo, int32_t hi) {
  float t = (float)acc * mult;
  t = fminf(fmaxf(t, -65536.0f), 65536.0f);
  int32_t q = (int32_t)nearbyintf(t) + zp; /* round-to-nearest-even */
  if (q < lo) q = lo;
  if (q > hi) q = hi;
  return (int8_t)q;
}

static void run_stem(vbto_model* m, const op_t* op, const uint8_t* frame) {
  const tens_t* ti = &m->tensors[op->inputs[0]];
  const tens_t* to = &m->tensors[op->output];
  const int8_t* w = (const int8_t*)(m->blob + op->w_off);
  const int32_t* b = (const int32_t*)(m->blob + op->b_off);
  const float* mu = (const float*)(m->blob + op->m_off);
  int8_t* out = m->data[op->output];
  int zx = ti->zero_point, k = op->k, ci = ti->c;
  for (int oy = 0; oy < to->h; oy++)
    for (int ox = 0; ox < to->w; ox++)
      for (int co = 0; co < to->c; co++) {
        int32_t acc = b[co];
        for (int ky = 0; ky < k; ky++) {
          int iy = oy * op->stride + ky - op->pad_t;
          if (iy < 0 || iy >= ti->h) continue; /* zero padding in real space: (z_x - z_x) * w = 0 */
          for (int kx = 0; kx < k; kx++) {
            int ix = ox * op->stride + kx - op->pad_l;
            if (ix < 0 || ix >= ti->w) continue;
            const uint8_t* px = frame + ((size_t)iy * ti->w + ix) * ci;
            const int8_t* wp = w + ((co * k + ky) * k + kx) * ci;
            for (int c = 0; c < ci; c++) acc += (((int)px[c] - 128) - zx) * (int)wp[c]; /* QUANTIZE u8->s8 */
          }
        }
        out[((size_t)oy * to->w + ox) * to->c + co] = requant(acc, mu[co], to->zero_point, op->act_min, op->act_max);
      }
}

static void run_pw(vbto_model* m, const op_t* op) {
  const tens_t* ti = &m->tensors[op->inputs[0]];
  const tens_t* to = &m->tensors[op->output];
  const int8_t* x = m->data[op->inputs[0]];
  const int8_t* w = (const int8_t*)(m->blob + op->w_off);
  const int32_t* b = (const int32_t*)(m->blob + op->b_off);
  const float* mu = (const float*)(m->blob + op->m_off);
  int8_t* out = m->data[op->output];
  int ci = ti->c, co_n = to->c, zx = ti->zero_point;
  size_t npx = (size_t)ti->h * ti->w;
  int16_t* xs = (int16_t*)malloc(sizeof(int16_t) * ci);
  for (size_t p = 0; p < npx; p++) {
    const int8_t* xp = x + p * ci;
    for (int c = 0; c < ci; c++) xs[c] = (int16_t)(xp[c] - zx);
    for (int co = 0; co < co_n; co++) {
      const int8_t* wp = w + (size_t)co * ci;
      int32_t acc = 0;
      for (int c = 0; c < ci; c++) acc += (int32_t)xs[c] * (int32_t)wp[c];
      out[p * co_n + co] = requant(acc + b[co], mu[co], to->zero_point, op->act_min, op->act_max);
    }
  }
  free(xs);
}

static void run_dw(vbto_model* m, const op_t* op) {
  const tens_t* ti = &m->tensors[op->inputs[0]];
  const tens_t* to = &m->tensors[op->output];
  const int8_t* x = m->data[op->inputs[0]];
  const int8_t* w = (const int8_t*)(m->blob + op->w_off); /* [ky][kx][C] */
  const int32_t* b = (const int32_t*)(m->blob + op->b_off);
  const float* mu = (const float*)(m->blob + op->m_off);
  int8_t* out = m->data[op->output];
  int C = ti->c, k = op->k, zx = ti->zero_point;
  int32_t* acc = (int32_t*)malloc(sizeof(int32_t) * C);
  for (int oy = 0; oy < to->h; oy++)
    for (int ox = 0; ox < to->w; ox++) {
      for (int c = 0; c < C; c++) acc[c] = b[c];
      for (int ky = 0; ky < k; ky++) {
        int iy = oy * op->stride + ky - op->pad_t;
        if (iy < 0 || iy >= ti->h) continue;
        for (int kx = 0; kx < k; kx++) {
          int ix = ox * op->stride + kx - op->pad_l;
          if (ix < 0 || ix >= ti->w) continue;
          const int8_t* xp = x + ((size_t)iy * ti->w + ix) * C;
          const int8_t* wp = w + (size_t)(ky * k + kx) * C;
          for (int c = 0; c < C; c++) acc[c] += ((int32_t)xp[c] - zx) * (int32_t)wp[c];
        }
      }
      int8_t* o = out + ((size_t)oy * to->w + ox) * C;
      for (int c = 0; c < C; c++) o[c] = requant(acc[c], mu[c], to->zero_point, op->act_min, op->act_max);
    }
  free(acc);
}

/* XNNPACK qs8 ADD: xnn_create_add_nd_qs8 (input_output_scale = input_scale / output_scale in float32, each in
 * [2^-10, 2^8)) + xnn_init_qs8_add_minmax_*_params.  Returns 0 when the scales are acceptable. */
typedef struct { int32_t bias, a_mult, b_mult, shift; } addq_t;
static int32_t scale_to_multiplier(float v, uint32_t shift) {
  uint32_t bits;
  memcpy(&bits, &v, 4);
  bits += shift << 23; /* v * 2^shift */
  float f;
  memcpy(&f, &bits, 4);
  return (int32_t)lrintf(f);
}
static int add_params(float s_a, float s_b, float s_out, int z_a, int z_b, addq_t* p) {
  const float a_os = s_a / s_out;
  const float b_os = s_b / s_out;
  if (!(a_os >= 0x1.0p-10f && a_os < 0x1.0p+8f)) return -1;
  if (!(b_os >= 0x1.0p-10f && b_os < 0x1.0p+8f)) return -1;
  const float max_os = fmaxf(a_os, b_os);
  uint32_t bits;
  memcpy(&bits, &max_os, 4);
  const int32_t max_exponent = (int32_t)(bits >> 23) - 127;
  const uint32_t shift = (uint32_t)(20 /* multiplier bits */ - max_exponent);
  if (shift < 12 || shift > 30) return -1;
  p->a_mult = scale_to_multiplier(a_os, shift);
  p->b_mult = scale_to_multiplier(b_os, shift);
  p->shift = (int32_t)shift;
  const int32_t rounding = (int32_t)1 << (shift - 1);
  p->bias = rounding - p->a_mult * (int32_t)z_a - p->b_mult * (int32_t)z_b;
  return 0;
}
static inline int32_t sat(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* XNNPACK xnn_create_sigmoid_nc_qs8 table entry for int8 input value i (output scale 1/256, zero point -128) */
static int sigmoid_lut_entry(float s_in, int z_in, int i) {
  const float x = s_in * (float)(i - z_in);
  float y = 256.0f / (1.0f + expf(-x));
  if (y < 0.0f) y = 0.0f;     /* output_min - output_zero_point */
  if (y > 255.0f) y = 255.0f; /* output_max - output_zero_point */
  return (int)lrintf(y) - 128;
}

/* test hooks: the parameter derivation and the element kernel on caller-provided vectors */
int vbto_add_params(float s_a, float s_b, float s_out, int z_a, int z_b, int32_t out4[4]) {
  addq_t p;
  if (add_params(s_a, s_b, s_out, z_a, z_b, &p)) return -1;
  out4[0] = p.bias; out4[1] = p.a_mult; out4[2] = p.b_mult; out4[3] = p.shift;
  return 0;
}
void vbto_add_vec(const int8_t* a, const int8_t* b, long n, const int32_t q4[4], int z_out, int lo, int hi, int8_t* out) {
  for (long i = 0; i < n; i++) {
    int32_t acc = q4[0] + (int32_t)a[i] * q4[1] + (int32_t)b[i] * q4[2];
    acc >>= q4[3];
    int32_t v = sat(acc, -32768, 32767);
    v = sat(v + z_out, -32768, 32767);
    v = sat(v, -128, 127);
    out[i] = (int8_t)sat(v, lo, hi);
  }
}
int vbto_sigmoid_lut_entry(float s_in, int z_in, int i) { return sigmoid_lut_entry(s_in, z_in, i); }

static int run_add(vbto_model* m, const op_t* op) {
  if (op->n_inputs != 2) return -1; /* TFLite ADD is binary */
  const tens_t* ta = &m->tensors[op->inputs[0]];
  const tens_t* tb = &m->tensors[op->inputs[1]];
  const tens_t* to = &m->tensors[op->output];
  addq_t p;
  if (add_params(ta->scale, tb->scale, to->scale, ta->zero_point, tb->zero_point, &p)) return -1;
  if (p.bias != op->add_q[0] || p.a_mult != op->add_q[1] || p.b_mult != op->add_q[2] || p.shift != op->add_q[3]) return -2;
  size_t n = (size_t)to->h * to->w * to->c;
  int8_t* out = m->data[op->output];
  const int8_t* a = m->data[op->inputs[0]];
  const int8_t* b = m->data[op->inputs[1]];
  for (size_t i = 0; i < n; i++) {
    int32_t acc = p.bias + (int32_t)a[i] * p.a_mult + (int32_t)b[i] * p.b_mult;
    acc >>= p.shift;                                  /* _mm256_sra_epi32: arithmetic */
    int32_t v = sat(acc, -32768, 32767);              /* _mm_packs_epi32 */
    v = sat(v + to->zero_point, -32768, 32767);       /* _mm_adds_epi16 */
    v = sat(v, -128, 127);                            /* _mm_packs_epi16 */
    if (v < op->act_min) v = op->act_min;             /* _mm_max_epi8 / _mm_min_epi8 */
    if (v > op->act_max) v = op->act_max;
    out[i] = (int8_t)v;
  }
  return 0;
}

static void run_maxpool(vbto_model* m, const op_t* op) {
  const tens_t* ti = &m->tensors[op->inputs[0]];
  const tens_t* to = &m->tensors[op->output];
  const int8_t* x = m->data[op->inputs[0]];
  int8_t* out = m->data[op->output];
  int C = ti->c;
  for (int oy = 0; oy < to->h; oy++)
    for (int ox = 0; ox < to->w; ox++)
      for (int c = 0; c < C; c++) {
        int best = -128;
        for (int ky = 0; ky < op->k; ky++) {
          int iy = oy * op->stride + ky - op->pad_t;
          if (iy < 0 || iy >= ti->h) continue;
          for (int kx = 0; kx < op->k; kx++) {
            int ix = ox * op->stride + kx - op->pad_l;
            if (ix < 0 || ix >= ti->w) continue;
            int v = x[((size_t)iy * ti->w + ix) * C + c];
            if (v > best) best = v;
          }
        }
        out[((size_t)oy * to->w + ox) * C + c] = (int8_t)best;
      }
}

/* TFLite reference_ops::ResizeNearestNeighbor / GetNearestNeighbor with align_corners = half_pixel_centers = false */
static int nearest_src(int dst, int in_size, int out_size) {
  const float scale = (float)in_size / (float)out_size;
  int v = (int)floorf((float)dst * scale);
  return v < in_size - 1 ? v : in_size - 1;
}
static void run_resize(vbto_model* m, const op_t* op) {
  const tens_t* ti = &m->tensors[op->inputs[0]];
  const tens_t* to = &m->tensors[op->output];
  const int8_t* x = m->data[op->inputs[0]];
  int8_t* out = m->data[op->output];
  int C = ti->c;
  for (int oy = 0; oy < to->h; oy++) {
    int iy = nearest_src(oy, ti->h, to->h);
    for (int ox = 0; ox < to->w; ox++) {
      int ix = nearest_src(ox, ti->w, to->w);
      memcpy(out + ((size_t)oy * to->w + ox) * C, x + ((size_t)iy * ti->w + ix) * C, C);
    }
  }
}

/* ---- TFLite_Detection_PostProcess, fast single-class path ---- */
typedef struct { float ymin, xmin, ymax, xmax; } box_t;

static float iou(box_t a, box_t b) {
  float area_a = (a.ymax - a.ymin) * (a.xmax - a.xmin);
  float area_b = (b.ymax - b.ymin) * (b.xmax - b.xmin);
  if (area_a <= 0.0f || area_b <= 0.0f) return 0.0f;
  float iy0 = fmaxf(a.ymin, b.ymin), ix0 = fmaxf(a.xmin, b.xmin);
  float iy1 = fminf(a.ymax, b.ymax), ix1 = fminf(a.xmax, b.xmax);
  float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
  return inter / (area_a + area_b - inter);
}

/* DecreasingArgSort of detection_postprocess.cc: std::stable_sort on the float SCORES, descending.  Two class bytes on a
 * plateau of the LOGISTIC table have the same score and therefore keep their anchor order. */
typedef struct { float score; int32_t q; int32_t idx; } cand_t;
static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa; const cand_t* b = (const cand_t*)pb;
  if (a->score != b->score) return a->score > b->score ? -1 : 1;  /* score descending */
  return a->idx - b->idx;                                         /* equal scores: lower anchor index first (stable) */
}

static int run_postprocess(vbto_model* m, const op_t* op, float* boxes, float* scores, float* classes, int32_t* count) {
  int A = m->hdr.num_anchors, maxdet = m->hdr.max_detections;
  const float* anchors = (const float*)(m->blob + op->aux_off);
  /* tables written by the model tool: score f32[256] | box f32[256] | dq f64[256] | ex f64[256] | scales f32[4] */
  const float* score_tab = (const float*)(m->blob + op->aux2_off);
  const float* box_tab = score_tab + 256;
  const double* dq_tab = (const double*)(box_tab + 256);
  const double* ex_tab = dq_tab + 256;
  const float* scale_values = (const float*)(ex_tab + 256); /* y, x, h, w */
  int nl = op->n_inputs / 2;
  const tens_t* tc = &m->tensors[op->inputs[0]];
  const tens_t* tb = &m->tensors[op->inputs[nl]];
  /* the tables are re-derived here from the tensor scales (LOGISTIC, DEQUANTIZE and the exp() of the decode) */
  float score_lut[256], box_lut[256];
  double dq_lut[256], ex_lut[256];
  for (int i = -128; i < 128; i++) {
    score_lut[i + 128] = (1.0f / 256.0f) * (float)(sigmoid_lut_entry(tc->scale, tc->zero_point, i) + 128);
    box_lut[i + 128] = tb->scale * (float)(i - tb->zero_point);
    dq_lut[i + 128] = (double)box_lut[i + 128] / (double)scale_values[0];
    ex_lut[i + 128] = exp((double)box_lut[i + 128] / (double)scale_values[2]);
    if (score_lut[i + 128] != score_tab[i + 128] || box_lut[i + 128] != box_tab[i + 128] || dq_lut[i + 128] != dq_tab[i + 128] ||
        ex_lut[i + 128] != ex_tab[i + 128])
      return -3;
  }
  if (scale_values[0] != scale_values[1] || scale_values[2] != scale_values[3]) return -3;
  for (int l = 1; l < nl; l++) /* CONCATENATION requires one quantisation for all of its inputs */
    if (m->tensors[op->inputs[l]].scale != tc->scale || m->tensors[op->inputs[l]].zero_point != tc->zero_point ||
        m->tensors[op->inputs[nl + l]].scale != tb->scale || m->tensors[op->inputs[nl + l]].zero_point != tb->zero_point)
      return -3;
  /* class predictions [A][C] (C = num_classes_with_background of detection_postprocess.cc; the op's num_classes equals it for the
   * reference's models, label_offset = 0).  NonMaxSuppressionMultiClassFastHelper with max_classes_per_detection = 1: per anchor the
   * best class by DecreasingPartialArgSort(.., 1) = ArgMaxVector on the float scores = the FIRST column holding the highest score
   * (two class bytes on a plateau of the LOGISTIC table are one score), then single-class NMS on those best scores. */
  const int C = m->hdr.num_classes > 0 ? m->hdr.num_classes : 1;
  int8_t* cls = (int8_t*)malloc((size_t)A);          /* class byte of the best column */
  int32_t* cid = (int32_t*)malloc(sizeof(int32_t) * (size_t)A);
  int8_t* box = (int8_t*)malloc((size_t)A * 4);
  size_t o = 0;
  for (int l = 0; l < nl; l++) { /* RESHAPE + CONCATENATION of the per-level head outputs */
    const tens_t* t = &m->tensors[op->inputs[l]];
    size_t n = (size_t)t->h * t->w * t->c / (size_t)C; /* anchors of the level */
    const int8_t* src = m->data[op->inputs[l]];
    for (size_t a = 0; a < n; a++) {
      int best = 0;
      for (int c = 1; c < C; c++)
        if (score_lut[src[a * C + c] + 128] > score_lut[src[a * C + best] + 128]) best = c;
      cls[o + a] = src[a * C + best];
      cid[o + a] = best;
    }
    memcpy(box + o * 4, m->data[op->inputs[nl + l]], n * 4);
    o += n;
  }
  cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * A);
  int nc = 0;
  for (int i = 0; i < A; i++) { /* SelectDetectionsAboveScoreThreshold: score >= threshold */
    float s = score_lut[cls[i] + 128];
    if (s >= m->hdr.nms_score_threshold) { cand[nc].score = s; cand[nc].q = cls[i]; cand[nc].idx = i; nc++; }
  }
  qsort(cand, nc, sizeof(cand_t), cand_cmp); /* DecreasingArgSort is a stable sort: ties keep anchor order */
  box_t sel[64];
  int ns = 0;
  for (int j = 0; j < nc && ns < maxdet; j++) {
    int i = cand[j].idx;
    const float* an = anchors + (size_t)i * 4; /* ycenter, xcenter, h, w */
    const int8_t* bq = box + (size_t)i * 4;    /* dy, dx, dh, dw */
    /* DecodeCenterSizeBoxes: double intermediates, one rounding to float per quantity */
    float yc = (float)(dq_lut[bq[0] + 128] * (double)an[2] + (double)an[0]);
    float xc = (float)(dq_lut[bq[1] + 128] * (double)an[3] + (double)an[1]);
    float hh = (float)(0.5 * ex_lut[bq[2] + 128] * (double)an[2]);
    float hw = (float)(0.5 * ex_lut[bq[3] + 128] * (double)an[3]);
    box_t b = { yc - hh, xc - hw, yc + hh, xc + hw };
    int keep = 1;
    for (int s = 0; s < ns; s++)
      if (iou(sel[s], b) > m->hdr.nms_iou_threshold) { keep = 0; break; }
    if (!keep) continue;
    sel[ns] = b;
    boxes[ns * 4 + 0] = b.ymin; boxes[ns * 4 + 1] = b.xmin; boxes[ns * 4 + 2] = b.ymax; boxes[ns * 4 + 3] = b.xmax;
    scores[ns] = score_lut[cand[j].q + 128];
    classes[ns] = (float)cid[i];
    ns++;
  }
  for (int s = ns; s < maxdet; s++) {
    boxes[s * 4] = boxes[s * 4 + 1] = boxes[s * 4 + 2] = boxes[s * 4 + 3] = 0.0f;
    scores[s] = 0.0f; classes[s] = 0.0f;
  }
  *count = ns;
  free(cand); free(cls); free(cid); free(box);
  return 0;
}

/* One frame through the whole graph.  frame: uint8 [S,S,3] RGB.  Outputs like reference odt.py:64-66. */
int vbto_run(vbto_model* m, const uint8_t* frame, float* boxes, float* scores, float* classes, int32_t* count) {
  for (int i = 0; i < m->hdr.num_ops; i++) {
    const op_t* op = &m->ops[i];
    switch (op->type) {
      case OP_STEM: run_stem(m, op, frame); break;
      case OP_PW: run_pw(m, op); break;
      case OP_DW: run_dw(m, op); break;
      case OP_ADD: if (run_add(m, op)) return -2; break;
      case OP_MAXPOOL: run_maxpool(m, op); break;
      case OP_RESIZE_NN: run_resize(m, op); break;
      case OP_POSTPROCESS: if (run_postprocess(m, op, boxes, scores, classes, count)) return -3; break;
      default: return -1;
    }
  }
  return 0;
}

/* Batch helper for the CPU baseline: frames [B,S,S,3]; one private model copy per thread. */
int vbto_run_batch(const char* path, const uint8_t* frames, int B, int threads,
                   float* boxes, float* scores, float* classes, int32_t* counts) {
  int err = 0;
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
    vbto_model* m = vbto_load(path);
    if (!m) {
#pragma omp atomic write
      err = -1;
    } else {
      size_t fsz = (size_t)m->hdr.image_size * m->hdr.image_size * 3;
      int md = m->hdr.max_detections;
#pragma omp for schedule(dynamic, 1)
      for (int b = 0; b < B; b++)
        if (vbto_run(m, frames + (size_t)b * fsz, boxes + (size_t)b * md * 4, scores + (size_t)b * md,
                     classes + (size_t)b * md, counts + b)) {
#pragma omp atomic write
          err = -2;
        }
      vbto_free(m);
    }
  }
  return err;
}
