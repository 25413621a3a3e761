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
 *   - TFLite full-integer conv / depthwise conv: int8 x int8 -> int32 accumulate,
 *     acc = sum (x_q - z_x) * w_q + bias_q ; per-output-channel requantisation in float32 as the
 *     XNNPACK QS8 "fp32" requantisation does: q = clamp(rne(float(acc) * M[c]) + z_y);
 *   - int8 ADD with per-input float multipliers, MAX_POOL_2D 3x3/2 SAME, RESIZE_NEAREST_NEIGHBOR
 *     (align_corners=false, half_pixel_centers=false);
 *   - int8 LOGISTIC (output scale 1/256, zp -128) as a 256-entry table;
 *   - TFLite_Detection_PostProcess (detection_postprocess.cc, fast single-class path):
 *     centre-size decode, score filter, descending sort, greedy IoU suppression, top-25.
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
  int32_t input_tensor, reserved[17];
} hdr_t;
typedef struct { int32_t h, w, c, zero_point; float scale; int32_t pad[3]; } tens_t;
typedef struct {
  int32_t type, n_inputs, inputs[12], output, k, stride, pad_t, pad_l, act_min, act_max, level;
  int64_t w_off, b_off, m_off, aux_off, aux2_off;
  float in_mult[3];
  int32_t reserved[5];
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
  if (fread(&m->hdr, sizeof(hdr_t), 1, f) != 1 || memcmp(m->hdr.magic, "VBTM0001", 8) != 0) { fclose(f); free(m); return NULL; }
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

static void run_add(vbto_model* m, const op_t* op) {
  const tens_t* to = &m->tensors[op->output];
  size_t n = (size_t)to->h * to->w * to->c;
  int8_t* out = m->data[op->output];
  const int8_t* a = m->data[op->inputs[0]];
  const int8_t* b = m->data[op->inputs[1]];
  const int8_t* c = op->n_inputs > 2 ? m->data[op->inputs[2]] : NULL;
  int za = m->tensors[op->inputs[0]].zero_point, zb = m->tensors[op->inputs[1]].zero_point;
  int zc = c ? m->tensors[op->inputs[2]].zero_point : 0;
  for (size_t i = 0; i < n; i++) {
    float r = (float)(a[i] - za) * op->in_mult[0];
    r = fmaf((float)(b[i] - zb), op->in_mult[1], r);
    if (c) r = fmaf((float)(c[i] - zc), op->in_mult[2], r);
    int32_t q = (int32_t)nearbyintf(r) + to->zero_point;
    if (q < op->act_min) q = op->act_min;
    if (q > op->act_max) q = op->act_max;
    out[i] = (int8_t)q;
  }
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

static void run_resize(vbto_model* m, const op_t* op) {
  const tens_t* ti = &m->tensors[op->inputs[0]];
  const tens_t* to = &m->tensors[op->output];
  const int8_t* x = m->data[op->inputs[0]];
  int8_t* out = m->data[op->output];
  int C = ti->c;
  for (int oy = 0; oy < to->h; oy++) {
    int iy = (oy * ti->h) / to->h;
    for (int ox = 0; ox < to->w; ox++) {
      int ix = (ox * ti->w) / to->w;
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

typedef struct { int32_t q; int32_t idx; } cand_t;
static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa; const cand_t* b = (const cand_t*)pb;
  if (a->q != b->q) return b->q - a->q;  /* score descending */
  return a->idx - b->idx;                /* ties: lower anchor index first */
}

static void run_postprocess(vbto_model* m, const op_t* op, float* boxes, float* scores, float* classes, int32_t* count) {
  int A = m->hdr.num_anchors, maxdet = m->hdr.max_detections;
  const float* anchors = (const float*)(m->blob + op->aux_off);
  const float* score_lut = (const float*)(m->blob + op->aux2_off);
  const float* delta_lut = score_lut + 256;
  const float* exp_lut = score_lut + 512;
  int nl = op->n_inputs / 2;
  int8_t* cls = (int8_t*)malloc(A);
  int8_t* box = (int8_t*)malloc((size_t)A * 4);
  size_t o = 0;
  for (int l = 0; l < nl; l++) { /* CONCATENATION of the per-level head outputs */
    const tens_t* t = &m->tensors[op->inputs[l]];
    size_t n = (size_t)t->h * t->w * t->c;
    memcpy(cls + o, m->data[op->inputs[l]], n);
    memcpy(box + o * 4, m->data[op->inputs[nl + l]], n * 4);
    o += n;
  }
  cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * A);
  int nc = 0;
  for (int i = 0; i < A; i++) {
    float s = score_lut[cls[i] + 128];
    if (s >= m->hdr.nms_score_threshold) { cand[nc].q = cls[i]; cand[nc].idx = i; nc++; }
  }
  qsort(cand, nc, sizeof(cand_t), cand_cmp);
  box_t sel[64];
  int ns = 0;
  for (int j = 0; j < nc && ns < maxdet; j++) {
    int i = cand[j].idx;
    const float* an = anchors + (size_t)i * 4; /* ycenter, xcenter, h, w */
    const int8_t* bq = box + (size_t)i * 4;    /* dy, dx, dh, dw */
    float yc = fmaf(delta_lut[bq[0] + 128], an[2], an[0]);
    float xc = fmaf(delta_lut[bq[1] + 128], an[3], an[1]);
    float hh = (0.5f * exp_lut[bq[2] + 128]) * an[2];
    float hw = (0.5f * exp_lut[bq[3] + 128]) * an[3];
    box_t b = { yc - hh, xc - hw, yc + hh, xc + hw };
    int keep = 1;
    for (int s = 0; s < ns; s++)
      if (iou(sel[s], b) > m->hdr.nms_iou_threshold) { keep = 0; break; }
    if (!keep) continue;
    sel[ns] = b;
    boxes[ns * 4 + 0] = b.ymin; boxes[ns * 4 + 1] = b.xmin; boxes[ns * 4 + 2] = b.ymax; boxes[ns * 4 + 3] = b.xmax;
    scores[ns] = score_lut[cand[j].q + 128];
    classes[ns] = 0.0f;
    ns++;
  }
  for (int s = ns; s < maxdet; s++) {
    boxes[s * 4] = boxes[s * 4 + 1] = boxes[s * 4 + 2] = boxes[s * 4 + 3] = 0.0f;
    scores[s] = 0.0f; classes[s] = 0.0f;
  }
  *count = ns;
  free(cand); free(cls); free(box);
}

/* One frame through the whole graph.  frame: uint8 [S,S,3] RGB.  Outputs like reference odt.py:64-66. */
int vbto_run(vbto_model* m, const uint8_t* frame, float* boxes, float* scores, float* classes, int32_t* count) {
  for (int i = 0; i < m->hdr.num_ops; i++) {
    const op_t* op = &m->ops[i];
    switch (op->type) {
      case OP_STEM: run_stem(m, op, frame); break;
      case OP_PW: run_pw(m, op); break;
      case OP_DW: run_dw(m, op); break;
      case OP_ADD: run_add(m, op); break;
      case OP_MAXPOOL: run_maxpool(m, op); break;
      case OP_RESIZE_NN: run_resize(m, op); break;
      case OP_POSTPROCESS: run_postprocess(m, op, boxes, scores, classes, count); break;
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
        vbto_run(m, frames + (size_t)b * fsz, boxes + (size_t)b * md * 4, scores + (size_t)b * md,
                 classes + (size_t)b * md, counts + b);
      vbto_free(m);
    }
  }
  return err;
}
