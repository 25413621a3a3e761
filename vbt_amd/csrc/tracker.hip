// On-device OC-SORT tracker, row assembly and rep analysis for gfx950 (MI355X).
//
// Replaces, per clip (reference track.py:157-234, plot.py:33-47,87-95):
//   ocsort.OCSort(max_age=30, asso_func="diou", iou_threshold=0.1).update(dets, [])   track.py:157,186
//   the kf.x[4:6] read-back and the 8-column row assembly                               track.py:189-234
//   the max-cumulative-distance id selection of the export                              track.py:107-115
//   rolling(5)/expanding means + VelocityTracker scan                                  plot.py:87-95,33-47
// Execution model: ONE WAVEFRONT PER CLIP (clips are independent, frames of a clip are strictly
// sequential: Kalman + phase state).  Lane t owns live tracker t (<= 64 per clip): predict, IoU /
// direction-cost columns, Kalman update and row emission are lane-parallel; the linear assignment
// is a shortest-augmenting-path solver with lanes = columns and wave min-reductions; list
// maintenance (births, deletions) uses ballots.  All arithmetic is FP64 with contraction off and
// follows the op order of the numpy formulation, so the Kalman velocities are bit-identical to
// the reference's committed DataFrames (tests/test_gpu_tracker.py).
//   The 7-state SORT filter decouples into three (position, velocity) 2x2 filters (cx, cy, s) and a
// scalar one (r): F/H/Q/R never couple them, so the dense 7x7 products of the numpy formulation
// reduce EXACTLY (same roundings; the dropped terms are exact zeros) to the closed forms below.
#include "common.h"

namespace vbt {

// Phase timers of the OC-SORT step (developer builds only: VBT_EXTRA_CXXFLAGS=-DVBT_TRK_PROF): s_memtime deltas per phase,
// accumulated by lane 0 of every wavefront; read back with vbt_tracker_prof_read.
#ifdef VBT_TRK_PROF
__device__ unsigned long long g_trk_prof[16];
#define TRK_T0() unsigned long long _tp = __builtin_amdgcn_s_memtime()
#define TRK_MARK(i) do { unsigned long long _tn = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_trk_prof[i], _tn - _tp); _tp = _tn; } while (0)
#else
#define TRK_T0() do {} while (0)
#define TRK_MARK(i) do {} while (0)
#endif

constexpr int MAXT = 64;
constexpr int MAXD = VBT_MAX_DETECTIONS;
constexpr int MAXPH = 512;  // phases kept per clip

struct Trk {
  double x[7];
  double B[3][4];  // covariance blocks (a b; c d) of (cx,vcx) (cy,vcy) (s,vs)
  double Pr;       // variance of r
  double sx[7], sB[3][4], sPr;  // frozen copy (observation-centric re-update)
  double last_z[4];
  double last_obs[5];
  double vel[2];
  double obs[4][5];  // observations keyed by age & 3
  double conf, cls;
  double cum, cum_c, prev_x, prev_y;  // running path length of the emitted rows (export id selection)
  int obs_age[4];
  int has_saved, observed, gap, has_obs, has_vel;
  int time_since_update, id, hits, hit_streak, age, nrows;
};

struct Row {
  long long id;
  double time, x, y, dx, dy, h, w;
};

struct ClipState {
  int ntrk, frame_count, next_id, overflow, nrows, rows_overflow, best_id, last_n;
  double best_cum;
  int order[MAXT];
  unsigned long long used;  // slot bitmap
  double last_out[MAXD][9];  // last update(): x1,y1,x2,y2,id,cls,score,dx,dy
  Trk trk[MAXT];
};

struct TrackParams {
  int max_age, min_hits, delta_t, asso;
  double iou_thr, inertia, det_thresh;
};

// ------------------------------------------------------------------------------------------
// Kalman filter pieces (see header comment)
// ------------------------------------------------------------------------------------------
// The filter state proper as a LOCAL value: x, the three 2x2 covariance blocks and the variance of r.  predict / update work on a copy
// in registers that is loaded from the track once and stored back once: through a `Trk&` into LDS or global memory the compiler has to
// assume that every store may change what the next load reads (the detection, the observation history and the state are all plain
// double arrays), so the filter arithmetic ran as a chain of store -> wait -> load.
struct KfCore {
  double x[7];
  double B[3][4];
  double Pr;
};
__device__ __forceinline__ void core_load(KfCore& c, const Trk& k) {
#pragma unroll
  for (int i = 0; i < 7; i++) c.x[i] = k.x[i];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) c.B[i][j] = k.B[i][j];
  c.Pr = k.Pr;
}
__device__ __forceinline__ void core_store(Trk& k, const KfCore& c) {
#pragma unroll
  for (int i = 0; i < 7; i++) k.x[i] = c.x[i];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) k.B[i][j] = c.B[i][j];
  k.Pr = c.Pr;
}
__device__ __forceinline__ void kf_predict(KfCore& k, double q44, double q66) {
#pragma unroll
  for (int i = 0; i < 3; i++) k.x[i] = k.x[i] + k.x[i + 4];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double a = k.B[i][0], b = k.B[i][1], c = k.B[i][2], d = k.B[i][3];
    double qv = i == 2 ? q66 : q44;
    k.B[i][0] = ((a + c) + (b + d)) + 1.0;
    k.B[i][1] = (b + d) + 0.0;
    k.B[i][2] = (c + d) + 0.0;
    k.B[i][3] = d + qv;
  }
  k.Pr = k.Pr + 1.0;
}

__device__ __forceinline__ void kf_update_math(KfCore& k, const double z[4]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double Rc = i == 2 ? 10.0 : 1.0;
    double a = k.B[i][0], b = k.B[i][1], c = k.B[i][2], d = k.B[i][3];
    double y = z[i] - k.x[i];
    double S = a + Rc;
    double si = 1.0 / S;
    double kp = a * si, kv = c * si;
    k.x[i] = k.x[i] + kp * y;
    k.x[i + 4] = k.x[i + 4] + kv * y;
    double omk = 1.0 - kp, nkv = 0.0 - kv;
    double M00 = omk * a, M01 = omk * b, M10 = nkv * a + c, M11 = nkv * b + d;
    double N00 = M00 * omk, N01 = M00 * nkv + M01, N10 = M10 * omk, N11 = M10 * nkv + M11;
    double KRp = kp * Rc, KRv = kv * Rc;
    k.B[i][0] = N00 + KRp * kp;
    k.B[i][1] = N01 + KRp * kv;
    k.B[i][2] = N10 + KRv * kp;
    k.B[i][3] = N11 + KRv * kv;
  }
  double y = z[3] - k.x[3];
  double S = k.Pr + 10.0;
  double si = 1.0 / S;
  double kk = k.Pr * si;
  k.x[3] = k.x[3] + kk * y;
  double omk = 1.0 - kk;
  k.Pr = (omk * k.Pr) * omk + (kk * 10.0) * kk;
}

// kf.update(z) with OC-SORT's freeze / unfreeze (observation-centric re-update); c = the track's filter state (registers)
__device__ inline void kf_update(Trk& k, KfCore& c, const double* z, double q44, double q66) {
  k.gap += 1;  // one more entry in history_obs since the last real observation
  if (z == nullptr) {
    if (k.observed) {  // first miss: freeze
#pragma unroll
      for (int i = 0; i < 7; i++) k.sx[i] = c.x[i];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) k.sB[i][j] = c.B[i][j];
      k.sPr = c.Pr;
      k.has_saved = 1;
    }
    k.observed = 0;
    return;
  }
  double zl[4] = {z[0], z[1], z[2], z[3]};
  if (!k.observed && k.has_saved) {  // unfreeze: replay a linear virtual trajectory over the gap
#pragma unroll
    for (int i = 0; i < 7; i++) c.x[i] = k.sx[i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) c.B[i][j] = k.sB[i][j];
    c.Pr = k.sPr;
    double x1 = k.last_z[0], y1 = k.last_z[1], s1 = k.last_z[2], r1 = k.last_z[3];
    double w1 = sqrt(s1 * r1), h1 = sqrt(s1 / r1);
    double x2 = z[0], y2 = z[1], s2 = z[2], r2 = z[3];
    double w2 = sqrt(s2 * r2), h2 = sqrt(s2 / r2);
    const int gap = k.gap;
    const double g = (double)gap;
    double dx = (x2 - x1) / g, dy = (y2 - y1) / g, dw = (w2 - w1) / g, dh = (h2 - h1) / g;
    for (int i = 0; i < gap; i++) {
      double f = (double)(i + 1);
      double xx = x1 + f * dx, yy = y1 + f * dy, ww = w1 + f * dw, hh = h1 + f * dh;
      double vz[4] = {xx, yy, ww * hh, ww / hh};
      kf_update_math(c, vz);
      if (i != gap - 1) kf_predict(c, q44, q66);
      else { zl[0] = vz[0]; zl[1] = vz[1]; zl[2] = vz[2]; zl[3] = vz[3]; }  // history ends with the virtual box
    }
    k.has_saved = 0;
  }
  k.observed = 1;
  kf_update_math(c, z);
  k.last_z[0] = zl[0]; k.last_z[1] = zl[1]; k.last_z[2] = zl[2]; k.last_z[3] = zl[3];
  k.gap = 0;
}

__device__ inline void bbox_to_z(const double* b, double z[4]) {
  double w = b[2] - b[0], h = b[3] - b[1];
  z[0] = b[0] + w / 2.0;
  z[1] = b[1] + h / 2.0;
  z[2] = w * h;
  z[3] = w / (h + 1e-6);
}
__device__ inline void x_to_bbox(const double* x, double o[4]) {
  double w = sqrt(x[2] * x[3]);
  double h = x[2] / w;
  o[0] = x[0] - w / 2.0; o[1] = x[1] - h / 2.0; o[2] = x[0] + w / 2.0; o[3] = x[1] + h / 2.0;
}

// `last_observation.sum() < 0` is how OC-SORT asks "no observation yet" (placeholder = five -1s); it
// also fires for a real box far enough outside the image, and that quirk is kept.
__device__ inline bool obs_sum_negative(const Trk& k) {
  double s = k.last_obs[0];
  s = s + k.last_obs[1]; s = s + k.last_obs[2]; s = s + k.last_obs[3]; s = s + k.last_obs[4];
  return s < 0.0;
}

// KalmanBoxTracker.update(bbox)   (bbox = x1,y1,x2,y2,score ; cls)
__device__ inline void trk_update(Trk& k, const double* det_in, double q44, double q66, int delta_t) {
  KfCore c;
  core_load(c, k);
  if (det_in == nullptr) { kf_update(k, c, nullptr, q44, q66); return; }     // (update(None) leaves the filter state as it is)
  const double det[6] = {det_in[0], det_in[1], det_in[2], det_in[3], det_in[4], det_in[5]};   // in registers before the first store
  k.conf = det[4];
  k.cls = det[5];
  if (!obs_sum_negative(k)) {
    const double* prev = nullptr;
    for (int i = 0; i < delta_t; i++) {
      int a = k.age - (delta_t - i);
      if (a >= 0 && k.obs_age[a & 3] == a) { prev = k.obs[a & 3]; break; }
    }
    if (!prev) prev = k.last_obs;
    const double p0 = prev[0], p1 = prev[1], p2 = prev[2], p3 = prev[3];
    double cx1 = (p0 + p2) / 2.0, cy1 = (p1 + p3) / 2.0;
    double cx2 = (det[0] + det[2]) / 2.0, cy2 = (det[1] + det[3]) / 2.0;
    double sy = cy2 - cy1, sx = cx2 - cx1;
    double norm = sqrt(sy * sy + sx * sx) + 1e-6;
    k.vel[0] = sy / norm;
    k.vel[1] = sx / norm;
    k.has_vel = 1;
  }
#pragma unroll
  for (int i = 0; i < 5; i++) { k.last_obs[i] = det[i]; k.obs[k.age & 3][i] = det[i]; }
  k.obs_age[k.age & 3] = k.age;
  k.has_obs = 1;
  k.time_since_update = 0;
  k.hits += 1;
  k.hit_streak += 1;
  double z[4];
  bbox_to_z(det, z);
  kf_update(k, c, z, q44, q66);
  core_store(k, c);
}

__device__ inline double iou_xyxy(const double* a, const double* b) {
  double xx1 = fmax(a[0], b[0]), yy1 = fmax(a[1], b[1]), xx2 = fmin(a[2], b[2]), yy2 = fmin(a[3], b[3]);
  double w = fmax(0.0, xx2 - xx1), h = fmax(0.0, yy2 - yy1);
  double wh = w * h;
  return wh / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - wh);
}
__device__ inline double diou_xyxy(const double* a, const double* b) {
  double iou = iou_xyxy(a, b);
  double cx1 = (a[0] + a[2]) / 2.0, cy1 = (a[1] + a[3]) / 2.0, cx2 = (b[0] + b[2]) / 2.0, cy2 = (b[1] + b[3]) / 2.0;
  double ex = cx1 - cx2, ey = cy1 - cy2;
  double inner = ex * ex + ey * ey;
  double xc1 = fmin(a[0], b[0]), yc1 = fmin(a[1], b[1]), xc2 = fmax(a[2], b[2]), yc2 = fmax(a[3], b[3]);
  double ox = xc2 - xc1, oy = yc2 - yc1;
  double outer = ox * ox + oy * oy;
  return (iou - inner / outer + 1.0) / 2.0;
}

// ------------------------------------------------------------------------------------------
// wave-parallel rectangular linear assignment (nr rows <= nc cols <= 64, lanes = columns).
// cost(i, j) = tr ? C[j*ld + i] : C[i*ld + j].  Writes row2col[0..nr).
// OC-SORT's second association compares detections with never-observed trackers whose placeholder
// boxes are identical, so exact cost ties are routine and the result depends on the solver's tie
// rule.  This is therefore a faithful lane-parallel port of the solver the oracle uses
// [EXTERNAL: scipy.optimize.linear_sum_assignment = Crouse's shortest augmenting path,
// rectangular_lsap.cpp]: same `remaining` order (reverse fill, swap-with-last removal), same
// selection rule (lowest cost; among equals the LAST unassigned column scanned, else the first),
// same dual updates and the same floating-point expression ((minVal + c) - u) - v.
// ------------------------------------------------------------------------------------------
// Wave-wide reductions on the DPP path (row shifts inside rows of 16 lanes, then the GFX9 row broadcasts; lane 63 ends with the
// result, which comes back over the scalar path): six dependent steps of two or three VALU instructions instead of six shuffles
// through the LDS crossbar per 32-bit half.  The values met are the ones the shuffles met, so every result is unchanged.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double old, double x) {
  const unsigned long long o = (unsigned long long)__double_as_longlong(old), v = (unsigned long long)__double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)v, CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(v >> 32), CTRL, ROWMASK, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double bcast63_f64(double x) {
  const unsigned long long v = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_min_f64(double x) {
  const double I = __builtin_inf();
  x = fmin(x, dpp_f64<0x111, 0xf>(I, x));
  x = fmin(x, dpp_f64<0x112, 0xf>(I, x));
  x = fmin(x, dpp_f64<0x114, 0xf>(I, x));
  x = fmin(x, dpp_f64<0x118, 0xf>(I, x));
  x = fmin(x, dpp_f64<0x142, 0xa>(I, x));
  x = fmin(x, dpp_f64<0x143, 0xc>(I, x));
  return bcast63_f64(x);
}
__device__ __forceinline__ double wave_max_f64(double x) {
  const double I = -__builtin_inf();
  x = fmax(x, dpp_f64<0x111, 0xf>(I, x));
  x = fmax(x, dpp_f64<0x112, 0xf>(I, x));
  x = fmax(x, dpp_f64<0x114, 0xf>(I, x));
  x = fmax(x, dpp_f64<0x118, 0xf>(I, x));
  x = fmax(x, dpp_f64<0x142, 0xa>(I, x));
  x = fmax(x, dpp_f64<0x143, 0xc>(I, x));
  return bcast63_f64(x);
}
__device__ __forceinline__ int wave_max_i32(int x) {
  constexpr int I = -2147483647 - 1;
  x = max(x, __builtin_amdgcn_update_dpp(I, x, 0x111, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(I, x, 0x112, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(I, x, 0x114, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(I, x, 0x118, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(I, x, 0x142, 0xa, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(I, x, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(x, 63);
}

#ifdef VBT_NO_LAP_SMALL
__device__ __forceinline__ bool lap_small_off() { return true; }
#else
__device__ __forceinline__ bool lap_small_off() { return false; }
#endif

struct LapShared {
  double u[MAXT];
  double spc[MAXT];
  int col4row[MAXT];
  int row4col[MAXT];
  int path[MAXT];
  int remaining[MAXT];
};

// The same solver for 2..4 rows (a frame of the reference holds at most 3 plates; the second association sees even fewer rows) with
// every array in registers: the per-row state (duals u, col4row) is wave-uniform, the per-column state (dual v, row4col, path,
// shortest path cost, position in `remaining`) belongs to the column's lane, `remaining[index]` is "the lane whose position is
// index" (a ballot), and a value of another lane comes over the scalar path (v_readlane).  No LDS array, no barrier inside; the
// arithmetic - ((minVal + c) - u) - v, the dual updates, the tie rule - is the general solver's, statement for statement.
__device__ __forceinline__ double readlane_f64(double x, int l) {
  const unsigned long long v = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ void lap_small(const double* C, int ld, bool tr, int nr, int nc, int* row2col, int lane) {
  const double INF = __builtin_inf();
  const bool col = lane < nc;
  double cst[4];
#pragma unroll
  for (int q = 0; q < 4; q++) cst[q] = (col && q < nr) ? (tr ? C[lane * ld + q] : C[q * ld + lane]) : INF;
  double u[4] = {0.0, 0.0, 0.0, 0.0};
  int c4r[4] = {-1, -1, -1, -1};
  double v = 0.0;
  int r4c = -1, path = -1;
  auto sel_d = [](const double a[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3]; };
  for (int cur = 0; cur < nr; cur++) {
    double minVal = 0.0;
    int num_rem = nc;
    int pos = col ? nc - 1 - lane : -1;
    double spc = INF;
    unsigned sr = 0u;
    int i = cur, sink = -1;
    while (sink == -1) {
      sr |= 1u << i;
      const double ui = sel_d(u, i);
      const bool active = col && pos >= 0;
      if (active) {
        const double c = sel_d(cst, i);
        const double r = ((minVal + c) - ui) - v;
        if (r < spc) { path = i; spc = r; }
      }
      const double lowest = wave_min_f64(active ? spc : INF);
      const bool cand = active && spc == lowest;
      const unsigned long long un = __ballot(cand && r4c == -1);
      int key;  // choose: unassigned candidates -> max position, else min position
      if (un) key = (cand && r4c == -1) ? pos : -1;
      else key = cand ? -pos : -(1 << 20);
      key = wave_max_i32(key);
      const int index = un ? key : -key;
      minVal = lowest;
      const int j = __ffsll((long long)__ballot(active && pos == index)) - 1;     // remaining[index]
      const int jrow = __builtin_amdgcn_readlane(r4c, j);
      if (jrow == -1) sink = j; else i = jrow;
      num_rem -= 1;
      const int jl = __ffsll((long long)__ballot(active && pos == num_rem)) - 1;  // remaining[num_rem]
      if (lane == j) pos = -1;
      if (lane == jl && jl != j) pos = index;
    }
    // ---- dual updates ----
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q >= nr) continue;
      if (q == cur) u[q] += minVal;
      else if ((sr >> q) & 1u) u[q] += minVal - readlane_f64(spc, c4r[q]);
    }
    if (col && pos < 0) v -= minVal - spc;
    // ---- augment ----
    int j = sink;
    while (true) {
      const int ii = __builtin_amdgcn_readlane(path, j);
      if (lane == j) r4c = ii;
      const int t = ii == 0 ? c4r[0] : ii == 1 ? c4r[1] : ii == 2 ? c4r[2] : c4r[3];
#pragma unroll
      for (int q = 0; q < 4; q++) if (q == ii) c4r[q] = j;
      j = t;
      if (ii == cur) break;
    }
  }
  if (lane < nr) row2col[lane] = lane == 0 ? c4r[0] : lane == 1 ? c4r[1] : lane == 2 ? c4r[2] : c4r[3];
  __syncthreads();
}

__device__ void lap_solve(const double* C, int ld, bool tr, int nr, int nc, int* row2col, LapShared& S, int lane) {
  const double INF = __builtin_inf();
  if (nr >= 2 && nr <= 4 && !lap_small_off()) { lap_small(C, ld, tr, nr, nc, row2col, lane); return; }
  if (nr == 1) {
    // One row (OC-SORT's second association usually has one unmatched detection): the first augmenting path of the solver
    // ends at the cheapest column; among equal costs it takes the LAST one it scans, and it scans remaining[] = nc-1 ... 0,
    // i.e. the lowest column index.  No dual update can change a one-row result.
    double c = lane < nc ? (tr ? C[lane * ld] : C[lane]) : INF;
    const double lowest = wave_min_f64(c);
    const unsigned long long cand = __ballot(lane < nc && c == lowest);
    if (lane == 0) row2col[0] = __ffsll((long long)cand) - 1;
    __syncthreads();
    return;
  }
  double v = 0.0;
  if (lane < nr) { S.u[lane] = 0.0; S.col4row[lane] = -1; }
  if (lane < nc) { S.row4col[lane] = -1; S.path[lane] = -1; }
  __syncthreads();
  for (int cur = 0; cur < nr; cur++) {
    // ---- augmenting_path ----
    double minVal = 0.0;
    int num_rem = nc;
    int pos = lane < nc ? nc - 1 - lane : -1;   // remaining[it] = nc - it - 1
    if (lane < nc) S.remaining[nc - 1 - lane] = lane;
    double spc = INF;
    unsigned long long sr = 0ull;
    int i = cur, sink = -1;
    __syncthreads();
    while (sink == -1) {
      sr |= 1ull << i;
      const double ui = S.u[i];
      const bool active = lane < nc && pos >= 0;
      const int r4c = lane < nc ? S.row4col[lane] : 0;
      if (active) {
        double c = tr ? C[lane * ld + i] : C[i * ld + lane];
        double r = ((minVal + c) - ui) - v;
        if (r < spc) { S.path[lane] = i; spc = r; }
      }
      const double lowest = wave_min_f64(active ? spc : INF);
      const bool cand = active && spc == lowest;
      const unsigned long long un = __ballot(cand && r4c == -1);
      int key;  // choose: unassigned candidates -> max position, else min position
      if (un) key = (cand && r4c == -1) ? pos : -1;
      else key = cand ? -pos : -(1 << 20);
      key = wave_max_i32(key);
      const int index = un ? key : -key;
      minVal = lowest;
      const int j = S.remaining[index];
      const int jrow = S.row4col[j];
      __syncthreads();
      if (jrow == -1) sink = j; else i = jrow;
      // SC[j] = true ; remaining[index] = remaining[--num_remaining]
      num_rem -= 1;
      const int jl = S.remaining[num_rem];
      __syncthreads();
      if (lane == j) pos = -1;
      if (lane == jl && jl != j) pos = index;
      if (lane == 0) S.remaining[index] = jl;
      __syncthreads();
    }
    // ---- dual updates ----
    if (lane < nc) S.spc[lane] = spc;
    __syncthreads();
    if (lane < nr) {
      if (lane == cur) S.u[lane] += minVal;
      else if ((sr >> lane) & 1ull) S.u[lane] += minVal - S.spc[S.col4row[lane]];
    }
    if (lane < nc && pos < 0) v -= minVal - spc;
    __syncthreads();
    // ---- augment ----
    if (lane == 0) {
      int j = sink;
      while (true) {
        int ii = S.path[j];
        S.row4col[j] = ii;
        int t = S.col4row[ii];
        S.col4row[ii] = j;
        j = t;
        if (ii == cur) break;
      }
    }
    __syncthreads();
  }
  if (lane < nr) row2col[lane] = S.col4row[lane];
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// one OCSort.update() for one clip, executed by one wavefront
// ------------------------------------------------------------------------------------------
struct StepShared {
  double det[MAXD][6];
  double tbox[MAXT][4];
  double tq[MAXT][5];  // per tracker: previous-observation centre (x, y), its validity, velocity direction (x, y)
  double iou[MAXD][MAXT];
  double cost[MAXD][MAXT];
  int d2t[MAXD];      // detection -> tracker position paired by the first association (or -1)
  int rej[MAXD];      // that pair was rejected (IoU below threshold)
  int taken[MAXD];    // detection consumed by a tracker (first or second association)
  int r2c[MAXT];      // assignment scratch
  int um_d[MAXD];     // unmatched detections, in the reference's list order
  int um_t[MAXT];
  int n_um_d, n_um_t, flag;
  LapShared lap;
};

__device__ void ocsort_step(ClipState& st, Row* rows, int rows_cap, StepShared& sh, int nd, double time, const TrackParams& p,
                            double q44, double q66, int lane) {
  TRK_T0();
  if (lane == 0) st.frame_count += 1;
  int T = st.ntrk;
  // ---- predict (KalmanBoxTracker.predict) ----
  bool isnan_box = false;
  if (lane < T) {
    Trk& k = st.trk[st.order[lane]];
    KfCore c;
    core_load(c, k);
    if ((c.x[6] + c.x[2]) <= 0.0) c.x[6] *= 0.0;
    kf_predict(c, q44, q66);
    core_store(k, c);
    k.age += 1;
    if (k.time_since_update > 0) k.hit_streak = 0;
    k.time_since_update += 1;
    double bx[4];
    x_to_bbox(c.x, bx);
    isnan_box = (bx[0] != bx[0]) || (bx[1] != bx[1]) || (bx[2] != bx[2]) || (bx[3] != bx[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) sh.tbox[lane][i] = bx[i];
  }
  unsigned long long nanmask = __ballot(isnan_box);
  if (nanmask) {  // drop trackers whose predicted box is NaN (stable compaction)
    unsigned long long keep = ~nanmask & (T >= 64 ? ~0ull : ((1ull << T) - 1ull));
    int slot = lane < T ? st.order[lane] : 0;
    double bx[4] = {0, 0, 0, 0};
    if (lane < T)
      for (int i = 0; i < 4; i++) bx[i] = sh.tbox[lane][i];
    __syncthreads();
    {  // slots of the dropped trackers go back to the pool: one lane clears their bits (no atomics: the state may sit in LDS)
      unsigned long long nm = nanmask, clr = 0ull;
      while (nm) {
        const int l = __ffsll((long long)nm) - 1;
        nm &= nm - 1;
        clr |= 1ull << __shfl(slot, l);
      }
      if (lane == 0) st.used &= ~clr;
    }
    if (lane < T && !isnan_box) {
      int np_ = __popcll(keep & ((1ull << lane) - 1ull));
      st.order[np_] = slot;
      for (int i = 0; i < 4; i++) sh.tbox[np_][i] = bx[i];
    }
    T = __popcll(keep);
    if (lane == 0) st.ntrk = T;
  }
  __syncthreads();
  const int slot = lane < T ? st.order[lane] : 0;
  TRK_MARK(0);   // predict
  // ---- first association: IoU + velocity-direction consistency ----
  // Per-tracker quantities by lane = tracker, then the (detection, tracker) cost entries dealt to the 64 lanes pair by pair:
  // every entry is the same sequence of double operations as before, but a frame with 20 detections and 3 trackers is one
  // pass of 60 lanes instead of 20 dependent passes of 3 (sqrt / acos in double dominate the step).
  if (lane < T) {
    Trk& k = st.trk[slot];
    const double* pobs = nullptr;  // k_previous_obs
    if (k.has_obs) {
      for (int i = 0; i < p.delta_t; i++) {
        int a = k.age - (p.delta_t - i);
        if (a >= 0 && k.obs_age[a & 3] == a) { pobs = k.obs[a & 3]; break; }
      }
      if (!pobs) pobs = k.last_obs;
    }
    double pcx = -1.0, pcy = -1.0, valid = 0.0;
    if (pobs) { pcx = (pobs[0] + pobs[2]) / 2.0; pcy = (pobs[1] + pobs[3]) / 2.0; valid = pobs[4] < 0.0 ? 0.0 : 1.0; }
    sh.tq[lane][0] = pcx; sh.tq[lane][1] = pcy; sh.tq[lane][2] = valid;
    sh.tq[lane][3] = k.has_vel ? k.vel[1] : 0.0;   // vx
    sh.tq[lane][4] = k.has_vel ? k.vel[0] : 0.0;   // vy
  }
  __syncthreads();
  {
    const double PI = 3.141592653589793;
    const int npairs = nd * T;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
      const int pr = p0 + lane;
      if (pr < npairs) {
        const int d = pr / T, t = pr - d * T;
        const double* dt = sh.det[d];
        const double pcx = sh.tq[t][0], pcy = sh.tq[t][1], valid = sh.tq[t][2], vx = sh.tq[t][3], vy = sh.tq[t][4];
        double io = iou_xyxy(dt, sh.tbox[t]);
        double dx = (dt[0] + dt[2]) / 2.0 - pcx, dy = (dt[1] + dt[3]) / 2.0 - pcy;
        double norm = sqrt(dx * dx + dy * dy) + 1e-6;
        double X = dx / norm, Y = dy / norm;
        double c = vx * X + vy * Y;
        c = fmin(fmax(c, -1.0), 1.0);
        double ang = (PI / 2.0 - fabs(acos(c))) / PI;
        double ac = ((valid * ang) * p.inertia) * dt[4];
        sh.iou[d][t] = io;
        sh.cost[d][t] = -(io + ac);
      }
    }
  }
  __syncthreads();
  int colsum = 0;
  if (lane < T)
    for (int d = 0; d < nd; d++) colsum += sh.iou[d][lane] > p.iou_thr ? 1 : 0;
  TRK_MARK(1);   // cost matrix
  if (lane < MAXD) { sh.d2t[lane] = -1; sh.rej[lane] = 0; sh.taken[lane] = 0; }
  __syncthreads();
  const int maxcol = wave_max_i32(colsum);
  int maxrow = 0;
  for (int d = 0; d < nd; d++) {
    unsigned long long m = __ballot(lane < T && sh.iou[d][lane] > p.iou_thr);
    maxrow = max(maxrow, __popcll(m));
  }
  int my_det = -1;  // detection matched to this lane's tracker
  if (nd > 0 && T > 0) {
    if (maxrow == 1 && maxcol == 1) {
      if (lane < T)
        for (int d = 0; d < nd; d++)
          if (sh.iou[d][lane] > p.iou_thr) my_det = d;
    } else {
      if (nd <= T) {
        lap_solve(&sh.cost[0][0], MAXT, false, nd, T, sh.r2c, sh.lap, lane);
        if (lane < T)
          for (int d = 0; d < nd; d++)
            if (sh.r2c[d] == lane) my_det = d;
      } else {
        lap_solve(&sh.cost[0][0], MAXT, true, T, nd, sh.r2c, sh.lap, lane);
        if (lane < T) my_det = sh.r2c[lane];
      }
    }
  }
  TRK_MARK(2);   // assignment
  // d2t[d]: tracker position the solver paired with detection d (-1 none); rej[d]: pair rejected (IoU < thr)
  const bool was_paired = lane < T && my_det >= 0;   // the solver paired this lane's tracker with a detection (accepted or not)
  if (was_paired) {
    sh.d2t[my_det] = lane;
    if (sh.iou[my_det][lane] < p.iou_thr) { sh.rej[my_det] = 1; my_det = -1; }
  }
  __syncthreads();
  if (lane < T && my_det >= 0) trk_update(st.trk[slot], sh.det[my_det], q44, q66, p.delta_t);
  TRK_MARK(11);  // matched Kalman updates
  // unmatched lists in the reference's order (it matters: the second association sees exact ties):
  //   detections: never paired ascending, then rejected pairs in matched (= detection) order
  //   trackers  : never paired ascending, then the trackers of the rejected pairs in the same order
  // lane = detection for the detection list and the rejected pairs, lane = tracker for the never-paired trackers: positions are
  // population counts of ballots (the lists used to be walked by lane 0, one dependent LDS round trip per element)
  {
    const unsigned long long below = (1ull << lane) - 1ull;
    const int dt_ = lane < nd ? sh.d2t[lane] : 0;
    const bool un_d = lane < nd && dt_ < 0, rj_d = lane < nd && dt_ >= 0 && sh.rej[lane] != 0;
    const unsigned long long mU = __ballot(un_d), mR = __ballot(rj_d), mT = __ballot(lane < T && !was_paired);
    const int nU = __popcll(mU), nR = __popcll(mR), nT = __popcll(mT);
    if (un_d) sh.um_d[__popcll(mU & below)] = lane;
    if (rj_d) { const int q = __popcll(mR & below); sh.um_d[nU + q] = lane; sh.um_t[nT + q] = dt_; }
    if (lane < T && !was_paired) sh.um_t[__popcll(mT & below)] = lane;
    if (lane == 0) { sh.n_um_d = nU + nR; sh.n_um_t = nT + nR; }
  }
  __syncthreads();
  TRK_MARK(3);   // matched updates + unmatched lists
  // ---- observation-centric recovery (second association on the last observations) ----
  int nud = sh.n_um_d, nut = sh.n_um_t;
  bool recovered = false;
  if (nud > 0 && nut > 0) {
    double mx = -1e300;
    for (int p0 = 0; p0 < nud * nut; p0 += 64) {   // (unmatched detection, unmatched tracker) pairs dealt to the lanes
      const int pr = p0 + lane;
      if (pr < nud * nut) {
        const int i = pr / nut, j = pr - i * nut;
        const Trk& k = st.trk[st.order[sh.um_t[j]]];
        double lb[4];
        for (int q = 0; q < 4; q++) lb[q] = k.last_obs[q];
        const double* dt = sh.det[sh.um_d[i]];
        double v = p.asso == 1 ? diou_xyxy(dt, lb) : iou_xyxy(dt, lb);
        sh.iou[i][j] = v;
        sh.cost[i][j] = -v;
        mx = fmax(mx, v);
      }
    }
    mx = wave_max_f64(mx);
    __syncthreads();
    TRK_MARK(12);  // second association: cost entries
    if (mx > p.iou_thr) {
      int mine = -1;  // index into um_d matched to um_t[lane]
      if (nud <= nut) {
        lap_solve(&sh.cost[0][0], MAXT, false, nud, nut, sh.r2c, sh.lap, lane);
        if (lane < nut)
          for (int i = 0; i < nud; i++)
            if (sh.r2c[i] == lane) mine = i;
      } else {
        lap_solve(&sh.cost[0][0], MAXT, true, nut, nud, sh.r2c, sh.lap, lane);
        if (lane < nut) mine = sh.r2c[lane];
      }
      TRK_MARK(13);  // second association: assignment
      if (lane < nut && mine >= 0 && !(sh.iou[mine][lane] < p.iou_thr)) {
        int tp = sh.um_t[lane];
        trk_update(st.trk[st.order[tp]], sh.det[sh.um_d[mine]], q44, q66, p.delta_t);
        sh.taken[sh.um_d[mine]] = 1;
        sh.um_t[lane] = -1;
      }
      recovered = true;
      __syncthreads();
      TRK_MARK(14);  // second association: recovered tracks' Kalman updates (incl. the re-update over the gap)
      {  // np.setdiff1d: sorted ascending (lane = detection)
        const bool left = lane < nd && (sh.d2t[lane] < 0 || sh.rej[lane] != 0) && !sh.taken[lane];
        const unsigned long long mL = __ballot(left);
        __syncthreads();     // every lane has read the old list entries it needs (um_d is rewritten in place)
        if (left) sh.um_d[__popcll(mL & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) sh.n_um_d = __popcll(mL);
      }
      __syncthreads();
    }
  }
  (void)recovered;
  TRK_MARK(4);   // second association
  // ---- unmatched trackers: update(None) ----
  if (lane < nut && sh.um_t[lane] >= 0) trk_update(st.trk[st.order[sh.um_t[lane]]], nullptr, q44, q66, p.delta_t);
  __syncthreads();
  // ---- births ----
  nud = sh.n_um_d;
  if (lane == 0) {
    for (int i = 0; i < nud; i++) {
      if (st.ntrk >= MAXT) { st.overflow += 1; continue; }
      int s = __ffsll((long long)~st.used) - 1;
      st.used |= 1ull << s;
      Trk& k = st.trk[s];
      const double* dt = sh.det[sh.um_d[i]];
      double z[4];
      bbox_to_z(dt, z);
      for (int j = 0; j < 7; j++) k.x[j] = j < 4 ? z[j] : 0.0;
      for (int b = 0; b < 3; b++) { k.B[b][0] = 10.0; k.B[b][1] = 0.0; k.B[b][2] = 0.0; k.B[b][3] = 10000.0; }
      k.Pr = 10.0;
      k.has_saved = 0; k.observed = 0; k.gap = 0; k.has_obs = 0; k.has_vel = 0;
      for (int j = 0; j < 5; j++) k.last_obs[j] = -1.0;
      for (int j = 0; j < 4; j++) { k.obs_age[j] = -1; k.last_z[j] = 0.0; }
      k.vel[0] = k.vel[1] = 0.0;
      k.time_since_update = 0; k.hits = 0; k.hit_streak = 0; k.age = 0; k.nrows = 0;
      k.cum = 0.0; k.cum_c = 0.0; k.prev_x = 0.0; k.prev_y = 0.0;
      k.conf = dt[4]; k.cls = dt[5];
      k.id = st.next_id++;
      st.order[st.ntrk++] = s;
    }
  }
  __syncthreads();
  TRK_MARK(5);   // update(None) + births
  // ---- emission (reverse list order) + deletion ----
  T = st.ntrk;
  bool emit = false, keep = true;
  int myslot = 0;
  if (lane < T) {
    myslot = st.order[lane];
    const Trk& k = st.trk[myslot];
    emit = k.time_since_update < 1 && (k.hit_streak >= p.min_hits || st.frame_count <= p.min_hits);
    keep = !(k.time_since_update > p.max_age);
  }
  unsigned long long em = __ballot(emit);
  const int nem = __popcll(em);
  const int base = st.nrows;
  if (emit) {
    Trk& k = st.trk[myslot];
    int ridx = lane >= 63 ? 0 : __popcll(em >> (lane + 1));  // rows of later trackers come first
    double bx[4];
    if (obs_sum_negative(k)) x_to_bbox(k.x, bx);
    else { bx[0] = k.last_obs[0]; bx[1] = k.last_obs[1]; bx[2] = k.last_obs[2]; bx[3] = k.last_obs[3]; }
    double xc = (bx[0] + bx[2]) / 2.0, yc = (bx[1] + bx[3]) / 2.0;
    if (ridx < MAXD) {
      double* lo = st.last_out[ridx];
      lo[0] = bx[0]; lo[1] = bx[1]; lo[2] = bx[2]; lo[3] = bx[3];
      lo[4] = (double)(k.id + 1); lo[5] = k.cls; lo[6] = k.conf; lo[7] = k.x[4]; lo[8] = k.x[5];
    }
    if (base + ridx < rows_cap) {
      Row r;
      r.id = k.id + 1; r.time = time; r.x = xc; r.y = yc; r.dx = k.x[4]; r.dy = k.x[5];
      r.h = fabs(bx[3] - bx[1]); r.w = fabs(bx[2] - bx[0]);
      rows[base + ridx] = r;
    }
    // running path length of this id (reference track.py:109-113: sqrt(dx^2+dy^2), cumulative per id)
    if (k.nrows > 0) {
      double ex = xc - k.prev_x, ey = yc - k.prev_y;
      double dist = sqrt(ex * ex + ey * ey);
      double yk = dist - k.cum_c;
      double tk = k.cum + yk;
      k.cum_c = (tk - k.cum) - yk;
      k.cum = tk;
    }
    k.prev_x = xc; k.prev_y = yc; k.nrows += 1;
  }
  if (lane == 0) {
    st.last_n = min(nem, MAXD);
    if (base + nem > rows_cap) { st.rows_overflow += base + nem - rows_cap; st.nrows = rows_cap; }
    else st.nrows = base + nem;
  }
  unsigned long long km = __ballot(lane < T && keep);
  if (km != (T >= 64 ? ~0ull : ((1ull << T) - 1ull))) {
    __syncthreads();
    if (lane < T && keep) st.order[__popcll(km & ((1ull << lane) - 1ull))] = myslot;
    {
      unsigned long long dm = ~km & (T >= 64 ? ~0ull : ((1ull << T) - 1ull)), clr = 0ull;
      while (dm) {
        const int l = __ffsll((long long)dm) - 1;
        dm &= dm - 1;
        clr |= 1ull << __shfl(myslot, l);
      }
      if (lane == 0) st.used &= ~clr;
    }
    // a finished track competes for the export id (max cumulative distance; ties -> lower id);
    // the few deaths of a frame are serialised
    unsigned long long dead = ~km & (T >= 64 ? ~0ull : ((1ull << T) - 1ull));
    while (dead) {
      int l = __ffsll((long long)dead) - 1;
      dead &= dead - 1;
      if (lane == l) {
        const Trk& k = st.trk[myslot];
        if (k.nrows >= 2 && (k.cum > st.best_cum || (k.cum == st.best_cum && k.id + 1 < st.best_id))) { st.best_cum = k.cum; st.best_id = k.id + 1; }
      }
      __syncthreads();
    }
    if (lane == 0) st.ntrk = __popcll(km);
  }
  __syncthreads();
  TRK_MARK(6);   // emission + deletion
#ifdef VBT_TRK_PROF
  if (threadIdx.x == 0) { atomicAdd(&g_trk_prof[8], 1ull); atomicAdd(&g_trk_prof[9], (unsigned long long)nd); atomicAdd(&g_trk_prof[10], (unsigned long long)st.ntrk); }
#endif
}

// Frames come either as double detections (host-provided, OCSort.update call shape) ...
__global__ __launch_bounds__(64) void tracker_kernel(ClipState* states, Row* rows, int rows_cap, const double* dets,
                                                     const int* counts, const double* times, int F, int nclips, TrackParams p,
                                                     double q44, double q66) {
  __shared__ StepShared sh;
  const int clip = blockIdx.x, lane = threadIdx.x;
  ClipState& st = states[clip];
  Row* myrows = rows + (size_t)clip * rows_cap;
  for (int f = 0; f < F; f++) {
    const int n = counts[(size_t)f * nclips + clip];
    if (n <= 0) continue;  // reference track.py:180-181: the tracker is not stepped on empty frames
    const double* d = dets + ((size_t)f * nclips + clip) * MAXD * 6;
    __syncthreads();
    if (lane == 0) {  // dets = dets[confs > det_thresh]
      int m = 0;
      for (int i = 0; i < n && i < MAXD; i++)
        if (d[i * 6 + 4] > p.det_thresh) { for (int j = 0; j < 6; j++) sh.det[m][j] = d[i * 6 + j]; m++; }
      sh.flag = m;
    }
    __syncthreads();
    ocsort_step(st, myrows, rows_cap, sh, sh.flag, times[(size_t)f * nclips + clip], p, q44, q66, lane);
  }
}

// The reference's own call shape - tracker.update(dets, []) once per frame on ONE clip, then tracker.trackers[i].id / .kf.x
// (track.py:186-199): the frame's detections and its time stamp travel in the kernel arguments (no allocation, no host-to-device
// copy), and what the caller reads back afterwards - update()'s rows and every live tracker's id + kf.x - is packed into one small
// block that comes back with ONE stream-ordered copy into pinned memory (the per-call path used to cost three hipMalloc, three
// blocking copies and two device-wide synchronisations).
struct OneFrameArg {
  double det[MAXD][6];
  double time;
  int n;
};
constexpr int VIEW_DOUBLES = 2 + MAXD * 9 + MAXT * 8;   // last_n, ntrk | last_out[25][9] | per tracker: id, x[7]
__global__ __launch_bounds__(64) void tracker_one_kernel(ClipState* states, Row* rows, int rows_cap, int clip, OneFrameArg a, TrackParams p,
                                                         double q44, double q66, double* view) {
  __shared__ StepShared sh;
  const int lane = threadIdx.x;
  ClipState& st = states[clip];
  if (lane == 0) {  // dets = dets[confs > det_thresh]
    int m = 0;
    for (int i = 0; i < a.n && i < MAXD; i++)
      if (a.det[i][4] > p.det_thresh) { for (int j = 0; j < 6; j++) sh.det[m][j] = a.det[i][j]; m++; }
    sh.flag = m;
  }
  __syncthreads();
  ocsort_step(st, rows + (size_t)clip * rows_cap, rows_cap, sh, sh.flag, a.time, p, q44, q66, lane);
  __syncthreads();
  if (lane == 0) { view[0] = (double)st.last_n; view[1] = (double)st.ntrk; }
  for (int i = lane; i < MAXD * 9; i += 64) view[2 + i] = (&st.last_out[0][0])[i];
  if (lane < st.ntrk) {
    const Trk& k = st.trk[st.order[lane]];
    double* o = view + 2 + MAXD * 9 + lane * 8;
    o[0] = (double)k.id;
    for (int j = 0; j < 7; j++) o[1 + j] = k.x[j];
  }
}

// ... or straight from the detector's device outputs (fused pipeline): applies the detection
// threshold of reference odt.py:70-75 and the reorder of odt.py:102-118.
// slot = position in the detector batch; clip = tracker state it feeds (map == nullptr: the same index; a negative entry
// or a negative time: the slot carries no frame in this step)
// The per-step metadata (frame time and clip of every slot) travels as a kernel argument: no host-to-device copy, no
// host buffer that has to outlive the call.  64 slots per launch; larger batches take several launches (slot0).
constexpr int META_SLOTS = 64;
struct StepMeta {
  double time[META_SLOTS];
  int clip[META_SLOTS];
};
// One detector slot -> the tracker's detection list, lane i = detection i (the slot's 25 scores / boxes arrive in one
// round trip instead of 25 dependent ones by lane 0): threshold of reference odt.py:70-75 (score >= det_threshold), the
// reorder of odt.py:102-118 and OC-SORT's own gate (score > det_thresh), order kept.  Returns the number of detections
// handed to the tracker, or -1 when run_odt would have returned [] (the frame is skipped, track.py:180-181).  Uniform.
// The slot's count, this lane's score and this lane's box are requested TOGETHER (the count used to gate the score load and the score
// the box load: three dependent round trips at the head of every frame of a walk), and a walk requests frame f + 1's while it steps
// through frame f.
struct RawDet {
  int n;
  float s;
  float4 b;   // ymin,xmin,ymax,xmax
};
__device__ __forceinline__ RawDet fetch_slot_detections(const float* boxes, const float* scores, const int* counts, int slot, int lane) {
  const int l = min(lane, MAXD - 1);   // lanes past the 25 entries re-read the last one (never used)
  RawDet d;
  d.n = counts[slot];
  d.s = scores[slot * MAXD + l];
  d.b = *(const float4*)(boxes + ((size_t)slot * MAXD + l) * 4);
  return d;
}
__device__ __forceinline__ int put_slot_detections(StepShared& sh, const RawDet& d, float det_threshold, double det_thresh, int lane) {
  const int n = min(d.n, MAXD);
  const float s = lane < n ? d.s : 0.0f;
  const bool kept = lane < n && s >= det_threshold;
  const bool used = kept && (double)s > det_thresh;
  const unsigned long long mk = __ballot(kept), mu = __ballot(used);
  if (used) {
    const int m = __popcll(mu & ((1ull << lane) - 1ull));
    sh.det[m][0] = (double)d.b.y; sh.det[m][1] = (double)d.b.x; sh.det[m][2] = (double)d.b.w; sh.det[m][3] = (double)d.b.z;
    sh.det[m][4] = (double)s; sh.det[m][5] = 0.0;
  }
  return mk ? __popcll(mu) : -1;
}
__device__ inline int load_slot_detections(StepShared& sh, const float* boxes, const float* scores, const int* counts, int slot,
                                           float det_threshold, double det_thresh, int lane) {
  return put_slot_detections(sh, fetch_slot_detections(boxes, scores, counts, slot, lane), det_threshold, det_thresh, lane);
}

__global__ __launch_bounds__(64) void tracker_from_det_kernel(ClipState* states, Row* rows, int rows_cap, const float* boxes,
                                                              const float* scores, const int* counts, StepMeta meta, int slot0,
                                                              float det_threshold, TrackParams p, double q44, double q66) {
  __shared__ StepShared sh;
  const int slot = slot0 + blockIdx.x, lane = threadIdx.x;
  const int clip = meta.clip[blockIdx.x];
  const double frame_time = meta.time[blockIdx.x];
  if (clip < 0 || !(frame_time >= 0.0)) return;
  ClipState& st = states[clip];
  const int nd = load_slot_detections(sh, boxes, scores, counts, slot, det_threshold, p.det_thresh, lane);
  __syncthreads();
  if (nd < 0) return;
  ocsort_step(st, rows + (size_t)clip * rows_cap, rows_cap, sh, nd, frame_time, p, q44, q66, lane);
}

// Time-batched form (the reference's unit of work is ONE video, track.py:85-126,159-247): the detector batch holds RUNS of
// consecutive frames of a clip instead of one frame of each of B clips - the detector is stateless, so a single clip fills
// the whole batch.  One wavefront per run walks its frames in order (frame f of the run sits in detector slot
// slot0 + f * slot_stride); the frame time is frame_count / fps (track.py:161,169) with frame_count = frame0 + f * frame_step,
// one IEEE division like the reference's.  Run descriptors travel in the kernel arguments (64 runs per launch).
constexpr int META_RUNS = 64;
struct RunMeta {
  int clip, slot0, slot_stride, n_frames, frame0, frame_step;
  double fps;
};
struct SeqMeta {
  RunMeta run[META_RUNS];
};
// A run of SEQ_LDS_MIN frames or more keeps the clip's tracker state in LDS for the whole walk: the header and the live
// tracks (a few KB) are copied in once and written back once, and every Kalman / association step in between works on LDS
// instead of on dependent global round trips (one clip alone: 21 us -> see DESIGN.md per frame).
constexpr int SEQ_LDS_MIN = 6;
__device__ inline void copy_words(void* dst, const void* src, int bytes, int lane) {   // 8-byte words, one wavefront
  unsigned long long* d = (unsigned long long*)dst;
  const unsigned long long* s_ = (const unsigned long long*)src;
  for (int i = lane; i < bytes / 8; i += 64) d[i] = s_[i];
}
static_assert(sizeof(Trk) % 8 == 0 && offsetof(ClipState, trk) % 8 == 0, "8-byte copy granularity");
// The clip state between global memory and its LDS copy, by one wavefront: the header, then the LIVE tracks only.  All loads of a pass
// are in flight together (the header in one round trip, the tracks four words per lane at a time): the per-track loop it replaces
// waited for every 512 bytes - 5.8 us for ten tracks - which only a long run could amortise.  slots = 64 ints of LDS scratch.
template <bool TO_LDS>
__device__ inline void clip_state_copy(ClipState* lst, ClipState* gst, int* slots, int lane) {
  constexpr int HW = (int)(offsetof(ClipState, trk) / 8), HI = (HW + 63) / 64, TW = (int)(sizeof(Trk) / 8);
  unsigned long long* l = (unsigned long long*)lst;
  unsigned long long* g = (unsigned long long*)gst;
  {
    unsigned long long hv[HI];
#pragma unroll
    for (int k = 0; k < HI; k++) { const int i = min(lane + 64 * k, HW - 1); hv[k] = TO_LDS ? g[i] : l[i]; }
#pragma unroll
    for (int k = 0; k < HI; k++) { const int i = lane + 64 * k; if (i < HW) (TO_LDS ? l : g)[i] = hv[k]; }
  }
  __syncthreads();
  const unsigned long long used = lst->used;
  if ((used >> lane) & 1ull) slots[__popcll(used & ((1ull << lane) - 1ull))] = lane;
  __syncthreads();
  const int total = __popcll(used) * TW;
  unsigned long long* lt = (unsigned long long*)&lst->trk[0];
  unsigned long long* gt = (unsigned long long*)&gst->trk[0];
  for (int base = 0; base < total; base += 256) {
    unsigned long long v[4];
    int off[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = min(base + lane + 64 * k, total - 1);
      const int ord = i / TW, w = i - ord * TW;
      off[k] = slots[ord] * TW + w;
      v[k] = TO_LDS ? gt[off[k]] : lt[off[k]];
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (base + lane + 64 * k < total) (TO_LDS ? lt : gt)[off[k]] = v[k];
  }
  __syncthreads();
}
__global__ __launch_bounds__(64) void tracker_seq_kernel(ClipState* states, Row* rows, int rows_cap, const float* boxes,
                                                         const float* scores, const int* counts, SeqMeta meta,
                                                         float det_threshold, TrackParams p, double q44, double q66, int lds_state) {
  __shared__ StepShared sh;
  extern __shared__ __attribute__((aligned(16))) unsigned char seq_dyn[];   // ClipState copy (only when lds_state != 0)
  const int lane = threadIdx.x;
  const RunMeta r = meta.run[blockIdx.x];
  if (r.clip < 0) return;
  ClipState* gst = &states[r.clip];
  ClipState* st = gst;
  const bool cached = lds_state != 0 && r.n_frames >= lds_state;   // lds_state = shortest run that is worth the copy in and out
  if (cached) {
    ClipState* lst = (ClipState*)seq_dyn;
    clip_state_copy<true>(lst, gst, sh.um_t, lane);
    st = lst;
  }
  Row* myrows = rows + (size_t)r.clip * rows_cap;
  // Two copies of the walk, one per home of the state: in each the compiler knows the address space of every access to the clip state
  // (LDS: ds_read / ds_write; global memory: global_load / global_store).  With one copy on a pointer that may be either, every
  // access was a FLAT instruction - slower to issue, and each one counted on both wait counters, so that every wait drained both
  // queues (the walk was 5 600 instructions with 270 flat accesses and 240 waits).
  auto walk = [&](ClipState& state) {
    RawDet cur = fetch_slot_detections(boxes, scores, counts, r.slot0, lane);
    for (int f = 0; f < r.n_frames; f++) {
      // frame f + 1's detections are requested now and looked at in the next iteration (the last frame re-requests itself)
      const RawDet nxt = fetch_slot_detections(boxes, scores, counts, r.slot0 + min(f + 1, r.n_frames - 1) * r.slot_stride, lane);
      __syncthreads();  // the previous frame's readers of sh.det are done
      const int nd = put_slot_detections(sh, cur, det_threshold, p.det_thresh, lane);
      __syncthreads();
      cur = nxt;
      if (nd < 0) continue;
      const double frame_time = (double)(r.frame0 + f * r.frame_step) / r.fps;
      ocsort_step(state, myrows, rows_cap, sh, nd, frame_time, p, q44, q66, lane);
    }
  };
  if (cached) walk(*(ClipState*)seq_dyn);
  else walk(*gst);
  if (cached) {   // write the state back: header + every slot that is live now (slots freed during the walk need no copy)
    __syncthreads();
    clip_state_copy<false>(st, gst, sh.um_t, lane);
  }
}

// ------------------------------------------------------------------------------------------
// export id selection + rep analysis
// ------------------------------------------------------------------------------------------
struct RollMean {  // pandas roll_mean state (Kahan add / remove)
  double sum, c_add, c_rem, prev;
  long nobs, neg, same;
  __device__ void init() { sum = 0; c_add = 0; c_rem = 0; prev = __builtin_nan(""); nobs = 0; neg = 0; same = 0; }
  __device__ void add(double v) {
    nobs += 1;
    double y = v - c_add;
    double t = sum + y;
    c_add = (t - sum) - y;
    sum = t;
    if (__builtin_signbit(v)) neg += 1;
    if (v == prev) same += 1; else same = 1;
    prev = v;
  }
  __device__ void remove(double v) {
    nobs -= 1;
    double y = -v - c_rem;
    double t = sum + y;
    c_rem = (t - sum) - y;
    sum = t;
    if (__builtin_signbit(v)) neg -= 1;
  }
  __device__ double mean() const {
    double r = sum / (double)nobs;
    if (same >= nobs) r = prev;
    else if (neg == 0 && r < 0) r = 0.0;
    else if (neg == nobs && r > 0) r = 0.0;
    return r;
  }
};

struct VtParams {
  double plate_diameter, diff_threshold, min_distance;
  int preprocess, flush;
};

struct VtState {  // reference VelocityTracker.py:30-48
  int phase, neg, pos, nph, n, has_prev, has_max;
  double y_prev, max_y_diff;
  // the single RunningAverage(30) fed width then height (VelocityTracker.py:44-45,98-99)
  double win[30];
  int whead, wcount;
  double wtotal;
};

__device__ inline double ra_update(VtState& s, double v) {  // reference RunningAverage.py:16-27
  s.win[(s.whead + s.wcount) % 30] = v;
  s.wcount += 1;
  s.wtotal += v;
  if (s.wcount >= 30) {
    double avg = s.wtotal / 30.0;
    s.wtotal -= s.win[s.whead];
    s.whead = (s.whead + 1) % 30;
    s.wcount -= 1;
    return avg;
  }
  return s.wtotal / (double)s.wcount;
}

__device__ inline void vt_filter(VtState& s, double* ph) {  // VelocityTracker.py:50-67
  double thr = s.max_y_diff / 2;
  int o = 0;
  for (int i = 0; i < s.nph; i++) {
    double yd = fabs(ph[i * 6 + 2] - ph[i * 6 + 3]);
    if (!(yd < thr)) {
      if (o != i) for (int j = 0; j < 6; j++) ph[o * 6 + j] = ph[i * 6 + j];
      o++;
    }
  }
  s.nph = o;
}

__device__ inline void vt_end_phase(VtState& s, const VtParams& p, const double* xs, const double* ys, const double* ws,
                                    const double* hs, const double* ts, double* ph) {  // VelocityTracker.py:171-222
  int imax = 0, imin = 0;
  for (int i = 1; i < s.n; i++) {
    if (ys[i] > ys[imax]) imax = i;
    if (ys[i] < ys[imin]) imin = i;
  }
  int st = s.phase == 0 ? imax : imin, en = s.phase == 0 ? imin : imax;
  double y_diff = fabs(ys[st] - ys[en]);
  if (!s.has_max || y_diff > s.max_y_diff) {
    s.max_y_diff = y_diff;
    s.has_max = 1;
    vt_filter(s, ph);
  }
  if (y_diff > s.max_y_diff * p.diff_threshold) {
    double distance = 0.0;
    for (int i = st + 1; i < en + 1; i++) {
      double ddx = fabs(xs[i] - xs[i - 1]) / ((ws[i] + ws[i - 1]) / 2) * p.plate_diameter;
      double ddy = fabs(ys[i] - ys[i - 1]) / ((hs[i] + hs[i - 1]) / 2) * p.plate_diameter;
      distance += ddx + ddy;
    }
    if (distance < p.min_distance) {
      s.neg = 0; s.pos = 0; s.phase = 2;
      return;
    }
    if (s.nph < MAXPH) {
      double* o = ph + s.nph * 6;
      o[0] = ts[st]; o[1] = ts[en]; o[2] = ys[st]; o[3] = ys[en]; o[4] = distance; o[5] = (double)s.phase;
      s.nph += 1;
    }
    vt_filter(s, ph);
  }
  s.phase = 2;
  s.pos = 0; s.neg = 0;
}

// cols: [T][7] = time,x,y,dx,dy,h,w of ONE track.  One lane per clip does the sequential scan.
__device__ void analyze_track(const double* cols, int T, const VtParams& p, double* scratch /*5*T*/, double* ph, int* nph_out) {
  double* xs = scratch; double* ys = xs + T; double* ws = ys + T; double* hs = ws + T; double* ts = hs + T;
  VtState s;
  s.phase = 2; s.neg = 0; s.pos = 0; s.nph = 0; s.n = 0; s.has_prev = 0; s.has_max = 0; s.y_prev = 0; s.max_y_diff = 0;
  s.whead = 0; s.wcount = 0; s.wtotal = 0.0;
  RollMean rx, ry, rh, rw;
  rx.init(); ry.init(); rh.init(); rw.init();
  for (int i = 0; i < T; i++) {
    const double* r = cols + (size_t)i * 7;
    double time = r[0], x = r[1], y = r[2], h = r[5], w = r[6];
    if (p.preprocess) {  // plot.py:90-95 (dx, dy columns are smoothed there too but never used downstream)
      if (i >= 5) { rx.remove(cols[(size_t)(i - 5) * 7 + 1]); ry.remove(cols[(size_t)(i - 5) * 7 + 2]); }
      rx.add(x); ry.add(y); rh.add(h); rw.add(w);
      x = rx.mean(); y = ry.mean(); h = rh.mean(); w = rw.mean();
    }
    // VelocityTracker.process_measurements (VelocityTracker.py:92-158)
    double width = ra_update(s, w);
    double height = ra_update(s, h);
    double dy = r[4];
    if (s.has_prev) dy = y - s.y_prev;
    else if (p.preprocess) dy = r[4];  // first sample: the (smoothed == raw) incoming dy
    if (s.phase != 2) { xs[s.n] = x; ys[s.n] = y; ws[s.n] = width; hs[s.n] = height; ts[s.n] = time; s.n++; }
    if (s.phase == 0) {
      if (dy > 0) { s.pos += 1; s.neg = 0; if (s.pos >= 1) vt_end_phase(s, p, xs, ys, ws, hs, ts, ph); }
      else s.pos = 0;
    }
    if (s.phase == 1) {
      if (dy < 0) { s.neg += 1; s.pos = 0; if (s.neg >= 1) vt_end_phase(s, p, xs, ys, ws, hs, ts, ph); }
      else { s.neg = 0; s.pos += 1; }
    }
    if (dy < 0 && s.phase == 2) {
      s.neg += 1; s.pos = 0;
      if (s.neg == 1) s.n = 0;
      else { xs[s.n] = x; ys[s.n] = y; ws[s.n] = width; hs[s.n] = height; ts[s.n] = time; s.n++; }
      if (s.neg >= 3) { s.phase = 0; s.pos = 0; s.neg = 0; }
    }
    if (dy > 0 && s.phase == 2) {
      s.pos += 1; s.neg = 0;
      if (s.pos == 1) s.n = 0;
      else { xs[s.n] = x; ys[s.n] = y; ws[s.n] = width; hs[s.n] = height; ts[s.n] = time; s.n++; }
      if (s.pos >= 3) { s.phase = 1; s.pos = 0; s.neg = 0; }
    }
    s.y_prev = y; s.has_prev = 1;
  }
  if (p.flush && s.phase != 2) vt_end_phase(s, p, xs, ys, ws, hs, ts, ph);  // end_processing, VelocityTracker.py:224-230
  *nph_out = s.nph;
}

__global__ __launch_bounds__(64) void analyze_kernel(const double* cols, const int* T, int stride_rows, VtParams p, double* scratch,
                                                     double* phases, int* nph) {
  const int clip = blockIdx.x;
  if (threadIdx.x != 0) return;
  analyze_track(cols + (size_t)clip * stride_rows * 7, T[clip], p, scratch + (size_t)clip * stride_rows * 5,
                phases + (size_t)clip * MAXPH * 6, nph + clip);
}

// pandas rolling(window, min_periods=1).mean() (window > 0) / expanding(min_periods=1).mean() (window == 0) of every
// column of a row-major [T][ncols] table; lane = column (plot.py:90-95, kinovea.py:99-105, qualysis.py:113-117).
__global__ __launch_bounds__(64) void window_means_kernel(const double* rows, int T, int ncols, const int* windows, double* out) {
  const int c = threadIdx.x;
  if (c >= ncols) return;
  const int w = windows[c];
  RollMean r;
  r.init();
  for (int i = 0; i < T; i++) {
    double v = rows[(size_t)i * ncols + c];
    if (w >= 0) {
      if (w > 0 && i >= w) r.remove(rows[(size_t)(i - w) * ncols + c]);
      r.add(v);
      v = r.mean();
    }
    out[(size_t)i * ncols + c] = v;
  }
}

// end of clip: live tracks compete for the export id too; then gather the rows of the winner.
__global__ __launch_bounds__(64) void select_gather_kernel(ClipState* states, const Row* rows, int rows_cap, double* cols, int* T,
                                                           int* best_ids) {
  const int clip = blockIdx.x, lane = threadIdx.x;
  ClipState& st = states[clip];
  __shared__ int s_best;
  if (lane == 0) {
    double bc = st.best_cum;
    int bi = st.best_id;
    for (int t = 0; t < st.ntrk; t++) {
      const Trk& k = st.trk[st.order[t]];
      if (k.nrows >= 2 && (k.cum > bc || (k.cum == bc && (bi < 0 || k.id + 1 < bi)))) { bc = k.cum; bi = k.id + 1; }
    }
    s_best = bi;
    best_ids[clip] = bi;
  }
  __syncthreads();
  const int best = s_best;
  const Row* r = rows + (size_t)clip * rows_cap;
  double* c = cols + (size_t)clip * rows_cap * 7;
  const int n = st.nrows;
  int outn = 0;  // stable, ordered gather with ballots
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    bool hit = i < n && best >= 0 && r[i].id == best;
    unsigned long long m = __ballot(hit);
    if (hit) {
      double* o = c + (size_t)(outn + __popcll(m & ((1ull << lane) - 1ull))) * 7;
      o[0] = r[i].time; o[1] = r[i].x; o[2] = r[i].y; o[3] = r[i].dx; o[4] = r[i].dy; o[5] = r[i].h; o[6] = r[i].w;
    }
    outn += __popcll(m);
  }
  if (lane == 0) T[clip] = outn;
}

// Clip close: everything the host reads per clip, packed into one block so that ONE copy fetches it:
//   record c = { int best_id, n_rows, n_phases, overflow ; double phases[cap][6] }
__global__ __launch_bounds__(64) void pack_summary_kernel(const ClipState* states, const int* best, const int* nph, const double* phases,
                                                          int cap, unsigned char* out) {
  const int clip = blockIdx.x, lane = threadIdx.x;
  const size_t rec = 16 + (size_t)cap * 48;
  unsigned char* o = out + clip * rec;
  const int n = nph[clip];
  if (lane == 0) {
    int* h = (int*)o;
    h[0] = best[clip]; h[1] = states[clip].nrows; h[2] = n; h[3] = states[clip].overflow | states[clip].rows_overflow;
  }
  double* ph = (double*)(o + 16);
  const double* src = phases + (size_t)clip * MAXPH * 6;
  for (int i = lane; i < min(n, cap) * 6; i += 64) ph[i] = src[i];
}

__global__ void init_states_kernel(ClipState* states, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ClipState& st = states[i];
  st.ntrk = 0; st.frame_count = 0; st.next_id = 0; st.overflow = 0; st.nrows = 0; st.rows_overflow = 0;
  st.best_id = -1; st.last_n = 0; st.best_cum = -1.0; st.used = 0ull;
}

}  // namespace vbt

using namespace vbt;

struct vbt_tracker {
  int n_clips = 0, rows_cap = 0, device = 0;
  TrackParams p;
  double q44 = 0, q66 = 0;
  ClipState* states = nullptr;
  Row* rows = nullptr;
  double* cols = nullptr;     // [n_clips][rows_cap][7] gathered rows of the export id
  double* scratch = nullptr;  // [n_clips][rows_cap][5]
  double* phases = nullptr;   // [n_clips][MAXPH][6]
  int* nph = nullptr;
  int* T = nullptr;
  int* best = nullptr;
  bool finished = false;
  hipStream_t finish_stream = nullptr;   // stream vbt_tracker_finish ran on: the close waits for it, not for the device
  unsigned char* d_summary = nullptr;    // packed close block (pack_summary_kernel), grown on demand
  unsigned char* h_summary = nullptr;    // its pinned host copy
  size_t summary_bytes = 0;
  double* d_view = nullptr;              // packed read-back block of the one-frame path (tracker_one_kernel)
  double* h_view = nullptr;              // its pinned host copy
  int view_clip = -1;                    // clip whose state h_view mirrors (-1: none: the state changed on another path)
};

extern "C" {

int vbt_tracker_create(int n_clips, int rows_cap, const vbt_tracker_params* prm, int device, vbt_tracker** out) {
  if (!out || !prm || n_clips < 1 || rows_cap < 1) { set_error("vbt_tracker_create: bad argument"); return VBT_ERR_ARG; }
  if (prm->delta_t < 1 || prm->delta_t > 3) { set_error("delta_t must be 1..3"); return VBT_ERR_ARG; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error("vbt_tracker_create: HIP device %d not available (%d visible) - no CPU fallback", device, ndev);
    return VBT_ERR_HIP;
  }
  VBT_HIP_CHECK(hipSetDevice(device));
  vbt_tracker* t = new vbt_tracker();
  t->n_clips = n_clips; t->rows_cap = rows_cap; t->device = device;
  t->p.max_age = prm->max_age; t->p.min_hits = prm->min_hits; t->p.delta_t = prm->delta_t; t->p.asso = prm->asso;
  t->p.iou_thr = prm->iou_threshold; t->p.inertia = prm->inertia; t->p.det_thresh = prm->det_thresh;
  // Q = eye(7); Q[-1,-1] *= 0.01; Q[4:,4:] *= 0.01
  double q = 1.0;
  q *= 0.01;
  t->q44 = q;
  double q6 = 1.0;
  q6 *= 0.01;
  q6 *= 0.01;
  t->q66 = q6;
  auto fail = [&](const char* what) { set_error("hipMalloc failed for %s", what); vbt_tracker_destroy(t); return VBT_ERR_HIP; };
  if (hipMalloc((void**)&t->states, sizeof(ClipState) * n_clips) != hipSuccess) return fail("tracker state");
  if (hipMalloc((void**)&t->rows, sizeof(Row) * (size_t)n_clips * rows_cap) != hipSuccess) return fail("rows");
  if (hipMalloc((void**)&t->cols, sizeof(double) * 7 * (size_t)n_clips * rows_cap) != hipSuccess) return fail("cols");
  if (hipMalloc((void**)&t->scratch, sizeof(double) * 5 * (size_t)n_clips * rows_cap) != hipSuccess) return fail("scratch");
  if (hipMalloc((void**)&t->phases, sizeof(double) * 6 * MAXPH * (size_t)n_clips) != hipSuccess) return fail("phases");
  if (hipMalloc((void**)&t->nph, sizeof(int) * n_clips) != hipSuccess) return fail("nph");
  if (hipMalloc((void**)&t->T, sizeof(int) * n_clips) != hipSuccess) return fail("T");
  if (hipMalloc((void**)&t->best, sizeof(int) * n_clips) != hipSuccess) return fail("best");
  // the clip-close record buffers (device + pinned host) at their largest size: no allocation on the first close
  t->summary_bytes = (16 + (size_t)MAXPH * 48) * n_clips;
  if (hipMalloc((void**)&t->d_summary, t->summary_bytes) != hipSuccess) return fail("summary");
  if (hipHostMalloc((void**)&t->h_summary, t->summary_bytes, hipHostMallocDefault) != hipSuccess) return fail("pinned summary");
  if (hipMalloc((void**)&t->d_view, sizeof(double) * VIEW_DOUBLES) != hipSuccess) return fail("view");
  if (hipHostMalloc((void**)&t->h_view, sizeof(double) * VIEW_DOUBLES, hipHostMallocDefault) != hipSuccess) return fail("pinned view");
  init_states_kernel<<<(n_clips + 63) / 64, 64>>>(t->states, n_clips);
  VBT_HIP_CHECK(hipDeviceSynchronize());
  *out = t;
  return VBT_OK;
}

void vbt_tracker_destroy(vbt_tracker* t) {
  if (!t) return;
  (void)hipFree(t->states); (void)hipFree(t->rows); (void)hipFree(t->cols); (void)hipFree(t->scratch);
  (void)hipFree(t->phases); (void)hipFree(t->nph); (void)hipFree(t->T); (void)hipFree(t->best);
  if (t->d_summary) (void)hipFree(t->d_summary);
  if (t->h_summary) (void)hipHostFree(t->h_summary);
  if (t->d_view) (void)hipFree(t->d_view);
  if (t->h_view) (void)hipHostFree(t->h_view);
  delete t;
}

int vbt_tracker_reset(vbt_tracker* t) {
  if (!t) { set_error("NULL tracker"); return VBT_ERR_ARG; }
  init_states_kernel<<<(t->n_clips + 63) / 64, 64>>>(t->states, t->n_clips);
  VBT_HIP_CHECK(hipDeviceSynchronize());
  t->finished = false;
  t->view_clip = -1;
  return VBT_OK;
}

int vbt_tracker_update(vbt_tracker* t, const double* dets, const int32_t* counts, const double* times, int F) {
  RoctxRange range("vbt:track");
  if (!t || !dets || !counts || !times || F < 1) { set_error("vbt_tracker_update: bad argument"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(t->device));
  t->view_clip = -1;
  if (F == 1 && t->n_clips == 1) {   // OCSort.update(dets, []) of the reference's loop: everything in the kernel arguments
    if (counts[0] <= 0) return VBT_OK;   // (track.py:180-181: the tracker is not stepped on empty frames)
    OneFrameArg a;
    a.n = std::min((int)counts[0], MAXD);
    a.time = times[0];
    memcpy(a.det, dets, sizeof(double) * 6 * a.n);
    tracker_one_kernel<<<1, 64, 0, nullptr>>>(t->states, t->rows, t->rows_cap, 0, a, t->p, t->q44, t->q66, t->d_view);
    VBT_HIP_CHECK(hipGetLastError());
    VBT_HIP_CHECK(hipMemcpyAsync(t->h_view, t->d_view, sizeof(double) * VIEW_DOUBLES, hipMemcpyDeviceToHost, nullptr));
    VBT_HIP_CHECK(hipStreamSynchronize(nullptr));
    t->view_clip = 0;
    t->finished = false;
    return VBT_OK;
  }
  size_t nd = (size_t)F * t->n_clips;
  double* dd = nullptr; int* dc = nullptr; double* dt = nullptr;
  VBT_HIP_CHECK(hipMalloc((void**)&dd, nd * MAXD * 6 * sizeof(double)));
  VBT_HIP_CHECK(hipMalloc((void**)&dc, nd * sizeof(int)));
  VBT_HIP_CHECK(hipMalloc((void**)&dt, nd * sizeof(double)));
  VBT_HIP_CHECK(hipMemcpy(dd, dets, nd * MAXD * 6 * sizeof(double), hipMemcpyHostToDevice));
  VBT_HIP_CHECK(hipMemcpy(dc, counts, nd * sizeof(int), hipMemcpyHostToDevice));
  VBT_HIP_CHECK(hipMemcpy(dt, times, nd * sizeof(double), hipMemcpyHostToDevice));
  tracker_kernel<<<t->n_clips, 64>>>(t->states, t->rows, t->rows_cap, dd, dc, dt, F, t->n_clips, t->p, t->q44, t->q66);
  hipError_t e = hipDeviceSynchronize();
  (void)hipFree(dd); (void)hipFree(dc); (void)hipFree(dt);
  if (e != hipSuccess) { set_error("tracker kernel failed: %s", hipGetErrorString(e)); return VBT_ERR_HIP; }
  t->finished = false;
  return VBT_OK;
}

// One tracker step for slots [0, n_slots): launches of at most META_SLOTS workgroups, metadata in the kernel arguments.
static int launch_steps(vbt_tracker* t, const float* boxes_dev, const float* scores_dev, const int32_t* counts_dev, const int32_t* clip_of_slot,
                        const double* times, int n_slots, float det_threshold, hipStream_t st) {
  RoctxRange range("vbt:track");
  if (((uintptr_t)boxes_dev & 15) != 0) { set_error("tracker update: boxes_dev must be 16-byte aligned"); return VBT_ERR_ARG; }
  for (int s0 = 0; s0 < n_slots; s0 += META_SLOTS) {
    const int nb = std::min(META_SLOTS, n_slots - s0);
    StepMeta meta;
    for (int i = 0; i < nb; i++) {
      meta.time[i] = times[s0 + i];
      meta.clip[i] = clip_of_slot ? clip_of_slot[s0 + i] : s0 + i;
    }
    tracker_from_det_kernel<<<nb, 64, 0, st>>>(t->states, t->rows, t->rows_cap, boxes_dev, scores_dev, counts_dev, meta, s0, det_threshold,
                                               t->p, t->q44, t->q66);
  }
  VBT_HIP_CHECK(hipGetLastError());
  t->finished = false;
  t->view_clip = -1;
  return VBT_OK;
}

int vbt_tracker_update_from_detections(vbt_tracker* t, const float* boxes_dev, const float* scores_dev, const int32_t* counts_dev,
                                       const double* times_host, float det_threshold, void* stream) {
  if (!t || !boxes_dev || !scores_dev || !counts_dev || !times_host) { set_error("bad argument"); return VBT_ERR_ARG; }
  return launch_steps(t, boxes_dev, scores_dev, counts_dev, nullptr, times_host, t->n_clips, det_threshold, (hipStream_t)stream);
}

int vbt_tracker_update_from_slots(vbt_tracker* t, const float* boxes_dev, const float* scores_dev, const int32_t* counts_dev,
                                  const int32_t* clip_of_slot_host, const double* times_host, int n_slots, float det_threshold,
                                  void* stream) {
  if (!t || !boxes_dev || !scores_dev || !counts_dev || !clip_of_slot_host || !times_host || n_slots < 1 || n_slots > t->n_clips) {
    set_error("vbt_tracker_update_from_slots: bad argument (n_slots must be in [1, n_clips])");
    return VBT_ERR_ARG;
  }
  for (int i = 0; i < n_slots; i++) {
    if (clip_of_slot_host[i] >= t->n_clips) { set_error("slot %d -> clip %d, tracker has %d clips", i, clip_of_slot_host[i], t->n_clips); return VBT_ERR_ARG; }
    for (int j = 0; j < i; j++)
      if (clip_of_slot_host[i] >= 0 && clip_of_slot_host[i] == clip_of_slot_host[j]) { set_error("clip %d sits in two slots", clip_of_slot_host[i]); return VBT_ERR_ARG; }
  }
  return launch_steps(t, boxes_dev, scores_dev, counts_dev, clip_of_slot_host, times_host, n_slots, det_threshold, (hipStream_t)stream);
}

int vbt_tracker_update_from_detections_seq(vbt_tracker* t, const float* boxes_dev, const float* scores_dev, const int32_t* counts_dev,
                                           int n_slots, const vbt_run* runs_host, int n_runs, float det_threshold, void* stream) {
  RoctxRange range("vbt:track");
  if (!t || !boxes_dev || !scores_dev || !counts_dev || !runs_host || n_runs < 1 || n_slots < 1) {
    set_error("vbt_tracker_update_from_detections_seq: bad argument");
    return VBT_ERR_ARG;
  }
  if (((uintptr_t)boxes_dev & 15) != 0) { set_error("vbt_tracker_update_from_detections_seq: boxes_dev must be 16-byte aligned"); return VBT_ERR_ARG; }
  // every run stays inside the detector batch and no clip appears twice (two wavefronts would step one Kalman state)
  std::vector<char> seen((size_t)t->n_clips, 0);
  for (int i = 0; i < n_runs; i++) {
    const vbt_run& r = runs_host[i];
    if (r.clip < 0) continue;  // an empty descriptor
    if (r.clip >= t->n_clips) { set_error("run %d: clip %d, tracker has %d clips", i, r.clip, t->n_clips); return VBT_ERR_ARG; }
    if (seen[r.clip]) { set_error("run %d: clip %d appears in two runs of one call", i, r.clip); return VBT_ERR_ARG; }
    seen[r.clip] = 1;
    if (r.n_frames < 1 || r.frame0 < 1 || r.frame_step < 1 || !(r.fps > 0.0)) { set_error("run %d: n_frames, frame0, frame_step >= 1 and fps > 0 required", i); return VBT_ERR_ARG; }
    const long long last = (long long)r.slot0 + (long long)(r.n_frames - 1) * r.slot_stride;
    if (r.slot0 < 0 || r.slot0 >= n_slots || last < 0 || last >= n_slots) {
      set_error("run %d: slots %d..%lld outside the detector batch of %d", i, r.slot0, last, n_slots);
      return VBT_ERR_ARG;
    }
    if ((long long)r.frame0 + (long long)(r.n_frames - 1) * r.frame_step > 0x7fffffffLL) { set_error("run %d: frame number overflow", i); return VBT_ERR_ARG; }
  }
  hipStream_t st = (hipStream_t)stream;
  VBT_HIP_CHECK(hipSetDevice(t->device));     // (the LDS opt-in below is a per-device function attribute)
  for (int r0 = 0; r0 < n_runs; r0 += META_RUNS) {
    const int nb = std::min(META_RUNS, n_runs - r0);
    SeqMeta meta;
    for (int i = 0; i < nb; i++) {
      const vbt_run& r = runs_host[r0 + i];
      meta.run[i] = RunMeta{r.clip, r.slot0, r.slot_stride, r.n_frames, r.frame0, r.frame_step, r.fps};
    }
    int longest = 0;
    for (int i = 0; i < nb; i++) longest = std::max(longest, meta.run[i].clip >= 0 ? meta.run[i].n_frames : 0);
    static const bool lds_off = getenv("VBT_SEQ_NO_LDS") != nullptr;
    static const int lds_min = getenv("VBT_SEQ_LDS_MIN") ? std::max(1, atoi(getenv("VBT_SEQ_LDS_MIN"))) : SEQ_LDS_MIN;
    int lds_state = (longest >= lds_min && !lds_off) ? lds_min : 0;
    if (lds_state) {
      // opt in to the dynamic LDS of the cached state once per device; if the runtime refuses, the walk works on global memory
      static int attr_state[64] = {0};   // per device: 0 = not tried, 1 = granted, -1 = refused
      const int di = t->device < 64 ? t->device : 63;
      if (attr_state[di] == 0)
        attr_state[di] = hipFuncSetAttribute(reinterpret_cast<const void*>(&tracker_seq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)sizeof(ClipState)) == hipSuccess ? 1 : -1;
      if (attr_state[di] < 0) lds_state = 0;
    }
    tracker_seq_kernel<<<nb, 64, lds_state ? sizeof(ClipState) : 0, st>>>(t->states, t->rows, t->rows_cap, boxes_dev, scores_dev, counts_dev, meta,
                                                                           det_threshold, t->p, t->q44, t->q66, lds_state);
  }
  VBT_HIP_CHECK(hipGetLastError());
  t->finished = false;
  t->view_clip = -1;
  return VBT_OK;
}

#ifdef VBT_TRK_PROF
int vbt_tracker_prof_read(unsigned long long* out16, int reset) {
  VBT_HIP_CHECK(hipDeviceSynchronize());
  VBT_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_trk_prof), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    VBT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trk_prof), z, sizeof(z)));
  }
  return VBT_OK;
}
#endif

static int fetch_state_header(vbt_tracker* t, int clip, ClipState* hdr_only) {
  // copies only the leading scalars + order + used (not the tracker array)
  VBT_HIP_CHECK(hipMemcpy(hdr_only, &t->states[clip], offsetof(ClipState, last_out), hipMemcpyDeviceToHost));
  return VBT_OK;
}

int vbt_tracker_last_output(vbt_tracker* t, int clip, double* out7, double* vel2, int cap, int* M) {
  if (!t || !out7 || !vel2 || !M || clip < 0 || clip >= t->n_clips) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (t->view_clip == clip) {   // the one-frame path brought it back already
    const int n = std::min((int)t->h_view[0], cap);
    for (int i = 0; i < n; i++) {
      const double* lo = t->h_view + 2 + i * 9;
      for (int j = 0; j < 7; j++) out7[i * 7 + j] = lo[j];
      vel2[i * 2] = lo[7];
      vel2[i * 2 + 1] = lo[8];
    }
    *M = n;
    return VBT_OK;
  }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  std::vector<char> buf(offsetof(ClipState, trk));
  VBT_HIP_CHECK(hipMemcpy(buf.data(), &t->states[clip], buf.size(), hipMemcpyDeviceToHost));
  const ClipState* st = (const ClipState*)buf.data();
  int n = std::min(st->last_n, cap);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < 7; j++) out7[i * 7 + j] = st->last_out[i][j];
    vel2[i * 2] = st->last_out[i][7];
    vel2[i * 2 + 1] = st->last_out[i][8];
  }
  *M = n;
  return VBT_OK;
}

int vbt_tracker_status(vbt_tracker* t, int clip, int32_t* n_rows, int32_t* n_trackers, int32_t* overflow, int32_t* rows_overflow,
                       int32_t* frame_count) {
  if (!t || clip < 0 || clip >= t->n_clips) { set_error("bad argument"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  std::vector<char> buf(offsetof(ClipState, last_out));
  ClipState* st = (ClipState*)buf.data();
  int rc = fetch_state_header(t, clip, st);
  if (rc) return rc;
  if (n_rows) *n_rows = st->nrows;
  if (n_trackers) *n_trackers = st->ntrk;
  if (overflow) *overflow = st->overflow;
  if (rows_overflow) *rows_overflow = st->rows_overflow;
  if (frame_count) *frame_count = st->frame_count;
  return VBT_OK;
}

int vbt_tracker_get_trackers(vbt_tracker* t, int clip, int32_t* ids, double* kfx, int cap, int* n) {
  if (!t || !ids || !kfx || !n || clip < 0 || clip >= t->n_clips) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (t->view_clip == clip) {
    const int m = std::min((int)t->h_view[1], cap);
    for (int i = 0; i < m; i++) {
      const double* o = t->h_view + 2 + MAXD * 9 + i * 8;
      ids[i] = (int32_t)o[0];
      for (int j = 0; j < 7; j++) kfx[i * 7 + j] = o[1 + j];
    }
    *n = m;
    return VBT_OK;
  }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  std::vector<char> buf(sizeof(ClipState));
  VBT_HIP_CHECK(hipMemcpy(buf.data(), &t->states[clip], sizeof(ClipState), hipMemcpyDeviceToHost));
  const ClipState* st = (const ClipState*)buf.data();
  int m = std::min(st->ntrk, cap);
  for (int i = 0; i < m; i++) {
    const Trk& k = st->trk[st->order[i]];
    ids[i] = k.id;
    for (int j = 0; j < 7; j++) kfx[i * 7 + j] = k.x[j];
  }
  *n = m;
  return VBT_OK;
}

int vbt_tracker_rows(vbt_tracker* t, int clip, int64_t* id, double* cols7, int cap, int* n) {
  if (!t || !id || !cols7 || !n || clip < 0 || clip >= t->n_clips) { set_error("bad argument"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  std::vector<char> hb(offsetof(ClipState, last_out));
  ClipState* st = (ClipState*)hb.data();
  int rc = fetch_state_header(t, clip, st);
  if (rc) return rc;
  if (st->overflow > 0) { set_error("clip %d: more than %d live tracks (%d births dropped)", clip, MAXT, st->overflow); return VBT_ERR_CAPACITY; }
  if (st->rows_overflow > 0) { set_error("clip %d: row capacity %d exceeded by %d", clip, t->rows_cap, st->rows_overflow); return VBT_ERR_CAPACITY; }
  int m = st->nrows;
  if (m > cap) { set_error("clip %d has %d rows, buffer holds %d", clip, m, cap); return VBT_ERR_CAPACITY; }
  std::vector<Row> r(m);
  if (m) VBT_HIP_CHECK(hipMemcpy(r.data(), t->rows + (size_t)clip * t->rows_cap, sizeof(Row) * m, hipMemcpyDeviceToHost));
  for (int i = 0; i < m; i++) {
    id[i] = r[i].id;
    double* o = cols7 + (size_t)i * 7;
    o[0] = r[i].time; o[1] = r[i].x; o[2] = r[i].y; o[3] = r[i].dx; o[4] = r[i].dy; o[5] = r[i].h; o[6] = r[i].w;
  }
  *n = m;
  return VBT_OK;
}

int vbt_tracker_finish(vbt_tracker* t, double plate_diameter, double diff_threshold, double min_distance, void* stream) {
  RoctxRange range("vbt:finish");
  if (!t) { set_error("NULL tracker"); return VBT_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  select_gather_kernel<<<t->n_clips, 64, 0, st>>>(t->states, t->rows, t->rows_cap, t->cols, t->T, t->best);
  VtParams vp{plate_diameter, diff_threshold, min_distance, 1, 1};
  analyze_kernel<<<t->n_clips, 64, 0, st>>>(t->cols, t->T, t->rows_cap, vp, t->scratch, t->phases, t->nph);
  VBT_HIP_CHECK(hipGetLastError());
  t->finished = true;
  t->finish_stream = st;
  return VBT_OK;
}

int vbt_tracker_phases(vbt_tracker* t, int clip, int32_t* best_id, double* phases6, int cap, int* P) {
  if (!t || !best_id || !phases6 || !P || clip < 0 || clip >= t->n_clips) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (!t->finished) { set_error("vbt_tracker_phases before vbt_tracker_finish"); return VBT_ERR_STATE; }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  int n = 0;
  VBT_HIP_CHECK(hipMemcpy(&n, t->nph + clip, sizeof(int), hipMemcpyDeviceToHost));
  VBT_HIP_CHECK(hipMemcpy(best_id, t->best + clip, sizeof(int), hipMemcpyDeviceToHost));
  if (n > cap) { set_error("clip %d has %d phases, buffer holds %d", clip, n, cap); return VBT_ERR_CAPACITY; }
  if (n) VBT_HIP_CHECK(hipMemcpy(phases6, t->phases + (size_t)clip * MAXPH * 6, sizeof(double) * 6 * n, hipMemcpyDeviceToHost));
  *P = n;
  return VBT_OK;
}

int vbt_analyze(const double* cols7, int T, int preprocess, int flush, double plate_diameter, double diff_threshold,
                double min_distance, double* phases6, int cap, int* P, int device) {
  if (!cols7 && T > 0) { set_error("vbt_analyze: NULL rows"); return VBT_ERR_ARG; }
  if (!phases6 || !P || T < 0) { set_error("vbt_analyze: bad argument"); return VBT_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error("vbt_analyze: HIP device %d not available (%d visible) - no CPU fallback", device, ndev);
    return VBT_ERR_HIP;
  }
  VBT_HIP_CHECK(hipSetDevice(device));
  *P = 0;
  if (T == 0) return VBT_OK;
  double *dc = nullptr, *ds = nullptr, *dp = nullptr;
  int *dn = nullptr, *dT = nullptr;
  VBT_HIP_CHECK(hipMalloc((void**)&dc, sizeof(double) * 7 * T));
  VBT_HIP_CHECK(hipMalloc((void**)&ds, sizeof(double) * 5 * T));
  VBT_HIP_CHECK(hipMalloc((void**)&dp, sizeof(double) * 6 * MAXPH));
  VBT_HIP_CHECK(hipMalloc((void**)&dn, sizeof(int)));
  VBT_HIP_CHECK(hipMalloc((void**)&dT, sizeof(int)));
  VBT_HIP_CHECK(hipMemcpy(dc, cols7, sizeof(double) * 7 * T, hipMemcpyHostToDevice));
  VBT_HIP_CHECK(hipMemcpy(dT, &T, sizeof(int), hipMemcpyHostToDevice));
  VtParams vp{plate_diameter, diff_threshold, min_distance, preprocess, flush};
  analyze_kernel<<<1, 64>>>(dc, dT, T, vp, ds, dp, dn);
  int n = 0;
  hipError_t e = hipMemcpy(&n, dn, sizeof(int), hipMemcpyDeviceToHost);
  int rc = VBT_OK;
  if (e != hipSuccess) { set_error("analyze kernel failed: %s", hipGetErrorString(e)); rc = VBT_ERR_HIP; }
  else if (n > cap) { set_error("%d phases, buffer holds %d", n, cap); rc = VBT_ERR_CAPACITY; }
  else {
    if (n) (void)hipMemcpy(phases6, dp, sizeof(double) * 6 * n, hipMemcpyDeviceToHost);
    *P = n;
  }
  (void)hipFree(dc); (void)hipFree(ds); (void)hipFree(dp); (void)hipFree(dn); (void)hipFree(dT);
  return rc;
}

int vbt_window_means(const double* rows, int T, int ncols, const int32_t* windows, double* out, int device) {
  if (T < 0 || ncols < 1 || ncols > 64 || !windows || (T > 0 && (!rows || !out))) { set_error("vbt_window_means: bad argument"); return VBT_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error("vbt_window_means: HIP device %d not available (%d visible) - no CPU fallback", device, ndev);
    return VBT_ERR_HIP;
  }
  if (T == 0) return VBT_OK;
  VBT_HIP_CHECK(hipSetDevice(device));
  double *din = nullptr, *dout = nullptr;
  int* dw = nullptr;
  const size_t bytes = sizeof(double) * (size_t)T * ncols;
  VBT_HIP_CHECK(hipMalloc((void**)&din, bytes));
  VBT_HIP_CHECK(hipMalloc((void**)&dout, bytes));
  VBT_HIP_CHECK(hipMalloc((void**)&dw, sizeof(int) * ncols));
  VBT_HIP_CHECK(hipMemcpy(din, rows, bytes, hipMemcpyHostToDevice));
  VBT_HIP_CHECK(hipMemcpy(dw, windows, sizeof(int) * ncols, hipMemcpyHostToDevice));
  window_means_kernel<<<1, 64>>>(din, T, ncols, dw, dout);
  hipError_t e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dw);
  if (e != hipSuccess) { set_error("window means kernel failed: %s", hipGetErrorString(e)); return VBT_ERR_HIP; }
  return VBT_OK;
}

// Clip close, host side: one pack kernel, ONE asynchronous copy into pinned memory, ONE stream synchronisation (on the
// stream vbt_tracker_finish ran on - not a device-wide one).
int vbt_tracker_summary(vbt_tracker* t, int32_t* best_ids, int32_t* n_rows, int32_t* n_phases, int32_t* overflow, double* phases6, int cap) {
  if (!t || !best_ids || !n_rows || !n_phases || !overflow || !phases6 || cap < 1) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (!t->finished) { set_error("vbt_tracker_summary before vbt_tracker_finish"); return VBT_ERR_STATE; }
  const int n = t->n_clips;
  const int host_cap = cap;     // stride of the caller's phases6 buffer: [n_clips][host_cap][6]
  cap = std::min(cap, MAXPH);   // phases packed per clip (a clip never holds more than MAXPH)
  const size_t rec = 16 + (size_t)cap * 48, bytes = rec * n;
  if (bytes > t->summary_bytes) {
    if (t->d_summary) (void)hipFree(t->d_summary);
    if (t->h_summary) (void)hipHostFree(t->h_summary);
    t->d_summary = nullptr; t->h_summary = nullptr; t->summary_bytes = 0;
    VBT_HIP_CHECK(hipMalloc((void**)&t->d_summary, bytes));
    VBT_HIP_CHECK(hipHostMalloc((void**)&t->h_summary, bytes, hipHostMallocDefault));
    t->summary_bytes = bytes;
  }
  hipStream_t st = t->finish_stream;
  pack_summary_kernel<<<n, 64, 0, st>>>(t->states, t->best, t->nph, t->phases, cap, t->d_summary);
  VBT_HIP_CHECK(hipMemcpyAsync(t->h_summary, t->d_summary, bytes, hipMemcpyDeviceToHost, st));
  VBT_HIP_CHECK(hipStreamSynchronize(st));
  for (int c = 0; c < n; c++) {
    const unsigned char* r = t->h_summary + c * rec;
    const int* h = (const int*)r;
    best_ids[c] = h[0]; n_rows[c] = h[1]; n_phases[c] = h[2]; overflow[c] = h[3];
    if (h[2] > cap) { set_error("clip %d has %d phases, buffer holds %d", c, h[2], cap); return VBT_ERR_CAPACITY; }
    memcpy(phases6 + (size_t)c * host_cap * 6, r + 16, (size_t)h[2] * 48);
  }
  return VBT_OK;
}

// DataFrame rows of EVERY clip (all ids, emission order) in one strided copy: rows_host = [n_clips][cap] records of
// 64 bytes {int64 id; double time, x, y, dx, dy, norm_plate_height, norm_plate_width} (reference track.py:227-234).
// counts[c] = rows of clip c.  rows_host may be pinned (then the copy is one DMA) or pageable.
int vbt_tracker_rows_all(vbt_tracker* t, int32_t* counts, void* rows_host, int cap, void* stream) {
  if (!t || !counts || !rows_host || cap < 1) { set_error("vbt_tracker_rows_all: bad argument"); return VBT_ERR_ARG; }
  static_assert(sizeof(Row) == 64, "row record");
  hipStream_t st = (hipStream_t)stream;
  const int n = t->n_clips;
  const size_t hdr = offsetof(ClipState, last_out);
  std::vector<char> heads((size_t)n * hdr);
  VBT_HIP_CHECK(hipMemcpy2DAsync(heads.data(), hdr, t->states, sizeof(ClipState), hdr, n, hipMemcpyDeviceToHost, st));
  VBT_HIP_CHECK(hipStreamSynchronize(st));
  int most = 0;
  for (int c = 0; c < n; c++) {
    const ClipState* cs = (const ClipState*)(heads.data() + (size_t)c * hdr);
    if (cs->overflow > 0) { set_error("clip %d: more than %d live tracks (%d births dropped)", c, MAXT, cs->overflow); return VBT_ERR_CAPACITY; }
    if (cs->rows_overflow > 0) { set_error("clip %d: row capacity %d exceeded by %d", c, t->rows_cap, cs->rows_overflow); return VBT_ERR_CAPACITY; }
    if (cs->nrows > cap) { set_error("clip %d has %d rows, buffer holds %d", c, cs->nrows, cap); return VBT_ERR_CAPACITY; }
    counts[c] = cs->nrows;
    most = std::max(most, cs->nrows);
  }
  if (most > 0) {
    VBT_HIP_CHECK(hipMemcpy2DAsync(rows_host, sizeof(Row) * (size_t)cap, t->rows, sizeof(Row) * (size_t)t->rows_cap, sizeof(Row) * (size_t)most, n,
                                   hipMemcpyDeviceToHost, st));
    VBT_HIP_CHECK(hipStreamSynchronize(st));
  }
  return VBT_OK;
}

}  // extern "C"
