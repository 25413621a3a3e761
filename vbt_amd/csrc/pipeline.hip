// vbt_pipeline: the clip loop of reference track.py:129-260 as one object behind the C ABI (include/vbt_hip.h, "pipeline").
//
// Host code only - every kernel it enqueues is reached through the library's own entry points (vbt_detect_async,
// vbt_tracker_update_from_*, vbt_resize_frames, vbt_gather_frames).  What lives here is the part of the fast path that is not a
// kernel: which stream a forward runs on and that the busy streams sit on distinct hardware queues, the ring of output slots and the
// events that order detector(t) -> tracker(t) -> slot reuse, the staging ring of the host-fed mode, the deferred tracker groups of
// the small-batch path, clip close.  Plain hipMalloc / hipHostMalloc / hipStream / hipEvent: no framework allocator, no
// framework streams.
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>

#include "common.h"

// (resize_frames_dev, common.h: preprocess_image reading either whole source frames [B][H][W][3] or, compact != 0, only the row pairs the
//  bilinear resize touches: [B][2h][W][3], pair d = source rows p(d), p(d) + 1 with p(d) = min(floor(src_y(d)), H - 2))

using namespace vbt;

namespace {

// ------------------------------------------------------------------------------------------------------------------------------
// Process-wide stream pool, one per device.  HIP binds a stream to one of GPU_MAX_HW_QUEUES hardware queues when it is created (a
// zig-zag that also counts streams created by others) and the queue cannot be queried, so a pool stream is CLASSIFIED once per
// process: timed with a spinning wave against one representative of every queue group known so far (vbt_streams_share_queue,
// ~0.3 ms per probe).  Streams are kept for the life of the process and handed out again when a pipeline goes away, so that any
// number of pipelines created one after the other end up on the same few streams.
// ------------------------------------------------------------------------------------------------------------------------------
struct StreamPool {
  std::vector<hipStream_t> streams;
  std::vector<int> free;          // indices not owned by a pipeline
  std::map<int, int> group;       // stream index -> hardware-queue group
  std::vector<int> reps;          // one stream index per known group
};
std::mutex g_pool_mu;
StreamPool g_pools[64];

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

}  // namespace

struct vbt_pipeline {
  vbt_pipeline_params prm{};
  int n = 0, n_trk = 0, depth = 0, ring = 0, defer = 0, device = 0, size = 0;
  bool trk_inline = false, placement_ok = true;
  std::vector<double> fps;
  std::vector<vbt_model*> models;
  vbt_tracker* trk = nullptr;
  // detector outputs: one block per tensor, [ring slot][batch slot]...: the deferred walk addresses frame f of clip c as slot (o0 + f) * n + c
  float *boxes = nullptr, *scores = nullptr, *classes = nullptr;
  int32_t* counts = nullptr;
  enum { PLAIN = 0, SLOTS = 1, RUNS = 2 };
  struct SlotMeta {
    int kind = PLAIN, B = 0, fc = 0;
    std::vector<double> times;
    std::vector<int32_t> cmap;
    std::vector<vbt_run> runs;
  };
  std::vector<SlotMeta> meta;
  std::vector<int> group, pending, own_streams;
  hipStream_t det_streams[8] = {nullptr}, copy_stream = nullptr, trk_stream = nullptr;
  std::vector<hipEvent_t> ev_in, ev_det, ev_trk;
  std::vector<int> trk_ev_of;      // ring slot -> index into ev_trk of the tracker launch that read it last (-1: none)
  int last_trk = -1;               // index into ev_trk of the most recent tracker launch
  struct Stage {
    uint8_t* buf = nullptr;
    size_t bytes = 0;
    hipEvent_t free_ev = nullptr, copy_ev = nullptr;
    bool free_set = false;
  };
  std::vector<Stage> stage;
  int stage_idx = 0;
  std::vector<uint8_t*> resized;
  std::vector<int64_t> clip_frames;
  std::vector<int> row_table;      // p(d) of the compact upload, for (row_H, row_h)
  int row_H = 0, row_h = 0;
  int frame_count = 0, step_idx = 0, last_B = 0;
  uint64_t h2d_bytes = 0, step_host_ns = 0, step_calls = 0;
};

namespace {

// host time of a step call, for vbt_pipeline_info (is the host or the GPU pacing a small-batch run?)
struct StepTimer {
  vbt_pipeline* p;
  std::chrono::steady_clock::time_point t0;
  explicit StepTimer(vbt_pipeline* p_) : p(p_), t0(std::chrono::steady_clock::now()) {}
  ~StepTimer() {
    p->step_host_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    p->step_calls++;
  }
};

#define PL_CHECK(expr)            \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != VBT_OK) return rc_; \
  } while (0)

// ---- stream pool ----
int pool_take(vbt_pipeline* p, StreamPool& pool, int i, bool create, int* out) {
  if (i < 0) {
    if (!pool.free.empty() && !create) {
      i = *std::min_element(pool.free.begin(), pool.free.end());
    } else {
      void* h = nullptr;
      PL_CHECK(vbt_stream_create(p->device, &h));
      pool.streams.push_back((hipStream_t)h);
      i = (int)pool.streams.size() - 1;
      pool.free.push_back(i);
    }
  }
  pool.free.erase(std::find(pool.free.begin(), pool.free.end(), i));
  p->own_streams.push_back(i);
  *out = i;
  return VBT_OK;
}

int streams_shared(StreamPool& pool, int i, int j, bool* shared) {
  // host-timed: a descheduled host thread can make one probe read "shared"; two in a row cannot
  for (int rep = 0; rep < 2; rep++) {
    int sh = 0;
    PL_CHECK(vbt_streams_share_queue((void*)pool.streams[i], (void*)pool.streams[j], 150, &sh));
    if (!sh) { *shared = false; return VBT_OK; }
  }
  *shared = true;
  return VBT_OK;
}

int group_of(StreamPool& pool, int i, int* g_out) {
  auto it = pool.group.find(i);
  if (it == pool.group.end()) {
    int g = -1;
    for (int k = 0; k < (int)pool.reps.size() && g < 0; k++) {
      bool sh = false;
      PL_CHECK(streams_shared(pool, i, pool.reps[k], &sh));
      if (sh) g = k;
    }
    if (g < 0) {
      g = (int)pool.reps.size();
      pool.reps.push_back(i);
    }
    it = pool.group.emplace(i, g).first;
  }
  *g_out = it->second;
  return VBT_OK;
}

// The streams that carry kernels side by side (detector slots, the copy stream, the tracker stream unless its step runs inline)
// must sit on distinct hardware queues: a pipeline takes its busy streams from distinct groups - a stream that once collided is
// simply left for another role - and only creates streams while some group is still unseen.
int place_streams(vbt_pipeline* p, StreamPool& pool, int role_idx[10]) {
  // roles: 0..depth-1 detector slots, 8 copy, 9 tracker
  if (env_int("VBT_PLACE_STREAMS", 1) == 0) return VBT_OK;
  VBT_HIP_CHECK(hipDeviceSynchronize());
  std::vector<int> busy;
  for (int k = 0; k < p->depth; k++) busy.push_back(k);
  if (p->depth < 4) busy.push_back(8);   // (four hardware queues: with four forwards in flight the copy stream has to share one, which costs a small batch nothing)
  if (!p->trk_inline) busy.push_back(9);
  auto is_busy = [&](int r) { return std::find(busy.begin(), busy.end(), r) != busy.end(); };
  const int all_roles[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
  const int nq = std::max(1, env_int("GPU_MAX_HW_QUEUES", 4));
  std::vector<int> used;
  auto in_used = [&](int g) { return std::find(used.begin(), used.end(), g) != used.end(); };
  bool failed = false;
  for (int role : busy) {
    int cur = role_idx[role], g = 0;
    PL_CHECK(group_of(pool, cur, &g));
    if (in_used(g)) {
      // another stream of a group this pipeline does not use yet: one it already holds for an idle role, a free pool stream, or -
      // while fewer groups than hardware queues are known, and at most 3 nq times - a new one
      int cand = -1;
      bool from_spare = false;
      for (int r : all_roles) {
        if (role_idx[r] < 0 || is_busy(r)) continue;
        int gr = 0;
        PL_CHECK(group_of(pool, role_idx[r], &gr));
        if (!in_used(gr)) { cand = role_idx[r]; from_spare = true; break; }
      }
      if (cand < 0) {
        std::vector<int> fr = pool.free;
        std::sort(fr.begin(), fr.end());
        for (int i : fr) {
          int gi = 0;
          PL_CHECK(group_of(pool, i, &gi));
          if (!in_used(gi)) { cand = i; break; }
        }
      }
      int created = 0;
      while (cand < 0 && (int)pool.reps.size() < nq && created < 3 * nq) {
        int i = -1, gi = 0;
        PL_CHECK(pool_take(p, pool, -1, true, &i));
        created++;
        PL_CHECK(group_of(pool, i, &gi));
        if (!in_used(gi)) {
          cand = i;
        } else {   // stays in the pool for a later pipeline / another role
          pool.free.push_back(i);
          p->own_streams.erase(std::find(p->own_streams.begin(), p->own_streams.end(), i));
        }
      }
      if (cand < 0) { failed = true; continue; }
      if (std::find(pool.free.begin(), pool.free.end(), cand) != pool.free.end()) {
        int dummy = 0;
        PL_CHECK(pool_take(p, pool, cand, false, &dummy));
      }
      if (from_spare)   // swap the two roles' streams
        for (int r : all_roles)
          if (role_idx[r] == cand) { role_idx[r] = cur; break; }
      role_idx[role] = cand;
      cur = cand;
      PL_CHECK(group_of(pool, cur, &g));
    }
    used.push_back(g);
  }
  // streams taken but left without a role go back to the pool
  for (size_t k = 0; k < p->own_streams.size();) {
    const int i = p->own_streams[k];
    bool held = false;
    for (int r : all_roles) held = held || role_idx[r] == i;
    if (held) { k++; continue; }
    p->own_streams.erase(p->own_streams.begin() + (long)k);
    pool.free.push_back(i);
  }
  if (failed) {
    p->placement_ok = false;
    char msg[512];
    snprintf(msg, sizeof(msg),
             "vbt_pipeline: could not give every pipeline stream its own hardware queue: %d busy streams (depth %d%s%s), %d distinct queues seen, "
             "GPU_MAX_HW_QUEUES=%d (too few queues for this configuration, or kernels are being serialised by a profiler); throughput will be lower",
             (int)busy.size(), p->depth, p->trk_inline ? "" : " + tracker stream", p->depth < 4 ? " + copy stream" : "", (int)pool.reps.size(), nq);
    const int strict = p->prm.strict_placement >= 0 ? p->prm.strict_placement : env_int("VBT_STRICT_PLACEMENT", 0);
    if (strict == 1) { set_error("%s", msg); return VBT_ERR_STATE; }
    fprintf(stderr, "%s\n", msg);
  }
  return VBT_OK;
}

// ---- output ring ----
inline float* boxes_of(vbt_pipeline* p, int o) { return p->boxes + (size_t)o * p->n * VBT_MAX_DETECTIONS * 4; }
inline float* scores_of(vbt_pipeline* p, int o) { return p->scores + (size_t)o * p->n * VBT_MAX_DETECTIONS; }
inline float* classes_of(vbt_pipeline* p, int o) { return p->classes + (size_t)o * p->n * VBT_MAX_DETECTIONS; }
inline int32_t* counts_of(vbt_pipeline* p, int o) { return p->counts + (size_t)o * p->n; }

int record_trk(vbt_pipeline* p, int o, hipStream_t T) {
  VBT_HIP_CHECK(hipEventRecord(p->ev_trk[o], T));
  p->trk_ev_of[o] = o;
  p->last_trk = o;
  return VBT_OK;
}

// the OC-SORT step(s) of ring slot o: on the tracker stream after the slot's detections ("own"), or at the end of the slot's own
// stream after the previous frame's tracker step ("inline": stream order gives "after this slot's detections")
int enqueue_tracker(vbt_pipeline* p, int o) {
  hipStream_t T;
  if (p->trk_inline) {
    T = p->det_streams[o % p->depth];
    if (p->last_trk >= 0) VBT_HIP_CHECK(hipStreamWaitEvent(T, p->ev_trk[p->last_trk], 0));   // tracker steps run in frame order
  } else {
    T = p->trk_stream;
    VBT_HIP_CHECK(hipStreamWaitEvent(T, p->ev_det[o], 0));
  }
  vbt_pipeline::SlotMeta& m = p->meta[o];
  // frame times / clip map / runs of the step travel in the kernel arguments (read during the call, no copy in flight)
  if (m.kind == vbt_pipeline::RUNS) {
    PL_CHECK(vbt_tracker_update_from_detections_seq(p->trk, boxes_of(p, o), scores_of(p, o), counts_of(p, o), m.B, m.runs.data(), (int)m.runs.size(),
                                                    p->prm.detection_threshold, (void*)T));
  } else if (m.kind == vbt_pipeline::SLOTS) {
    PL_CHECK(vbt_tracker_update_from_slots(p->trk, boxes_of(p, o), scores_of(p, o), counts_of(p, o), m.cmap.data(), m.times.data(), p->n,
                                           p->prm.detection_threshold, (void*)T));
  } else {
    PL_CHECK(vbt_tracker_update_from_detections(p->trk, boxes_of(p, o), scores_of(p, o), counts_of(p, o), m.times.data(), p->prm.detection_threshold,
                                                (void*)T));
  }
  return record_trk(p, o, T);
}

// Hand the deferred plain steps to the tracker: ONE launch of the time-batched walk on the stream of the group's last forward,
// after the other members' forwards (events) and the previous tracker launch.
int flush_group(vbt_pipeline* p) {
  std::vector<int> g;
  g.swap(p->group);
  if (g.empty()) return VBT_OK;
  for (int o : g) p->pending.erase(std::find(p->pending.begin(), p->pending.end(), o));
  if (g.size() == 1) return enqueue_tracker(p, g[0]);
  const int n = p->n, last = g.back();
  hipStream_t T = p->det_streams[last % p->depth];
  for (size_t i = 0; i + 1 < g.size(); i++) VBT_HIP_CHECK(hipStreamWaitEvent(T, p->ev_det[g[i]], 0));
  if (p->last_trk >= 0) VBT_HIP_CHECK(hipStreamWaitEvent(T, p->ev_trk[p->last_trk], 0));
  const int fstep = p->meta[g[1]].fc - p->meta[g[0]].fc;
  std::vector<vbt_run> ra((size_t)n);
  for (int c = 0; c < n; c++) ra[c] = vbt_run{c, c, n, (int)g.size(), p->meta[g[0]].fc, fstep, p->fps[c]};
  // slot (o - g[0]) * n + c of the block that starts at ring slot g[0]
  PL_CHECK(vbt_tracker_update_from_detections_seq(p->trk, boxes_of(p, g[0]), scores_of(p, g[0]), counts_of(p, g[0]), (int)g.size() * n, ra.data(), n,
                                                  p->prm.detection_threshold, (void*)T));
  VBT_HIP_CHECK(hipEventRecord(p->ev_trk[last], T));
  for (int o : g) p->trk_ev_of[o] = last;
  p->last_trk = last;
  return VBT_OK;
}

int drain(vbt_pipeline* p) {
  PL_CHECK(flush_group(p));
  while (!p->pending.empty()) {
    const int o = p->pending.front();
    p->pending.erase(p->pending.begin());
    PL_CHECK(enqueue_tracker(p, o));
  }
  if (p->trk_inline && p->last_trk >= 0)   // clip close / row reads run on the tracker stream
    VBT_HIP_CHECK(hipStreamWaitEvent(p->trk_stream, p->ev_trk[p->last_trk], 0));
  return VBT_OK;
}

// ---- staging ring of the host-fed / gathered input ----
// Before an H2D copy into staging buffer j is enqueued the forward that last read the buffer must be done.  The wait is on the
// HOST (the event is depth + 2 steps old: it has completed unless the caller is that many steps ahead of the GPU, and then blocking
// the caller is the back-pressure wanted), NOT a stream wait on the copy stream: a cross-stream event wait in front of a DMA copy
// makes hipMemcpyAsync itself block the calling thread on this stack (profiles/r04_h2d_pinned_order.md).
int stage_take(vbt_pipeline* p, size_t bytes, bool host_gate, hipStream_t S, int* j_out) {
  const int j = p->stage_idx % (int)p->stage.size();
  p->stage_idx++;
  vbt_pipeline::Stage& st = p->stage[j];
  if (st.bytes < bytes) {
    // a buffer that has to be replaced may still be read by a forward or written by a copy in flight (up to depth + 2 steps)
    if (st.buf) {
      if (st.free_set) VBT_HIP_CHECK(hipEventSynchronize(st.free_ev));
      VBT_HIP_CHECK(hipEventSynchronize(st.copy_ev));
      VBT_HIP_CHECK(hipFree(st.buf));
      st.buf = nullptr;
      st.bytes = 0;
      st.free_set = false;
    }
    VBT_HIP_CHECK(hipMalloc((void**)&st.buf, bytes + 64));
    st.bytes = bytes;
  }
  if (st.free_set) {
    if (host_gate) VBT_HIP_CHECK(hipEventSynchronize(st.free_ev));
    else VBT_HIP_CHECK(hipStreamWaitEvent(S, st.free_ev, 0));
  }
  *j_out = j;
  return VBT_OK;
}

// p(d): first source row of the pair output row d reads, in the resize kernel's own float32 arithmetic
const std::vector<int>& row_table(vbt_pipeline* p, int H, int h) {
  if (p->row_H != H || p->row_h != h) {
    p->row_table.resize((size_t)h);
    const float sy = (float)H / (float)h;
    for (int d = 0; d < h; d++) {
      const float iy = ((float)d + 0.5f) * sy - 0.5f;
      const int y0 = std::max((int)floorf(iy), 0);
      p->row_table[d] = std::min(y0, H - 2);
    }
    p->row_H = H;
    p->row_h = h;
  }
  return p->row_table;
}

inline bool compact_rows(const vbt_pipeline* p, int H, int W) {
  // the resize reads two source rows per output row (tf.image.resize bilinear without antialias, odt.py:15-16): when the source holds
  // more than twice as many rows, only those are uploaded
  static const bool off = getenv("VBT_NO_ROW_UPLOAD") != nullptr;
  return !off && H > 0 && W > 0 && H >= 2 && 2 * p->size < H;
}

// nf host frames [nf][H][W][3] -> staging buffer (frame slot0 onwards) on the copy stream; compact: only the row pairs
int h2d_frames(vbt_pipeline* p, uint8_t* stage_buf, int slot0, const uint8_t* host, int nf, int H, int W, bool compact) {
  hipStream_t C = p->copy_stream;
  const size_t row = (size_t)W * 3;
  if (!compact) {
    const size_t fb = (size_t)H * row;
    VBT_HIP_CHECK(hipMemcpyAsync(stage_buf + (size_t)slot0 * fb, host, (size_t)nf * fb, hipMemcpyHostToDevice, C));
    p->h2d_bytes += (uint64_t)nf * fb;
    return VBT_OK;
  }
  const int h = p->size;
  const std::vector<int>& tab = row_table(p, H, h);
  uint8_t* dst = stage_buf + (size_t)slot0 * 2 * h * row;
  const int step = h > 1 ? tab[1] - tab[0] : 0;
  bool uniform = h > 1;
  for (int d = 1; d < h && uniform; d++) uniform = tab[d] - tab[d - 1] == step;
  if (uniform && (long)step * h == H) {
    // integer scale: the pairs sit at one pitch across the whole batch - ONE strided copy
    VBT_HIP_CHECK(hipMemcpy2DAsync(dst, 2 * row, host + (size_t)tab[0] * row, (size_t)step * row, 2 * row, (size_t)nf * h, hipMemcpyHostToDevice, C));
  } else if (uniform) {
    for (int b = 0; b < nf; b++)
      VBT_HIP_CHECK(hipMemcpy2DAsync(dst + (size_t)b * 2 * h * row, 2 * row, host + ((size_t)b * H + tab[0]) * row, (size_t)step * row, 2 * row, (size_t)h,
                                     hipMemcpyHostToDevice, C));
  } else {
    // a row table: pair d of every frame in one strided copy (pitch = one source frame / one compact frame)
    for (int d = 0; d < h; d++)
      VBT_HIP_CHECK(hipMemcpy2DAsync(dst + (size_t)d * 2 * row, (size_t)2 * h * row, host + (size_t)tab[d] * row, (size_t)H * row, 2 * row, (size_t)nf,
                                     hipMemcpyHostToDevice, C));
  }
  p->h2d_bytes += (uint64_t)nf * 2 * h * row;
  return VBT_OK;
}

int ensure_resized(vbt_pipeline* p, int k) {
  if (!p->resized[k]) VBT_HIP_CHECK(hipMalloc((void**)&p->resized[k], (size_t)p->n * p->size * p->size * 3 + 64));
  return VBT_OK;
}

struct Sources {
  const uint8_t* frames = nullptr;              // assembled batch, or
  const uint8_t* const* run_sources = nullptr;  // one source per run
  bool on_device = false;
  int src_h = 0, src_w = 0, swap_rb = 0;
};

// Brings B frames to the network resolution on stream S of forward slot k and returns the device pointer the detector reads.
int prepare_frames(vbt_pipeline* p, int k, hipStream_t S, const Sources& src, const vbt_run* runs, int n_runs, int B, void* caller_stream,
                   const uint8_t** frames_dev, int* stage_j) {
  const int size = p->size;
  const bool resize = src.src_h > 0 && src.src_w > 0 && (src.src_h != size || src.src_w != size || src.swap_rb);
  const int H = src.src_h > 0 ? src.src_h : size, W = src.src_w > 0 ? src.src_w : size;
  const size_t fb = (size_t)H * W * 3;
  *stage_j = -1;
  const uint8_t* ptr = nullptr;
  bool compact = false;
  if (src.on_device) {
    // frames are ready once the caller's stream gets here
    const int ki = k;
    VBT_HIP_CHECK(hipEventRecord(p->ev_in[ki], (hipStream_t)caller_stream));
    VBT_HIP_CHECK(hipStreamWaitEvent(S, p->ev_in[ki], 0));
  }
  if (src.frames) {
    if (src.on_device) {
      ptr = src.frames;
    } else {
      compact = resize && compact_rows(p, H, W);
      int j = 0;
      PL_CHECK(stage_take(p, compact ? (size_t)p->n * 2 * size * W * 3 : (size_t)p->n * fb, true, S, &j));
      PL_CHECK(h2d_frames(p, p->stage[j].buf, 0, src.frames, B, H, W, compact));
      VBT_HIP_CHECK(hipEventRecord(p->stage[j].copy_ev, p->copy_stream));
      VBT_HIP_CHECK(hipStreamWaitEvent(S, p->stage[j].copy_ev, 0));
      ptr = p->stage[j].buf;
      *stage_j = j;
    }
  } else {
    // one source per run: the batch is assembled in a staging buffer
    compact = !src.on_device && resize && compact_rows(p, H, W);
    int j = 0;
    PL_CHECK(stage_take(p, compact ? (size_t)p->n * 2 * size * W * 3 : (size_t)p->n * fb, !src.on_device, S, &j));
    uint8_t* st = p->stage[j].buf;
    if (!src.on_device) {
      for (int i = 0; i < n_runs; i++) PL_CHECK(h2d_frames(p, st, runs[i].slot0, src.run_sources[i], runs[i].n_frames, H, W, compact));
      VBT_HIP_CHECK(hipEventRecord(p->stage[j].copy_ev, p->copy_stream));
      VBT_HIP_CHECK(hipStreamWaitEvent(S, p->stage[j].copy_ev, 0));
    } else {
      bool aligned = fb % 16 == 0;
      for (int i = 0; i < n_runs && aligned; i++) aligned = ((uintptr_t)src.run_sources[i] & 15) == 0;
      if (aligned) {
        std::vector<const uint8_t*> ptrs((size_t)B, nullptr);
        for (int i = 0; i < n_runs; i++)
          for (int f = 0; f < runs[i].n_frames; f++) ptrs[(size_t)runs[i].slot0 + f] = src.run_sources[i] + (size_t)f * fb;
        PL_CHECK(vbt_gather_frames(st, ptrs.data(), B, fb, (void*)S));
      } else {
        // a frame size that is not a multiple of 16 bytes (any source resolution is allowed): the gather kernel moves 16-byte
        // pieces, so the batch is assembled by one device copy per run instead
        for (int i = 0; i < n_runs; i++)
          VBT_HIP_CHECK(hipMemcpyAsync(st + (size_t)runs[i].slot0 * fb, src.run_sources[i], (size_t)runs[i].n_frames * fb, hipMemcpyDeviceToDevice, S));
      }
    }
    ptr = st;
    *stage_j = j;
  }
  if (resize) {
    PL_CHECK(ensure_resized(p, k));
    PL_CHECK(resize_frames_dev(ptr, B, H, W, p->resized[k], size, size, src.swap_rb, compact ? 1 : 0, S));
    ptr = p->resized[k];
  }
  *frames_dev = ptr;
  return VBT_OK;
}

int after_detect(vbt_pipeline* p, int o, int stage_j, hipStream_t S) {
  VBT_HIP_CHECK(hipEventRecord(p->ev_det[o], S));
  if (stage_j >= 0) {
    VBT_HIP_CHECK(hipEventRecord(p->stage[stage_j].free_ev, S));
    p->stage[stage_j].free_set = true;
  }
  return VBT_OK;
}

int wait_slot_free(vbt_pipeline* p, int o, hipStream_t S) {
  if (std::find(p->pending.begin(), p->pending.end(), o) != p->pending.end()) {
    set_error("vbt_pipeline: ring slot %d still holds a step whose tracker update has not been enqueued", o);
    return VBT_ERR_STATE;
  }
  if (p->trk_ev_of[o] >= 0) VBT_HIP_CHECK(hipStreamWaitEvent(S, p->ev_trk[p->trk_ev_of[o]], 0));   // the tracker is done with this slot's previous outputs
  return VBT_OK;
}

// One time-batched step: `asm_runs` say where the sources sit in the batch (assembly), `walk_runs` what the tracker walks.
int step_runs_impl(vbt_pipeline* p, const Sources& src, const vbt_run* asm_runs, int n_asm, std::vector<vbt_run>& walk_runs, int B, int track,
                   float* out_boxes, float* out_scores, float* out_classes, int32_t* out_counts, void* caller_stream) {
  PL_CHECK(flush_group(p));
  const int o = p->step_idx % p->ring, k = o % p->depth;
  hipStream_t S = p->det_streams[k];
  PL_CHECK(wait_slot_free(p, o, S));
  p->step_idx++;
  const uint8_t* fd = nullptr;
  int stage_j = -1;
  PL_CHECK(prepare_frames(p, k, S, src, asm_runs, n_asm, B, caller_stream, &fd, &stage_j));
  const bool outs = out_boxes != nullptr;
  float* b = outs ? out_boxes : boxes_of(p, o);
  float* s = outs ? out_scores : scores_of(p, o);
  float* c = outs ? out_classes : classes_of(p, o);
  int32_t* cnt = outs ? out_counts : counts_of(p, o);
  PL_CHECK(vbt_detect_async(p->models[k], fd, B, (void*)S, b, s, c, cnt));
  PL_CHECK(after_detect(p, o, stage_j, S));
  vbt_pipeline::SlotMeta& m = p->meta[o];
  m.kind = vbt_pipeline::RUNS;
  m.runs.swap(walk_runs);
  m.B = B;
  p->last_B = B;
  if (!track) return VBT_OK;
  p->pending.push_back(o);
  while ((int)p->pending.size() >= (p->trk_inline ? 1 : p->depth)) {
    const int q = p->pending.front();
    p->pending.erase(p->pending.begin());
    PL_CHECK(enqueue_tracker(p, q));
  }
  return VBT_OK;
}

}  // namespace

extern "C" {

void vbt_pipeline_default_params(vbt_pipeline_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->defer = -1;
  p->selfcheck = -1;
  p->strict_placement = -1;
  p->model_flags = VBT_MODEL_DEFAULT_FLAGS;
  p->detection_threshold = 0.5f;
  p->plate_diameter = 0.45;
  p->diff_threshold = 0.6;
  p->min_distance = 0.1;
  p->tracker.max_age = 30;
  p->tracker.min_hits = 3;
  p->tracker.delta_t = 3;
  p->tracker.asso = 1;
  p->tracker.iou_threshold = 0.1;
  p->tracker.inertia = 0.2;
  p->tracker.det_thresh = 0.2;
}

void vbt_pipeline_destroy(vbt_pipeline* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  for (int k = 0; k < p->depth; k++)
    if (p->det_streams[k]) (void)hipStreamSynchronize(p->det_streams[k]);
  if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
  if (p->trk_stream) (void)hipStreamSynchronize(p->trk_stream);
  for (vbt_model* m : p->models) vbt_model_destroy(m);
  if (p->trk) vbt_tracker_destroy(p->trk);
  (void)hipFree(p->boxes);
  (void)hipFree(p->scores);
  (void)hipFree(p->classes);
  (void)hipFree(p->counts);
  for (auto& s : p->stage) {
    if (s.buf) (void)hipFree(s.buf);
    if (s.free_ev) (void)hipEventDestroy(s.free_ev);
    if (s.copy_ev) (void)hipEventDestroy(s.copy_ev);
  }
  for (uint8_t* r : p->resized)
    if (r) (void)hipFree(r);
  for (hipEvent_t e : p->ev_in) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_det) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_trk) (void)hipEventDestroy(e);
  {
    // streams are never destroyed: they go back to the process-wide pool, classified, for the next pipeline
    std::lock_guard<std::mutex> lock(g_pool_mu);
    StreamPool& pool = g_pools[p->device];
    for (int i : p->own_streams) pool.free.push_back(i);
  }
  delete p;
}

int vbt_pipeline_create(const char* container_path, const vbt_pipeline_params* prm, const double* fps_host, vbt_pipeline** out) {
  if (!container_path || !prm || !fps_host || !out) { set_error("vbt_pipeline_create: NULL argument"); return VBT_ERR_ARG; }
  *out = nullptr;
  if (prm->n_slots < 1 || prm->n_clips < 0 || prm->rows_cap < 1 || prm->depth < 0 || prm->depth > 8 || prm->device < 0 || prm->device >= 64) {
    set_error("vbt_pipeline_create: n_slots >= 1, n_clips >= 0, rows_cap >= 1, depth 0..8 required");
    return VBT_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || prm->device >= ndev) {
    set_error("vbt_pipeline_create: HIP device %d not available (%d visible) - the HIP path has no CPU fallback", prm->device, ndev);
    return VBT_ERR_HIP;
  }
  VBT_HIP_CHECK(hipSetDevice(prm->device));
  vbt_pipeline* p = new vbt_pipeline();
  p->prm = *prm;
  p->n = prm->n_slots;
  p->n_trk = prm->n_clips > 0 ? prm->n_clips : prm->n_slots;
  p->device = prm->device;
  for (int c = 0; c < p->n_trk; c++) {
    if (!(fps_host[c] > 0.0)) { delete p; set_error("vbt_pipeline_create: fps of clip %d must be > 0", c); return VBT_ERR_ARG; }
    p->fps.push_back(fps_host[c]);
  }
  // forwards in flight: 3 at batch 64 (four hardware queues: three forwards + the copy stream); a batch of one to
  // eight frames is launch latency, where a fourth forward still pays
  p->depth = prm->depth > 0 ? prm->depth : env_int("VBT_PIPELINE_DEPTH", p->n <= 8 ? 4 : 3);
  p->depth = std::max(1, std::min(p->depth, 8));
  int mode = prm->tracker_stream;
  if (mode == 0) {
    const char* ts = getenv("VBT_TRACKER_STREAM");
    if (ts && strcmp(ts, "own") != 0 && strcmp(ts, "inline") != 0) { delete p; set_error("VBT_TRACKER_STREAM must be 'own' or 'inline'"); return VBT_ERR_ARG; }
    mode = ts ? (strcmp(ts, "inline") == 0 ? 2 : 1) : (p->depth >= 3 ? 2 : 1);
  }
  p->trk_inline = mode == 2;
  // Deferred tracker steps (small batches): one single-wave tracker launch plus its cross-stream event at the end of EVERY forward
  // costs a fifth of a batch-1 step; with deferral the detections of `depth` consecutive steps stay in a ring of 2 x depth output
  // slots and ONE launch of the time-batched walk follows the group's last forward.  Only plain steps are deferred.
  const bool defer_ok = p->n_trk == p->n && p->depth >= 2 && p->trk_inline;
  int want = prm->defer >= 0 ? prm->defer : env_int("VBT_TRACKER_DEFER", (p->n <= 8 && p->n_trk == p->n && p->depth >= 2) ? 1 : 0);
  p->defer = (want == 1 && defer_ok) ? p->depth : 0;
  p->ring = p->defer ? 2 * p->depth : p->depth;
  auto fail = [&](int rc) { vbt_pipeline_destroy(p); return rc; };
  int rc = VBT_OK;
  for (int k = 0; k < p->depth; k++) {
    vbt_model* m = nullptr;
    if ((rc = vbt_model_create_ex(container_path, p->device, p->n, prm->model_flags, &m)) != VBT_OK) return fail(rc);
    p->models.push_back(m);
  }
  int shp[4];
  if ((rc = vbt_model_input_shape(p->models[0], shp)) != VBT_OK) return fail(rc);
  p->size = shp[1];
  if ((rc = vbt_tracker_create(p->n_trk, prm->rows_cap, &prm->tracker, p->device, &p->trk)) != VBT_OK) return fail(rc);
  const size_t R = (size_t)p->ring, n = (size_t)p->n, md = VBT_MAX_DETECTIONS;
  if (hipMalloc((void**)&p->boxes, R * n * md * 16) != hipSuccess || hipMalloc((void**)&p->scores, R * n * md * 4) != hipSuccess ||
      hipMalloc((void**)&p->classes, R * n * md * 4) != hipSuccess || hipMalloc((void**)&p->counts, R * n * 4) != hipSuccess) {
    set_error("vbt_pipeline_create: hipMalloc of the detector output ring failed");
    return fail(VBT_ERR_HIP);
  }
  (void)hipMemset(p->counts, 0, R * n * 4);
  p->meta.resize(R);
  for (auto& m : p->meta) m.times.assign(n, 0.0);
  p->trk_ev_of.assign(R, -1);
  auto new_events = [&](std::vector<hipEvent_t>& v, size_t cnt) {
    for (size_t i = 0; i < cnt; i++) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
      v.push_back(e);
    }
    return true;
  };
  bool ok = new_events(p->ev_in, (size_t)p->depth) && new_events(p->ev_det, R) && new_events(p->ev_trk, R);
  // host-fed mode: H2D copies run on their own stream into a ring of depth + 2 staging buffers, i.e. up to two steps ahead of the
  // forwards, so that a slot's forward never waits for its own copy
  p->stage.resize((size_t)p->depth + 2);
  for (auto& s : p->stage)
    ok = ok && hipEventCreateWithFlags(&s.free_ev, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.copy_ev, hipEventDisableTiming) == hipSuccess;
  if (!ok) { set_error("vbt_pipeline_create: hipEventCreate failed"); return fail(VBT_ERR_HIP); }
  p->resized.assign((size_t)p->depth, nullptr);
  p->clip_frames.assign(n, 0);
  {
    // the pipeline's own HIP streams (detector slots, copy, tracker): each is bound to its hardware queue at creation
    // (vbt_stream_create); then checked pair by pair
    std::lock_guard<std::mutex> lock(g_pool_mu);
    StreamPool& pool = g_pools[p->device];
    int role_idx[10] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    for (int k = 0; k < p->depth && rc == VBT_OK; k++) rc = pool_take(p, pool, -1, false, &role_idx[k]);
    if (rc == VBT_OK) rc = pool_take(p, pool, -1, false, &role_idx[8]);
    if (rc == VBT_OK) rc = pool_take(p, pool, -1, false, &role_idx[9]);
    if (rc == VBT_OK) rc = place_streams(p, pool, role_idx);
    if (rc == VBT_OK) {
      for (int k = 0; k < p->depth; k++) p->det_streams[k] = pool.streams[role_idx[k]];
      p->copy_stream = pool.streams[role_idx[8]];
      p->trk_stream = pool.streams[role_idx[9]];
    }
  }
  if (rc != VBT_OK) return fail(rc);
  // Self-check: every slot runs its whole plan on a blank batch before the first real frame, so a plan the kernels reject (LDS
  // budget, tile shape) fails here and not in the middle of a clip; the first real step then also finds code objects, arenas and
  // GPU clocks warm.  Detector only: no tracker state is touched.
  const int n_check = prm->selfcheck >= 0 ? prm->selfcheck : env_int("VBT_PIPELINE_SELFCHECK", 1);
  if (n_check > 0) {
    uint8_t* blank = nullptr;
    const size_t bytes = n * p->size * p->size * 3;
    if (hipMalloc((void**)&blank, bytes + 64) != hipSuccess) { set_error("vbt_pipeline_create: hipMalloc of the self-check batch failed"); return fail(VBT_ERR_HIP); }
    (void)hipMemset(blank, 0, bytes);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < n_check && rc == VBT_OK; i++)
      for (int k = 0; k < p->depth && rc == VBT_OK; k++)
        rc = vbt_detect_async(p->models[k], blank, p->n, (void*)p->det_streams[k], boxes_of(p, k), scores_of(p, k), classes_of(p, k), counts_of(p, k));
    hipError_t e = hipSuccess;
    for (int k = 0; k < p->depth; k++) {
      const hipError_t ek = hipStreamSynchronize(p->det_streams[k]);
      if (e == hipSuccess) e = ek;
    }
    (void)hipFree(blank);
    if (rc != VBT_OK) return fail(rc);
    if (e != hipSuccess) { set_error("vbt_pipeline_create: self-check forward failed: %s", hipGetErrorString(e)); return fail(VBT_ERR_HIP); }
  }
  *out = p;
  return VBT_OK;
}

int vbt_pipeline_step(vbt_pipeline* p, const uint8_t* frames, int frames_on_device, int src_h, int src_w, int swap_rb, const uint8_t* active,
                      const int32_t* clip_map, const int32_t* frame_idx, int track, void* caller_stream) {
  if (!p || !frames) { set_error("vbt_pipeline_step: NULL argument"); return VBT_ERR_ARG; }
  if ((clip_map != nullptr) != (frame_idx != nullptr)) { set_error("vbt_pipeline_step: clip_map and frame_idx come together"); return VBT_ERR_ARG; }
  if ((src_h > 0) != (src_w > 0)) { set_error("vbt_pipeline_step: src_h and src_w come together"); return VBT_ERR_ARG; }
  if (active && p->n_trk != p->n) { set_error("vbt_pipeline_step: `active` needs one clip per slot"); return VBT_ERR_ARG; }
  if (clip_map)
    for (int i = 0; i < p->n; i++)
      if (clip_map[i] >= p->n_trk) { set_error("vbt_pipeline_step: slot %d -> clip %d, the pipeline follows %d clips", i, clip_map[i], p->n_trk); return VBT_ERR_ARG; }
  if (!clip_map && track && p->n_trk > p->n) { set_error("vbt_pipeline_step: %d clips on %d slots needs clip_map / frame_idx (or vbt_pipeline_step_runs)", p->n_trk, p->n); return VBT_ERR_ARG; }
  StepTimer timer(p);
  VBT_HIP_CHECK(hipSetDevice(p->device));
  const int o = p->step_idx % p->ring, k = o % p->depth;   // output slot; forward slot (model instance, stream)
  const bool plain = !clip_map && !active && track;
  if (!p->group.empty()) {
    const std::vector<int>& g = p->group;
    // ring wrap, another kind of step, or skip_frames() changed the frame stride
    if (!plain || o <= g.back() ||
        (g.size() >= 2 && p->frame_count + 1 - p->meta[g.back()].fc != p->meta[g[1]].fc - p->meta[g[0]].fc))
      PL_CHECK(flush_group(p));
  }
  hipStream_t S = p->det_streams[k];
  PL_CHECK(wait_slot_free(p, o, S));
  p->step_idx++;
  p->frame_count++;
  Sources src;
  src.frames = frames;
  src.on_device = frames_on_device != 0;
  src.src_h = src_h; src.src_w = src_w; src.swap_rb = swap_rb;
  const uint8_t* fd = nullptr;
  int stage_j = -1;
  PL_CHECK(prepare_frames(p, k, S, src, nullptr, 0, p->n, caller_stream, &fd, &stage_j));
  vbt_pipeline::SlotMeta& m = p->meta[o];
  m.kind = vbt_pipeline::PLAIN;
  m.fc = p->frame_count;
  m.B = p->n;
  if (clip_map) {
    m.kind = vbt_pipeline::SLOTS;
    m.cmap.assign(clip_map, clip_map + p->n);
    for (int i = 0; i < p->n; i++) m.times[i] = clip_map[i] >= 0 ? (double)frame_idx[i] / p->fps[clip_map[i]] : -1.0;
  } else if (!active) {
    for (int i = 0; i < p->n; i++) m.times[i] = (double)p->frame_count / p->fps[std::min(i, p->n_trk - 1)];   // time = frame_count / fps (track.py:169)
  } else {
    for (int i = 0; i < p->n; i++) {
      if (active[i]) {
        p->clip_frames[i]++;
        m.times[i] = (double)p->clip_frames[i] / p->fps[i];
      } else {
        m.times[i] = -1.0;
      }
    }
  }
  PL_CHECK(vbt_detect_async(p->models[k], fd, p->n, (void*)S, boxes_of(p, o), scores_of(p, o), classes_of(p, o), counts_of(p, o)));
  PL_CHECK(after_detect(p, o, stage_j, S));
  p->last_B = p->n;
  if (!track) return VBT_OK;   // detector-only step (measurement splits)
  p->pending.push_back(o);
  if (p->defer && plain) {
    p->group.push_back(o);
    if ((int)p->group.size() >= p->defer || o % p->defer == p->defer - 1) PL_CHECK(flush_group(p));   // groups are aligned: their slots never wrap
    return VBT_OK;
  }
  // own stream: keep depth-1 detector steps ahead of the tracker; inline: the step follows its forward directly
  while ((int)p->pending.size() >= (p->trk_inline ? 1 : p->depth)) {
    const int q = p->pending.front();
    p->pending.erase(p->pending.begin());
    PL_CHECK(enqueue_tracker(p, q));
  }
  return VBT_OK;
}

int vbt_pipeline_step_runs(vbt_pipeline* p, const uint8_t* frames, const uint8_t* const* run_sources, int frames_on_device, const vbt_run* runs,
                           int n_runs, int src_h, int src_w, int swap_rb, int track, float* out_boxes, float* out_scores, float* out_classes,
                           int32_t* out_counts, void* caller_stream) {
  if (!p || !runs || n_runs < 1 || ((frames != nullptr) == (run_sources != nullptr))) {
    set_error("vbt_pipeline_step_runs: runs and exactly one of frames / run_sources required");
    return VBT_ERR_ARG;
  }
  const bool outs = out_boxes || out_scores || out_classes || out_counts;
  if (outs && (!out_boxes || !out_scores || !out_classes || !out_counts || track)) {
    set_error("vbt_pipeline_step_runs: out_* come together and are for detector-only steps (track = 0)");
    return VBT_ERR_ARG;
  }
  if ((src_h > 0) != (src_w > 0)) { set_error("vbt_pipeline_step_runs: src_h and src_w come together"); return VBT_ERR_ARG; }
  StepTimer timer(p);
  VBT_HIP_CHECK(hipSetDevice(p->device));
  std::vector<vbt_run> ra(runs, runs + n_runs);
  int B = 0;
  for (int i = 0; i < n_runs; i++) {
    vbt_run& r = ra[i];
    if (r.slot_stride == 0) r.slot_stride = 1;
    if (r.frame_step == 0) r.frame_step = 1;
    if (r.clip < 0 || r.clip >= p->n_trk || r.n_frames < 1 || r.slot0 < 0 || r.slot_stride != 1 || (long)r.slot0 + r.n_frames > p->n) {
      set_error("run %d: clip %d, slots %d..%ld outside %d clips / %d slots", i, r.clip, r.slot0, (long)r.slot0 + r.n_frames - 1, p->n_trk, p->n);
      return VBT_ERR_ARG;
    }
    if (!(r.fps > 0.0)) r.fps = p->fps[r.clip];
    if (run_sources && !run_sources[i]) { set_error("run %d: NULL source", i); return VBT_ERR_ARG; }
    B = std::max(B, r.slot0 + r.n_frames);
  }
  if (run_sources) {   // raw pointers base + f * frame_bytes go to the gather kernel / the copies: a hole would be detected on garbage
    std::vector<char> used((size_t)B, 0);
    for (const vbt_run& r : ra)
      for (int f = 0; f < r.n_frames; f++) used[(size_t)r.slot0 + f] = 1;
    for (char u : used)
      if (!u) { set_error("vbt_pipeline_step_runs: the runs leave a hole in the detector batch"); return VBT_ERR_ARG; }
  }
  Sources src;
  src.frames = frames;
  src.run_sources = run_sources;
  src.on_device = frames_on_device != 0;
  src.src_h = src_h; src.src_w = src_w; src.swap_rb = swap_rb;
  std::vector<vbt_run> asm_runs(ra);
  return step_runs_impl(p, src, asm_runs.data(), n_runs, ra, B, track, out_boxes, out_scores, out_classes, out_counts, caller_stream);
}

int vbt_pipeline_skip_frames(vbt_pipeline* p, int n) {
  if (!p || n < 0) { set_error("vbt_pipeline_skip_frames: bad argument"); return VBT_ERR_ARG; }
  p->frame_count += n;
  return VBT_OK;
}

int vbt_pipeline_set_frame_count(vbt_pipeline* p, int frame_count) {
  if (!p || frame_count < 0) { set_error("vbt_pipeline_set_frame_count: bad argument"); return VBT_ERR_ARG; }
  p->frame_count = frame_count;
  return VBT_OK;
}

int vbt_pipeline_reset(vbt_pipeline* p) {
  if (!p) { set_error("NULL pipeline"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(p->device));
  PL_CHECK(drain(p));
  VBT_HIP_CHECK(hipDeviceSynchronize());
  PL_CHECK(vbt_tracker_reset(p->trk));
  p->frame_count = 0;
  p->step_idx = 0;
  std::fill(p->trk_ev_of.begin(), p->trk_ev_of.end(), -1);
  p->last_trk = -1;
  std::fill(p->clip_frames.begin(), p->clip_frames.end(), 0);
  p->step_host_ns = p->step_calls = 0;
  return VBT_OK;
}

int vbt_pipeline_join_detectors(vbt_pipeline* p, void* stream) {
  if (!p) { set_error("NULL pipeline"); return VBT_ERR_ARG; }
  for (hipEvent_t e : p->ev_det) VBT_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, e, 0));
  return VBT_OK;
}

int vbt_pipeline_drain(vbt_pipeline* p) {
  if (!p) { set_error("NULL pipeline"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(p->device));
  return drain(p);
}

int vbt_pipeline_finish(vbt_pipeline* p) {
  if (!p) { set_error("NULL pipeline"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(p->device));
  PL_CHECK(drain(p));
  PL_CHECK(vbt_tracker_finish(p->trk, p->prm.plate_diameter, p->prm.diff_threshold, p->prm.min_distance, (void*)p->trk_stream));
  VBT_HIP_CHECK(hipStreamSynchronize(p->trk_stream));
  return VBT_OK;
}

int vbt_pipeline_close(vbt_pipeline* p, int32_t* best_ids, int32_t* n_rows, int32_t* n_phases, int32_t* overflow, double* phases6, int cap) {
  if (!p) { set_error("NULL pipeline"); return VBT_ERR_ARG; }
  if (!best_ids || !n_rows || !n_phases || !overflow || !phases6) {
    if (best_ids || n_rows || n_phases || overflow || phases6) { set_error("vbt_pipeline_close: the output arrays come together"); return VBT_ERR_ARG; }
    return vbt_pipeline_finish(p);
  }
  VBT_HIP_CHECK(hipSetDevice(p->device));
  PL_CHECK(drain(p));
  PL_CHECK(vbt_tracker_finish(p->trk, p->prm.plate_diameter, p->prm.diff_threshold, p->prm.min_distance, (void*)p->trk_stream));
  return vbt_tracker_summary(p->trk, best_ids, n_rows, n_phases, overflow, phases6, cap);
}

int vbt_pipeline_rows_all(vbt_pipeline* p, int32_t* counts, void* rows_host, int cap) {
  if (!p) { set_error("NULL pipeline"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(p->device));
  PL_CHECK(drain(p));
  return vbt_tracker_rows_all(p->trk, counts, rows_host, cap, (void*)p->trk_stream);
}

int vbt_pipeline_rows(vbt_pipeline* p, int clip, int64_t* id, double* cols7, int cap, int* n) {
  if (!p) { set_error("NULL pipeline"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(p->device));
  PL_CHECK(drain(p));
  VBT_HIP_CHECK(hipStreamSynchronize(p->trk_stream));
  return vbt_tracker_rows(p->trk, clip, id, cols7, cap, n);
}

int vbt_pipeline_detections(vbt_pipeline* p, float* boxes, float* scores, float* classes, int32_t* counts, int cap_slots, int* B) {
  if (!p || !boxes || !scores || !classes || !counts || !B) { set_error("vbt_pipeline_detections: NULL argument"); return VBT_ERR_ARG; }
  if (p->step_idx < 1) { set_error("vbt_pipeline_detections: no step yet"); return VBT_ERR_STATE; }
  const int o = (p->step_idx - 1) % p->ring, nb = p->last_B;
  if (nb > cap_slots) { set_error("vbt_pipeline_detections: %d slots, buffers hold %d", nb, cap_slots); return VBT_ERR_CAPACITY; }
  hipStream_t S = p->det_streams[o % p->depth];
  const size_t md = VBT_MAX_DETECTIONS;
  VBT_HIP_CHECK(hipMemcpyAsync(boxes, boxes_of(p, o), nb * md * 16, hipMemcpyDeviceToHost, S));
  VBT_HIP_CHECK(hipMemcpyAsync(scores, scores_of(p, o), nb * md * 4, hipMemcpyDeviceToHost, S));
  VBT_HIP_CHECK(hipMemcpyAsync(classes, classes_of(p, o), nb * md * 4, hipMemcpyDeviceToHost, S));
  VBT_HIP_CHECK(hipMemcpyAsync(counts, counts_of(p, o), (size_t)nb * 4, hipMemcpyDeviceToHost, S));
  VBT_HIP_CHECK(hipStreamSynchronize(S));
  *B = nb;
  return VBT_OK;
}

int vbt_pipeline_tracker_only_steps(vbt_pipeline* p, int count, int slot) {
  if (!p || count < 1 || slot < 0 || slot >= p->ring) { set_error("vbt_pipeline_tracker_only_steps: bad argument"); return VBT_ERR_ARG; }
  if (p->n_trk != p->n) { set_error("vbt_pipeline_tracker_only_steps needs one tracker clip per detector slot"); return VBT_ERR_STATE; }
  PL_CHECK(flush_group(p));   // (deferred steps first: tracker launches stay in frame order)
  hipStream_t T = p->trk_stream;
  VBT_HIP_CHECK(hipStreamWaitEvent(T, p->ev_det[slot], 0));
  if (p->last_trk >= 0) VBT_HIP_CHECK(hipStreamWaitEvent(T, p->ev_trk[p->last_trk], 0));
  std::vector<double> tm((size_t)p->n);
  for (int i = 0; i < count; i++) {
    p->frame_count++;
    for (int c = 0; c < p->n; c++) tm[c] = (double)p->frame_count / p->fps[c];
    PL_CHECK(vbt_tracker_update_from_detections(p->trk, boxes_of(p, slot), scores_of(p, slot), counts_of(p, slot), tm.data(), p->prm.detection_threshold, (void*)T));
  }
  return record_trk(p, slot, T);
}

int vbt_pipeline_get_info(const vbt_pipeline* p, vbt_pipeline_info* out) {
  if (!p || !out) { set_error("vbt_pipeline_get_info: NULL argument"); return VBT_ERR_ARG; }
  memset(out, 0, sizeof(*out));
  out->n_slots = p->n; out->n_clips = p->n_trk; out->rows_cap = p->prm.rows_cap; out->device = p->device; out->depth = p->depth;
  out->ring = p->ring; out->defer = p->defer; out->tracker_inline = p->trk_inline ? 1 : 0; out->image_size = p->size;
  out->frame_count = p->frame_count; out->steps_enqueued = p->step_idx; out->placement_ok = p->placement_ok ? 1 : 0;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    out->queue_groups_seen = (int)g_pools[p->device].reps.size();
  }
  for (int k = 0; k < p->depth; k++) out->det_streams[k] = (void*)p->det_streams[k];
  out->copy_stream = (void*)p->copy_stream;
  out->tracker_stream = (void*)p->trk_stream;
  out->h2d_bytes = p->h2d_bytes;
  out->step_host_ns = p->step_host_ns;
  out->step_calls = p->step_calls;
  return VBT_OK;
}

vbt_model* vbt_pipeline_model(vbt_pipeline* p, int k) { return (p && k >= 0 && k < p->depth) ? p->models[(size_t)k] : nullptr; }
vbt_tracker* vbt_pipeline_tracker(vbt_pipeline* p) { return p ? p->trk : nullptr; }

int vbt_track_clip(vbt_pipeline* p, const uint8_t* frames, int frames_on_device, int T, int src_h, int src_w, int swap_rb, int frame_stride,
                   int64_t* id, double* cols7, int cap, int* n_rows) {
  if (!p || !frames || T < 0 || !id || !cols7 || !n_rows || cap < 0) { set_error("vbt_track_clip: bad argument"); return VBT_ERR_ARG; }
  if (p->n_trk != 1) { set_error("vbt_track_clip: the pipeline must follow exactly one clip (n_clips = 1), it follows %d", p->n_trk); return VBT_ERR_ARG; }
  if ((src_h > 0) != (src_w > 0)) { set_error("vbt_track_clip: src_h and src_w come together"); return VBT_ERR_ARG; }
  const int stride = std::max(frame_stride, 1);
  const int H = src_h > 0 ? src_h : p->size, W = src_w > 0 ? src_w : p->size;
  const size_t fb = (size_t)H * W * 3;
  PL_CHECK(vbt_pipeline_reset(p));
  // frames whose 1-based number is not a multiple of the stride are read and dropped (track.py:161-167): they only advance the time
  const int kept = T / stride, F = p->n;
  for (int i0 = 0; i0 < kept; i0 += F) {
    const int nf = std::min(F, kept - i0);
    const int first = (i0 + 1) * stride;   // 1-based frame number of the chunk's first kept frame
    vbt_run run{0, 0, 1, nf, first, stride, p->fps[0]};
    if (stride == 1) {
      PL_CHECK(vbt_pipeline_step_runs(p, frames + (size_t)(first - 1) * fb, nullptr, frames_on_device, &run, 1, src_h, src_w, swap_rb, 1, nullptr, nullptr,
                                      nullptr, nullptr, nullptr));
    } else {
      // one single-frame run per kept frame: sources `stride` frames apart, assembled by the gather launch / the copy stream
      std::vector<vbt_run> rr((size_t)nf);
      std::vector<const uint8_t*> srcs((size_t)nf);
      for (int f = 0; f < nf; f++) {
        rr[(size_t)f] = vbt_run{0, f, 1, 1, first + f * stride, 1, p->fps[0]};
        srcs[(size_t)f] = frames + (size_t)(first + f * stride - 1) * fb;
      }
      // (the tracker sees ONE run for the clip: assembled from the per-frame sources, walked as the single run)
      VBT_HIP_CHECK(hipSetDevice(p->device));
      Sources src;
      src.run_sources = srcs.data();
      src.on_device = frames_on_device != 0;
      src.src_h = src_h; src.src_w = src_w; src.swap_rb = swap_rb;
      std::vector<vbt_run> walk(1, run);
      PL_CHECK(step_runs_impl(p, src, rr.data(), nf, walk, nf, 1, nullptr, nullptr, nullptr, nullptr, nullptr));
    }
  }
  PL_CHECK(vbt_pipeline_finish(p));
  return vbt_tracker_rows(p->trk, 0, id, cols7, cap, n_rows);
}

int vbt_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) { set_error("vbt_host_alloc: bad argument"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return VBT_OK;
}
int vbt_host_free(void* ptr) {
  if (ptr) VBT_HIP_CHECK(hipHostFree(ptr));
  return VBT_OK;
}
int vbt_device_alloc(int device, size_t bytes, void** out) {
  if (!out || bytes == 0) { set_error("vbt_device_alloc: bad argument"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipSetDevice(device));
  VBT_HIP_CHECK(hipMalloc(out, bytes));
  return VBT_OK;
}
int vbt_device_free(void* ptr) {
  if (ptr) VBT_HIP_CHECK(hipFree(ptr));
  return VBT_OK;
}
int vbt_memcpy(void* dst, const void* src, size_t bytes, int kind) {
  if (!dst || !src || kind < 0 || kind > 1) { set_error("vbt_memcpy: bad argument"); return VBT_ERR_ARG; }
  VBT_HIP_CHECK(hipMemcpy(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
  return VBT_OK;
}
int vbt_stream_synchronize(void* stream) {
  VBT_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return VBT_OK;
}
int vbt_device_synchronize(int device) {
  VBT_HIP_CHECK(hipSetDevice(device));
  VBT_HIP_CHECK(hipDeviceSynchronize());
  return VBT_OK;
}

}  // extern "C"
