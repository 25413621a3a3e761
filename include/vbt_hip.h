/*
 * vbt_hip.h - C ABI of libvbt_hip.so, the MI355X (gfx950) drop-in for the hot loop of the
 * reference's track.py (reference track.py:159-247) and plot.py:analyze_df (plot.py:33-47).
 *
 * The reference has no FFI layer of its own: its hot path sits behind duck-typed Python objects
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it replaces;
 * the Python classes in vbt_amd/ bind them with ctypes and keep the reference's call shapes.
 *
 * Conventions: every function returns 0 on success or a negative vbt_status; the message of the
 * last failure on the calling thread is vbt_last_error().  Handles are opaque, owned by the
 * library, used by one host thread at a time.  "dev" pointers are HIP device pointers; `stream`
 * is a hipStream_t passed as void* (NULL = the default stream).  No torch types anywhere.
 */
#ifndef VBT_HIP_H
#define VBT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VBT_OK = 0,
  VBT_ERR_ARG = -1,      /* bad argument (NULL, out of range, shape mismatch) */
  VBT_ERR_IO = -2,       /* container file missing / malformed */
  VBT_ERR_HIP = -3,      /* a HIP runtime call failed */
  VBT_ERR_CAPACITY = -4, /* batch / row / track capacity exceeded */
  VBT_ERR_STATE = -5     /* call sequence error */
} vbt_status;

#define VBT_MAX_DETECTIONS 25

const char* vbt_last_error(void);
int vbt_device_count(void);

/* ------------------------------------------------------------------ detector ----------------
 * Replaces tflite_runtime.interpreter.Interpreter (reference track.py:93-94, eval.py:167-168):
 *   Interpreter(model_path, num_threads) + allocate_tensors()        -> vbt_model_create
 *   get_input_details()[0]['shape']            (reference odt.py:86-87) -> vbt_model_input_shape
 *   get_signature_runner()(images=uint8[1,H,W,3]) (reference odt.py:58-66) -> vbt_detect
 */
typedef struct vbt_model vbt_model;

/* container_path: a VBTM model container (vbt_amd/container.py); max_batch frames per vbt_detect. */
int vbt_model_create(const char* container_path, int device, int max_batch, vbt_model** out);
/* flags = 0 (VBT_MODEL_DEFAULT_FLAGS): the fused, autotuned execution plan.  The plan-shaping flags used by the parity tests and
 * the tuning tools are lab equipment and live in vbt_hip_diag.h.  vbt_model_create == _ex with flags 0 (or the integer in the
 * environment variable VBT_FUSION_FLAGS). */
#define VBT_MODEL_DEFAULT_FLAGS 0
int vbt_model_create_ex(const char* container_path, int device, int max_batch, int flags, vbt_model** out);
void vbt_model_destroy(vbt_model* m);
/* shape = {max_batch, H, W, 3} */
int vbt_model_input_shape(const vbt_model* m, int shape[4]);
int vbt_model_num_tensors(const vbt_model* m);
int vbt_model_num_ops(const vbt_model* m);
/* shape = {H, W, C} of graph tensor `tensor_id` */
int vbt_model_tensor_shape(const vbt_model* m, int tensor_id, int shape[3]);

/* frames: uint8 [B,H,W,3] RGB (host pointer if frames_on_device == 0, else device pointer).
 * Outputs follow the TFLite_Detection_PostProcess signature read at reference odt.py:64-66:
 *   boxes  float32 [B,25,4]  (ymin,xmin,ymax,xmax) normalised   = output_3
 *   scores float32 [B,25]                                        = output_1
 *   classes float32 [B,25]                                       = output_2
 *   counts int32   [B]                                           = output_0
 * Output pointers are host (outputs_on_device == 0; the call synchronises the stream) or device. */
int vbt_detect(vbt_model* m, const uint8_t* frames, int B, int frames_on_device, void* stream,
               float* boxes, float* scores, float* classes, int32_t* counts, int outputs_on_device);

/* Enqueue only (no copies, no sync): frames and outputs are device pointers. Used by the fused
 * pipeline and by bench.py inside HIP-event brackets. */
int vbt_detect_async(vbt_model* m, const uint8_t* frames_dev, int B, void* stream,
                     float* boxes_dev, float* scores_dev, float* classes_dev, int32_t* counts_dev);

/* Streams for a pipeline that keeps several forwards in flight (the reference runs one interpreter.invoke() at a time,
 * odt.py:58-61; there is no reference counterpart).  A HIP stream is bound to a hardware queue at its first command,
 * round-robin: the stream returned here has already run one empty launch, so streams created back to back use distinct
 * queues, whatever a framework's stream pool has been used for before. */
int vbt_stream_create(int device, void** stream_out);
int vbt_stream_destroy(void* stream);

/* preprocess_image (reference odt.py:10-19): bilinear resize (half-pixel centres, float32) of
 * uint8 [B,H,W,3] frames to [B,h,w,3] + truncating uint8 cast; swap_rb != 0 also swaps channels 0
 * and 2 (cv2 BGR -> RGB, reference track.py:171).  src/dst are host or device pointers. */
int vbt_resize_frames(const uint8_t* src, int B, int H, int W, int src_on_device, uint8_t* dst, int h, int w,
                      int dst_on_device, int swap_rb, int device, void* stream);

/* ------------------------------------------------------------------ tracker -----------------
 * Replaces ocsort.OCSort (reference track.py:17,157,186-199), the row assembly of
 * reference track.py:189-234 and the export id selection of track.py:107-115.
 * One handle tracks `n_clips` independent clips in parallel (one wavefront per clip).
 */
typedef struct vbt_tracker vbt_tracker;
typedef struct {
  int32_t max_age;       /* reference track.py:22,157 -> 30 */
  int32_t min_hits;      /* OC-SORT default 3 */
  int32_t delta_t;       /* OC-SORT default 3 (1..3 supported) */
  int32_t asso;          /* second-round association: 0 = iou, 1 = diou (reference track.py:157) */
  double iou_threshold;  /* reference track.py:157 -> 0.1 */
  double inertia;        /* OC-SORT default 0.2 */
  double det_thresh;     /* OC-SORT score gate (detections with score <= det_thresh are dropped) */
} vbt_tracker_params;

/* rows_cap: capacity of the per-clip row log (rows = emitted (id,time,...) records). */
int vbt_tracker_create(int n_clips, int rows_cap, const vbt_tracker_params* p, int device, vbt_tracker** out);
void vbt_tracker_destroy(vbt_tracker* t);
int vbt_tracker_reset(vbt_tracker* t);

/* OCSort.update for F consecutive frames of every clip (host pointers; synchronous):
 *   dets   float64 [F][n_clips][25][6] = x1,y1,x2,y2,score,cls   (reference odt.py:102-118)
 *   counts int32   [F][n_clips]   (0 = empty frame: the tracker is not stepped, track.py:180-181)
 *   times  float64 [F][n_clips]   (frame_count / fps, reference track.py:169) */
int vbt_tracker_update(vbt_tracker* t, const double* dets, const int32_t* counts, const double* times, int F);

/* Fused path: one frame per clip straight from vbt_detect_async's device outputs; applies the
 * detection threshold of reference odt.py:70-75 (score >= det_threshold). times_host [n_clips]; a negative time marks a
 * clip that has no frame in this step (clips of different lengths batched together): its state is left untouched.
 * The host arrays are read during the call and travel in the kernel arguments (64 slots per launch): no host-to-device
 * copy is enqueued, the caller may reuse them as soon as the call returns. */
int vbt_tracker_update_from_detections(vbt_tracker* t, const float* boxes_dev, const float* scores_dev,
                                       const int32_t* counts_dev, const double* times_host, float det_threshold,
                                       void* stream);
/* The same with fewer detector slots than clips: slot i of the batch carries a frame of clip clip_of_slot_host[i]
 * (-1: none), times_host is per slot.  A slot can move on to the next clip of its queue when one ends, so a corpus of
 * clips of different lengths keeps every slot of the detector batch busy (the reference runs clips one after the other,
 * track.py:85-126).  Export ids / rows / phases stay per clip. */
int vbt_tracker_update_from_slots(vbt_tracker* t, const float* boxes_dev, const float* scores_dev, const int32_t* counts_dev,
                                  const int32_t* clip_of_slot_host, const double* times_host, int n_slots, float det_threshold,
                                  void* stream);

/* Time-batched form.  The reference's unit of work is ONE video (track.py:85-126): `while cap.isOpened()` reads frame
 * after frame of one clip (track.py:159-247).  The detector has no state, so a batch may hold RUNS of consecutive frames of
 * a clip instead of one frame of each of B clips; the tracker then walks every run in frame order inside one launch (one
 * wavefront per run).  Frame f (0-based) of a run sits in detector slot slot0 + f * slot_stride and is the reference's
 * frame number frame_count = frame0 + f * frame_step (track.py:161; frame_step = the `frame_count % 16` stride of
 * track.py:166, normally 1); its time stamp is frame_count / fps (track.py:169), one IEEE double division.  Frames on which
 * run_odt returns [] do not step the tracker (track.py:180-181).  A clip may appear in at most one run per call; calls on
 * one stream are ordered, so consecutive calls continue a clip.  runs_host is read during the call (kernel arguments). */
typedef struct {
  int32_t clip;         /* tracker clip the run belongs to (< 0: descriptor ignored) */
  int32_t slot0;        /* detector slot of the run's first frame */
  int32_t slot_stride;  /* 1: the run's frames are neighbours in the batch */
  int32_t n_frames;     /* frames in the run */
  int32_t frame0;       /* 1-based frame number of the first frame */
  int32_t frame_step;   /* frame-number increment per frame of the run (>= 1) */
  double fps;           /* cap.get(cv2.CAP_PROP_FPS), reference track.py:138 */
} vbt_run;
int vbt_tracker_update_from_detections_seq(vbt_tracker* t, const float* boxes_dev, const float* scores_dev,
                                           const int32_t* counts_dev, int n_slots, const vbt_run* runs_host, int n_runs,
                                           float det_threshold, void* stream);

/* Assemble a detector batch from frames that live elsewhere in device memory (frame runs of different clips, or a clip
 * whose frames are cycled): dst_dev[i] = *src_frames_host[i] for i < n_frames, frame_bytes each (a multiple of 16; all
 * pointers 16-byte aligned).  src_frames_host is an array of DEVICE pointers held in host memory; it is read during the
 * call (kernel arguments).  One launch per 64 frames, enqueued on `stream`. */
int vbt_gather_frames(uint8_t* dst_dev, const uint8_t* const* src_frames_host, int n_frames, size_t frame_bytes, void* stream);

/* What OCSort.update returned for the clip's most recent stepped frame: out7 [M,7] =
 * x1,y1,x2,y2,id(1-based),cls,score (reference track.py:190) and vel2 [M,2] = kf.x[4:6] of the
 * same track (reference track.py:194-199). */
int vbt_tracker_last_output(vbt_tracker* t, int clip, double* out7, double* vel2, int cap, int* M);
/* tracker.trackers: ids (0-based, reference track.py:195) and kf.x (7 values each) in list order */
int vbt_tracker_get_trackers(vbt_tracker* t, int clip, int32_t* ids, double* kfx, int cap, int* n);
int vbt_tracker_status(vbt_tracker* t, int clip, int32_t* n_rows, int32_t* n_trackers, int32_t* overflow,
                       int32_t* rows_overflow, int32_t* frame_count);
/* The clip's row log in emission order = the dict of reference track.py:144-145,227-234:
 * id int64 [n]; cols7 float64 [n,7] = time,x,y,dx,dy,norm_plate_height,norm_plate_width */
int vbt_tracker_rows(vbt_tracker* t, int clip, int64_t* id, double* cols7, int cap, int* n);

/* End of clips: pick each clip's id with the largest cumulative path length (reference
 * track.py:107-115) and run plot.py's preprocessing + VelocityTracker over its rows on device. */
int vbt_tracker_finish(vbt_tracker* t, double plate_diameter, double diff_threshold, double min_distance, void* stream);
/* phases6 [P,6] = time_start,time_end,y_start,y_end,rom,type (reference Phase.py:16-22) */
int vbt_tracker_phases(vbt_tracker* t, int clip, int32_t* best_id, double* phases6, int cap, int* P);
/* Clip close of every clip at once (arrays of n_clips entries; phases6 is [n_clips][cap][6], rows beyond n_phases[c] are
 * not written): export ids, DataFrame row counts, phase counts and overflow flags are packed on the device into one
 * block, fetched by ONE asynchronous copy into pinned memory and ONE synchronisation of the stream vbt_tracker_finish ran
 * on (not of the device). */
int vbt_tracker_summary(vbt_tracker* t, int32_t* best_ids, int32_t* n_rows, int32_t* n_phases, int32_t* overflow, double* phases6, int cap);
/* DataFrame rows (reference track.py:227-234) of EVERY clip in one strided device-to-host copy on `stream` (which the
 * call synchronises): rows_host = [n_clips][cap] records of 64 bytes {int64 id; double time, x, y, dx, dy,
 * norm_plate_height, norm_plate_width}, all ids in emission order; counts[c] = number of rows of clip c.
 * rows_host may be pinned host memory (one DMA) or pageable. */
int vbt_tracker_rows_all(vbt_tracker* t, int32_t* counts, void* rows_host, int cap, void* stream);

/* ------------------------------------------------------------------ pipeline ----------------
 * Replaces the clip loop of the reference (track.py:129-260: `while cap.isOpened()` -> cap.read -> run_odt -> OCSort.update ->
 * row assembly, then the export of track.py:103-126 and analyze_df of plot.py:33-47) as ONE object behind the C ABI: SURVEY.md 8b's
 * "fused on-device path".  The handle owns everything the fast path needs and nothing of it lives in the caller's language:
 *   - `depth` detector instances (activation arenas) on `depth` HIP streams of their own, each checked at creation to sit on its own
 *     hardware queue (a pipeline whose busy streams share a queue loses a third of its throughput); the reference runs one
 *     interpreter.invoke() at a time (odt.py:58-61) - the detector has no state, so consecutive steps may overlap;
 *   - a ring of detector output slots and the events that order detector(t) -> tracker(t) -> slot reuse; the OC-SORT steps stay in
 *     frame order (track.py:186 is sequential per clip);
 *   - a copy stream with a ring of depth + 2 device staging buffers: frames handed over in host memory (the reference's situation,
 *     track.py:160) are uploaded up to two steps ahead of their forward; with frames at source resolution only the rows the bilinear
 *     resize of odt.py:10-19 reads are uploaded;
 *   - for batches of <= 8 frames, groups of `depth` tracker steps walked by one launch (the forward of a small batch is launch latency).
 * One handle is used by one host thread at a time.  Every call only ENQUEUES work unless its comment says it synchronises.
 * Frames: uint8 [*,H,W,3]; H,W = the network resolution (vbt_model_input_shape) unless src_h/src_w say otherwise.
 */
typedef struct vbt_pipeline vbt_pipeline;
typedef struct {
  int32_t n_slots;           /* frames per detector batch (>= 1) */
  int32_t n_clips;           /* clips the tracker follows; 0 = n_slots (one clip per slot) */
  int32_t rows_cap;          /* capacity of each clip's row log (rows = emitted (id,time,...) records) */
  int32_t device;
  int32_t depth;             /* forwards in flight, 1..8; 0 = default (environment VBT_PIPELINE_DEPTH, else 4 for n_slots <= 8, else 3) */
  int32_t tracker_stream;    /* where a frame's OC-SORT step runs: 0 = default (VBT_TRACKER_STREAM, else inline from depth 3), 1 = on a
                                stream of its own, 2 = inline at the end of the forward's stream */
  int32_t defer;             /* tracker steps of a small batch walked in groups of `depth`: -1 = default (VBT_TRACKER_DEFER, else on
                                for n_slots <= 8 with one clip per slot), 0 = off, 1 = on */
  int32_t selfcheck;         /* forwards on a blank batch per detector instance at creation: -1 = default (VBT_PIPELINE_SELFCHECK, else 1) */
  int32_t strict_placement;  /* busy streams that cannot be put on distinct hardware queues: 1 = vbt_pipeline_create fails with
                                VBT_ERR_STATE, 0 = a note on stderr, -1 = environment VBT_STRICT_PLACEMENT (default 0) */
  int32_t model_flags;       /* vbt_model_create_ex flags; VBT_MODEL_DEFAULT_FLAGS */
  float detection_threshold; /* detection_treshold of reference track.py:129,174 / odt.py:70-75 */
  float reserved0;
  double plate_diameter;     /* VelocityTracker(plate_diameter, diff_threshold, min_distance), reference VelocityTracker.py:16 */
  double diff_threshold;
  double min_distance;
  vbt_tracker_params tracker; /* OCSort(...) of reference track.py:157 */
} vbt_pipeline_params;
/* the reference's settings: OCSort(max_age=30, asso_func="diou", iou_threshold=0.1) (track.py:22,157), threshold 0.5,
 * VelocityTracker(0.45, 0.6, 0.1); everything else 0 / -1 = default */
void vbt_pipeline_default_params(vbt_pipeline_params* p);
/* fps_host [n_clips]: cap.get(cv2.CAP_PROP_FPS) of every clip (reference track.py:138) */
int vbt_pipeline_create(const char* container_path, const vbt_pipeline_params* p, const double* fps_host, vbt_pipeline** out);
void vbt_pipeline_destroy(vbt_pipeline* p);

/* One frame of every clip: frames = uint8 [n_slots,H,W,3], frame number frame_count + 1 of each clip (track.py:161), time stamp
 * frame_count / fps (track.py:169).  frames_on_device != 0: a device pointer, valid on `caller_stream` when the call is made (the
 * forward waits for that point of the stream) and left untouched until the step has run (up to `depth` steps later).
 * frames_on_device == 0: host memory, final when the call is made; pinned memory is copied by DMA on the copy stream up to two steps
 * ahead and must stay untouched until the step has run; the call blocks only when the caller is depth + 2 steps ahead of the GPU.
 * src_h / src_w > 0: the frames are at source resolution and go through preprocess_image (odt.py:10-19) on the device; swap_rb: BGR
 * input (track.py:171).  active [n_slots] (or NULL): clips that have a frame in this step; the others keep their state and frame
 * counter.  clip_map / frame_idx [n_slots] (or NULL): slot i carries frame number frame_idx[i] (1-based) of clip clip_map[i] (-1:
 * empty slot) - more clips than slots.  track == 0: detector only (measurement splits). */
int vbt_pipeline_step(vbt_pipeline* p, const uint8_t* frames, int frames_on_device, int src_h, int src_w, int swap_rb,
                      const uint8_t* active, const int32_t* clip_map, const int32_t* frame_idx, int track, void* caller_stream);
/* Time-batched step: the batch holds RUNS of consecutive frames of a clip (vbt_run; fps <= 0 = the clip's own, slot_stride 0 = 1);
 * OC-SORT walks every run in frame order inside one launch.  Either `frames` is the assembled batch [B,H,W,3] (B = slots the runs
 * cover) or it is NULL and run_sources[i] points at the n_frames contiguous frames of run i (all device or all host): the batch is
 * then assembled here (one gather launch, or one copy per run on the copy stream).  The runs must cover slots 0..B-1 without a hole.
 * out_* != NULL (with track == 0): the detections go to these device buffers ([B,25,4], [B,25], [B,25], [B]) instead of the ring -
 * the frame-major multi-GPU mode collects them for its gather. */
int vbt_pipeline_step_runs(vbt_pipeline* p, const uint8_t* frames, const uint8_t* const* run_sources, int frames_on_device,
                           const vbt_run* runs, int n_runs, int src_h, int src_w, int swap_rb, int track,
                           float* out_boxes, float* out_scores, float* out_classes, int32_t* out_counts, void* caller_stream);
/* frames read from the source but not processed (`frame_count % 16`, track.py:161-167): they advance the clip time only */
int vbt_pipeline_skip_frames(vbt_pipeline* p, int n);
int vbt_pipeline_set_frame_count(vbt_pipeline* p, int frame_count);
/* back to frame 0 of fresh clips; models, streams and buffers are kept (synchronises) */
int vbt_pipeline_reset(vbt_pipeline* p);
/* `stream` waits for every forward enqueued so far (after detector-only steps their outputs are then safe to read on it) */
int vbt_pipeline_join_detectors(vbt_pipeline* p, void* stream);
/* Clip close (track.py:103-126 export id, plot.py:33-47,87-95 rep analysis) of every clip: drains the pipeline, runs both on the
 * device, then ONE packed copy and ONE stream synchronisation.  Arrays of n_clips entries; phases6 [n_clips][cap][6].  Any of the
 * output pointers may be NULL (vbt_pipeline_finish = close without the read-back, synchronises too). */
int vbt_pipeline_close(vbt_pipeline* p, int32_t* best_ids, int32_t* n_rows, int32_t* n_phases, int32_t* overflow, double* phases6, int cap);
int vbt_pipeline_finish(vbt_pipeline* p);
/* enqueue every tracker step still held back (deferred groups, the `depth - 1` steps the own-stream mode keeps ahead); no synchronisation */
int vbt_pipeline_drain(vbt_pipeline* p);
/* vbt_tracker_rows_all / vbt_tracker_rows / vbt_tracker_phases after draining the pipeline (synchronise) */
int vbt_pipeline_rows_all(vbt_pipeline* p, int32_t* counts, void* rows_host, int cap);
int vbt_pipeline_rows(vbt_pipeline* p, int clip, int64_t* id, double* cols7, int cap, int* n);
/* most recent step's detector outputs copied to host arrays [B,25,4], [B,25], [B,25], [B] (synchronises that forward; tests) */
int vbt_pipeline_detections(vbt_pipeline* p, float* boxes, float* scores, float* classes, int32_t* counts, int cap_slots, int* B);
/* measurement split: `count` tracker steps of all clips on the detections sitting in ring slot `slot` */
int vbt_pipeline_tracker_only_steps(vbt_pipeline* p, int count, int slot);
typedef struct {
  int32_t n_slots, n_clips, rows_cap, device, depth, ring, defer, tracker_inline, image_size, frame_count, steps_enqueued, placement_ok;
  int32_t queue_groups_seen;  /* distinct hardware queues the placement probe has seen on this device */
  int32_t reserved[3];
  void* det_streams[8];
  void* copy_stream;
  void* tracker_stream;
  uint64_t h2d_bytes;         /* bytes copied host -> device by this pipeline so far */
  uint64_t step_host_ns;      /* host time spent inside vbt_pipeline_step / _step_runs so far (enqueue cost incl. any back-pressure wait) */
  uint64_t step_calls;        /* ... over this many calls (both zeroed by vbt_pipeline_reset) */
} vbt_pipeline_info;
int vbt_pipeline_get_info(const vbt_pipeline* p, vbt_pipeline_info* out);
/* borrowed handles (owned by the pipeline): detector instance k < depth, the tracker */
vbt_model* vbt_pipeline_model(vbt_pipeline* p, int k);
vbt_tracker* vbt_pipeline_tracker(vbt_pipeline* p);

/* track(src, interpreter, detection_treshold, ...) of reference track.py:129-260 for ONE clip held in memory: T frames uint8
 * [T,H,W,3] (host or device; any resolution: src_h/src_w as above, 0 = network resolution), every frame_stride-th frame processed
 * (track.py:166), `p` created with n_clips = 1; n_slots consecutive kept frames per detector batch.  Returns the clip's rows in
 * emission order = the dict of track.py:144-145,227-234 (id int64 [n]; cols7 float64 [n,7] = time,x,y,dx,dy,h,w).  Synchronises. */
int vbt_track_clip(vbt_pipeline* p, const uint8_t* frames, int frames_on_device, int T, int src_h, int src_w, int swap_rb,
                   int frame_stride, int64_t* id, double* cols7, int cap, int* n_rows);

/* Pinned host memory for callers without a runtime of their own (frames handed to vbt_pipeline_step from it are copied by DMA) */
int vbt_host_alloc(size_t bytes, void** out);
int vbt_host_free(void* ptr);
int vbt_device_alloc(int device, size_t bytes, void** out);
int vbt_device_free(void* ptr);
/* blocking copies for the same callers (kind: 0 = host -> device, 1 = device -> host) */
int vbt_memcpy(void* dst, const void* src, size_t bytes, int kind);
int vbt_stream_synchronize(void* stream);
int vbt_device_synchronize(int device);

/* ------------------------------------------------------------------ rep analysis ------------
 * Replaces VelocityTracker (reference VelocityTracker.py:15-230) as driven by analyze_df
 * (reference plot.py:33-47): cols7 [T,7] = time,x,y,dx,dy,norm_plate_height,norm_plate_width
 * of one track. preprocess != 0 applies plot.py:90-95 (rolling(5)/expanding means) first;
 * flush != 0 runs end_processing(). */
int vbt_analyze(const double* cols7, int T, int preprocess, int flush, double plate_diameter, double diff_threshold,
                double min_distance, double* phases6, int cap, int* P, int device);

/* Trailing / expanding window means of the columns of a row-major float64 table rows[T][ncols] (ncols <= 64),
 * bit-identical to pandas `Series.rolling(window, center=False, min_periods=1).mean()` (windows[c] > 0) and
 * `Series.expanding(min_periods=1).mean()` (windows[c] == 0); windows[c] < 0 copies column c.
 * Replaces the smoothing of reference plot.py:90-95, kinovea.py:99-105 and qualysis.py:113-117. */
int vbt_window_means(const double* rows, int T, int ncols, const int32_t* windows, double* out, int device);

#ifdef __cplusplus
}
#endif
#endif /* VBT_HIP_H */
