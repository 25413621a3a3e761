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

/* Parity/debug: copy graph tensor `tensor_id` ([B,H,W,C] int8) of the last vbt_detect to host. */
int vbt_model_read_tensor(vbt_model* m, int tensor_id, int B, int8_t* host_out);

/* Enqueue only (no copies, no sync): frames and outputs are device pointers. Used by the fused
 * pipeline and by bench.py inside HIP-event brackets. */
int vbt_detect_async(vbt_model* m, const uint8_t* frames_dev, int B, void* stream,
                     float* boxes_dev, float* scores_dev, float* classes_dev, int32_t* counts_dev);

/* Per-kernel-family accounting of the last enqueued forward: fills up to `cap` entries.
 * Algorithmic bytes = inputs read once + output written once + weights once. */
typedef struct {
  char name[32];
  int launches;
  double algorithmic_bytes;
  double macs;
} vbt_kernel_stat;
int vbt_model_kernel_stats(const vbt_model* m, int B, vbt_kernel_stat* out, int cap, int* n);

/* Time each kernel family with HIP events on `stream` over `reps` forwards of batch B
 * (frames must be device-resident). ms_out[i] = average milliseconds per forward spent in
 * family i (same order as vbt_model_kernel_stats). */
int vbt_model_profile(vbt_model* m, const uint8_t* frames_dev, int B, int reps, void* stream,
                      double* ms_out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* VBT_HIP_H */
