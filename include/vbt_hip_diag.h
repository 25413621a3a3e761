/*
 * vbt_hip_diag.h - measurement and test entry points of libvbt_hip.so.  NOT part of the drop-in boundary (vbt_hip.h): nothing
 * here has a counterpart in the reference; these are what the parity tests (every graph tensor, every plan mode), the
 * autotuner tools and bench.py's roofline block use.
 */
#ifndef VBT_HIP_DIAG_H
#define VBT_HIP_DIAG_H

#include "vbt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ execution-plan flags (vbt_model_create_ex) ---------- */
/* flags: VBT_MODEL_NO_FUSION runs every graph op as its own kernel (every tensor readable by
 * vbt_model_read_tensor); the default fuses MBConv / SeparableConv blocks on LDS tiles.
 * vbt_model_create == _ex with VBT_MODEL_DEFAULT_FLAGS (or the integer in the environment variable
 * VBT_FUSION_FLAGS). */
#define VBT_MODEL_NO_FUSION 1
#define VBT_MODEL_NO_MBCONV_FUSION 2
#define VBT_MODEL_NO_SEPCONV_FUSION 4
#define VBT_MODEL_NO_NODE_FUSION 16  /* keep BiFPN resample/add ops out of the fused SeparableConv kernels */
#define VBT_MODEL_SINGLE_STREAM 32   /* do not split the batch over side streams */
#define VBT_MODEL_NO_GRAPH 64        /* never replay the forward from a captured hipGraph */
#define VBT_MODEL_NO_HEAD_BATCHING 128 /* one launch per head layer and level instead of one per layer */
#define VBT_MODEL_NO_STEM_FUSION 256  /* stem conv as its own kernel instead of stem + first SeparableConv fused */
#define VBT_MODEL_CHUNK48 2048        /* heuristic plan: 48-channel chunks in fused MBConv blocks whose expanded width allows it */
#define VBT_MODEL_IMAGE_BLOCKS 512    /* heuristic plan: whole-image MBConv kernel wherever it applies (autotuning decides otherwise) */
#define VBT_MODEL_TILE128 16384       /* heuristic plan: 128-pixel tiles in the fused MBConv blocks that allow them */
#define VBT_MODEL_NO_BAND 8192        /* do not use the row-band SeparableConv kernel of the BiFPN nodes / head layers */
#define VBT_MODEL_NO_EXPDW 4096       /* do not use the whole-image expand+depthwise kernel of the low-resolution MBConv blocks */
#define VBT_MODEL_NO_PW_MERGE 32768    /* keep the P6 conv, its pools and the lateral convs of the first BiFPN cell as separate launches */
#define VBT_MODEL_NO_AUTOTUNE 8     /* keep the heuristic plan (most fused alternative, default kernel variants) */

/* ------------------------------------------------------------------ graph introspection -------------------------------- */
/* 1 if graph tensor `tensor_id` is written to HBM by the execution plan, 0 if it only exists in LDS */
int vbt_model_tensor_materialized(const vbt_model* m, int tensor_id);
/* kernels launched per forward */
int vbt_model_num_launches(const vbt_model* m);

/* Parity/debug: copy graph tensor `tensor_id` ([B,H,W,C] int8) of the last vbt_detect to host. */
int vbt_model_read_tensor(vbt_model* m, int tensor_id, int B, int8_t* host_out);

/* ------------------------------------------------------------------ stream placement probe ----------------------------- */
/* *shared = 1 when the two streams sit on one hardware queue (their kernels cannot overlap): a single wave spins `us`
 * microseconds on each and the pair is timed.  The device must be otherwise idle. */
int vbt_streams_share_queue(void* a, void* b, int us, int* shared);

/* ------------------------------------------------------------------ per-kernel accounting and timing ------------------- */
/* Per-kernel-family accounting of the last enqueued forward: fills up to `cap` entries.
 * Algorithmic bytes = inputs read once + output written once + weights once. */
typedef struct {
  char name[32];
  int launches;
  double algorithmic_bytes;
  double macs;
} vbt_kernel_stat;
int vbt_model_kernel_stats(const vbt_model* m, int B, vbt_kernel_stat* out, int cap, int* n);
/* Per-launch profile of the execution plan (one forward in flight, HIP events around every launch; average of `reps`):
 * entry i = launch i of the forward.  `op` / `first_op` = last / first graph op the launch covers. */
typedef struct {
  char family[32];
  int op, first_op, variant;
  double ms, algorithmic_bytes, macs;
} vbt_step_time;
int vbt_model_profile_steps(vbt_model* m, const uint8_t* frames_dev, int B, int reps, void* stream, vbt_step_time* out, int cap, int* n);
/* Measurement: each plan step `reps` times on one stream (single_ms, per launch) and on `nstreams` streams at once
 * (conc_ms, wall time per launch): how much of a step the other forwards in flight can hide. */
int vbt_model_profile_overlap(vbt_model* m, int B, int reps, int nstreams, float* single_ms, float* conc_ms, int cap, int* n);

/* Time each kernel family with HIP events on `stream` over `reps` forwards of batch B
 * (frames must be device-resident). ms_out[i] = average milliseconds per forward spent in
 * family i (same order as vbt_model_kernel_stats). */
int vbt_model_profile(vbt_model* m, const uint8_t* frames_dev, int B, int reps, void* stream,
                      double* ms_out, int cap);

/* The same, measured the way rocprofv3 --kernel-trace sees it: all launches of a family back to back between ONE pair of HIP
 * events (`reps` passes), no event pair around every short launch.  ms_out[i] = milliseconds per pass of family i. */
int vbt_model_profile_families(vbt_model* m, int B, int reps, void* stream, double* ms_out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* VBT_HIP_DIAG_H */
