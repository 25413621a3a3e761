/*
 * track_clip.c - the reference's clip loop (track.py:129-260) from a plain C host over the C ABI of libvbt_hip.so: no Python, no
 * framework.  Reads a raw clip file (uint8 frames [T][H][W][3], RGB), tracks it on the time-batched device path and prints the
 * DataFrame rows of track.py:227-234 (id, time, x, y, dx, dy, norm_plate_height, norm_plate_width) as CSV.
 *
 *   gcc -O2 -std=c11 examples/track_clip.c -Iinclude -Lvbt_amd -lvbt_hip -Wl,-rpath,$PWD/vbt_amd -o track_clip
 *   ./track_clip models/efficientdet_lite0_synth.vbtm clip.raw <T> <H> <W> <fps> [frame_stride]
 *
 * (tests/test_gpu_cli.py::test_c_host_tracks_a_clip_like_the_python_wrapper builds and runs it and compares the rows with
 *  vbt_amd.track.track_frames on the same clip.)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "vbt_hip.h"

static int fail(const char* what) {
  fprintf(stderr, "%s: %s\n", what, vbt_last_error());
  return 1;
}

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s model.vbtm clip.raw T H W fps [frame_stride]\n", argv[0]);
    return 2;
  }
  const int T = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]);
  const double fps = atof(argv[6]);
  const int stride = argc > 7 ? atoi(argv[7]) : 1;
  if (T < 1 || H < 1 || W < 1 || !(fps > 0.0) || stride < 1) { fprintf(stderr, "bad clip geometry\n"); return 2; }
  const size_t bytes = (size_t)T * H * W * 3;

  /* frames in pinned host memory: the pipeline uploads them by DMA on its copy stream (at source resolution only the rows the
     bilinear resize of odt.py:15-16 reads) */
  void* frames = NULL;
  if (vbt_host_alloc(bytes, &frames) != VBT_OK) return fail("vbt_host_alloc");
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(frames, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, argv[2]); return 1; }
  fclose(f);

  vbt_pipeline_params prm;
  vbt_pipeline_default_params(&prm);          /* OCSort(max_age=30, asso_func="diou", iou_threshold=0.1), threshold 0.5 (track.py:22,157,174) */
  const int kept = T / stride;
  prm.n_slots = kept < 64 ? (kept < 1 ? 1 : kept) : 64;   /* consecutive kept frames of the clip per detector batch */
  prm.n_clips = 1;
  prm.rows_cap = 25 * (kept > 0 ? kept : 1) + 75;
  vbt_pipeline* p = NULL;
  if (vbt_pipeline_create(argv[1], &prm, &fps, &p) != VBT_OK) return fail("vbt_pipeline_create");

  int shape[4];
  if (vbt_model_input_shape(vbt_pipeline_model(p, 0), shape) != VBT_OK) return fail("vbt_model_input_shape");
  const int native = H == shape[1] && W == shape[2];

  int64_t* id = (int64_t*)malloc(sizeof(int64_t) * (size_t)prm.rows_cap);
  double* cols = (double*)malloc(sizeof(double) * 7 * (size_t)prm.rows_cap);
  int n = 0;
  if (vbt_track_clip(p, (const uint8_t*)frames, /*frames_on_device=*/0, T, native ? 0 : H, native ? 0 : W, /*swap_rb=*/0, stride, id, cols,
                     prm.rows_cap, &n) != VBT_OK)
    return fail("vbt_track_clip");

  printf("id,time,x,y,dx,dy,norm_plate_height,norm_plate_width\n");
  for (int i = 0; i < n; i++) {
    const double* c = cols + 7 * (size_t)i;
    printf("%lld,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n", (long long)id[i], c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
  }
  free(id);
  free(cols);
  vbt_pipeline_destroy(p);
  vbt_host_free(frames);
  return 0;
}
