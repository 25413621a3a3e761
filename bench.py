#!/usr/bin/env python3
"""Headline benchmark: frames/sec end-to-end (detect + NMS + track), EfficientDet-Lite0 320x320.

One STEP = one pass of the hot path over one batch: frame t of each of `--clips` (default 64)
synthetic clips per GPU -> int8 EfficientDet-Lite0 -> decode + NMS -> one OC-SORT step per clip,
all enqueued on one HIP stream with the frames already resident in HBM.  After the K timed steps
the clips are closed inside the timed region too: export-id selection + preprocessing +
VelocityTracker on the device, then (N > 1) one RCCL all-gather of the per-clip result records.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; ranks own disjoint clips, no
   data-path collective -> "weak" scaling; the only exchange is the final result gather.)

Prints ONE JSON line on rank 0 (see the keys at the bottom).  `roofline` is for the dominant kernel
family, timed with HIP events on the launch stream in a separate pass of the same process;
`cpu_baseline` times the CPU oracle (a port: the reference's TFLite path cannot run here) on a
bounded sample of the same workload on rank 0 at N = 1.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")   # before torch initialises HIP (see vbt_amd/__init__.py)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
# The kernel plan tuned on an MI355X at B = 64 is pinned so that every run (and the committed rocprofv3 / PMC
# summaries under profiles/) executes the same kernels; a different batch size or missing file re-tunes.
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
COUNTERS_JSON = os.path.join(ROOT, "profiles", "r02_counters.json")
# VBT_BENCH_MODEL: rehearsal knob (e.g. a Lite2 container for BASELINE config 4); the contract line is always Lite0
MODEL = os.environ.get("VBT_BENCH_MODEL", os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm"))


def make_frames(clip_seeds, t0, n_steps, size=320):
    """[n_steps, n_clips, S, S, 3] uint8: frame t0+i of every clip."""
    from vbt_amd import synth
    bgs = [synth.background(s, size) for s in clip_seeds]
    out = np.empty((n_steps, len(clip_seeds), size, size, 3), np.uint8)
    for i in range(n_steps):
        for c, bg in enumerate(bgs):
            out[i, c] = synth.render(bg, t0 + i)
    return out


def cpu_leg(n_frames, threads):
    """Oracle (CPU port) on a bounded sample: detector for n_frames frames (OpenMP over frames),
    then OC-SORT + rep analysis in numpy/python over the detections, as 8 clips."""
    from oracle import detector_ref, ocsort_np, velocity
    n_clips = 8
    per = max(n_frames // n_clips, 1)
    frames = make_frames(list(range(n_clips)), 0, per)               # [per, 8, ...]
    flat = np.ascontiguousarray(frames.reshape(-1, *frames.shape[2:]))
    t0 = time.perf_counter()
    boxes, scores, classes, counts = detector_ref.run_batch(MODEL, flat, threads=threads)
    t_det = time.perf_counter() - t0
    t1 = time.perf_counter()
    boxes = boxes.reshape(per, n_clips, 25, 4)
    scores = scores.reshape(per, n_clips, 25)
    counts = counts.reshape(per, n_clips)
    for c in range(n_clips):
        dets, times = [], []
        for f in range(per):
            d = [[boxes[f, c, i, 1], boxes[f, c, i, 0], boxes[f, c, i, 3], boxes[f, c, i, 2], scores[f, c, i], 0.0]
                 for i in range(counts[f, c]) if scores[f, c, i] >= 0.5]
            dets.append(np.asarray(d, np.float64).reshape(-1, 6))
            times.append((f + 1) / 60.0)
        rows = ocsort_np.track_boxes(dets, times)
        if rows["id"]:
            ids = np.asarray(rows["id"])
            m = ids == np.bincount(ids).argmax()
            velocity.analyze_track(*[np.asarray(rows[k])[m].tolist() for k in
                                     ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width")])
    t_trk = time.perf_counter() - t1
    n = per * n_clips
    return {"value": n / (t_det + t_trk), "unit": "frames/s", "cores": threads, "detector_frames_per_s": n / t_det,
            "tracker_clip_frames_per_s": n / t_trk,
            "sample": f"{n} synthetic 320x320 frames ({n_clips} clips x {per}), oracle/detector.c with {threads} OpenMP threads "
                      f"({t_det:.2f} s) + oracle OC-SORT/VelocityTracker in numpy, 1 thread ({t_trk:.2f} s)"}


def cpu_baseline(n_frames):
    """Two legs: 4 threads - the reference's `--threads` default (track.py:72) - and every core this
    process may use.  `value` is the all-cores leg.  kind "port": the reference's TFLite CPU path cannot run here."""
    avail = usable_cores()
    allc = max(1, min(avail, 32))         # beyond ~32 threads the frame-parallel port stops scaling (memory-bound scalar loops)
    leg4 = cpu_leg(max(n_frames // 4, 64), threads=min(4, avail))
    legn = cpu_leg(n_frames, threads=allc) if allc > 4 else leg4
    return {"value": legn["value"], "unit": "frames/s", "cores": legn["cores"], "kind": "port", "sample": legn["sample"],
            "legs": {"threads_4": leg4, "all_cores": legn}, "host_cpu": _cpu_name(), "host_cores_visible": os.cpu_count(),
            "host_cores_usable": avail}


def usable_cores():
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:                      # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="frames per clip inside the timed region (the reference clips hold 700-3300 frames)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU = detector batch")
    ap.add_argument("--unique-steps", type=int, default=64, help="distinct frame sets kept in HBM and cycled")
    ap.add_argument("--cpu-frames", type=int, default=2048, help="frames of the all-cores CPU baseline leg (the 4-thread leg takes a quarter); about 25 s of CPU work in all (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the H2D-inclusive pass and the detect-only / track-only splits")
    ap.add_argument("--settle-steps", type=int, default=200,
                    help="detector-only steps run before the warm-up so that the GPU is at steady clocks when the contract's W warm-up steps "
                         "start: the first GPU process on an idle MI355X measured 71.6 k instead of 89-93 k frames/s on the 20-step run "
                         "(0 = off; reported in the JSON line as settle_steps)")
    ap.add_argument("--seed-offset", type=int, default=0, help="rehearsal: run this rank on the clips another rank would own")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU, RCCL
        # rendezvous on 127.0.0.1) BEFORE this process makes any GPU call, relay their output (rank 0 prints the JSON
        # line) and exit with their return code.  The parent never touches the GPU and is never replaced by exec.
        return self_launch(args.gpus)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal knobs (not used by the driver): VBT_BENCH_SAME_DEVICE=1 puts every rank on GPU 0 and
    # VBT_BENCH_BACKEND=gloo gathers through host memory, so the N > 1 code path can run on a 1-GPU box.
    if os.environ.get("VBT_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("VBT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where collective buffers live

    from vbt_amd import _lib
    from vbt_amd.track import Pipeline
    n, K, W = args.clips, args.steps, args.warmup
    U = max(1, min(args.unique_steps, K + W))
    seeds = [(rank + args.seed_offset) * n + c for c in range(n)]   # ranks own disjoint clips
    from vbt_amd.container import Container
    size = int(Container(MODEL).header["image_size"])
    frames_np = make_frames(seeds, 0, U, size)
    frames = torch.from_numpy(frames_np).to(dev)                      # resident in HBM before timing
    XK = 200                                                          # steps of the extra (untimed-region) passes
    pipe = Pipeline(MODEL, n, max_frames=max(K + W, XK) + 8, fps=60.0, detection_treshold=0.5, device=local_rank, rows_per_frame=8)
    stream = torch.cuda.current_stream().cuda_stream
    fbytes = frames[0].numel()
    PH = 32                                                           # phases kept in the fixed-size result record
    trace = os.environ.get("VBT_BENCH_TRACE") == "1"

    def run_steps(count, start, **kw):
        for i in range(count):
            pipe.step(frames.data_ptr() + ((start + i) % U) * fbytes, stream, **kw)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.settle_steps > 0:   # clock settle, not part of the contract's warm-up: no tracker state is touched
        run_steps(args.settle_steps, 0, track=False)
        torch.cuda.synchronize()
        pipe.reset()
    run_steps(W, 0)
    fence()
    t0 = time.perf_counter()
    run_steps(K, W)
    t_enq = time.perf_counter()
    # clip close inside the timed region: pipeline drain, export-id selection + rep analysis on the device, ONE packed D2H
    best, rows_n, nph, ovf, ph = pipe.close(cap=PH)
    t_close = time.perf_counter()
    # result record per clip: [best_id, n_rows, n_phases, PH x (t0,t1,y0,y1,rom,type)]
    rec = np.zeros((n, 3 + PH * 6), np.float64)
    nrows = int(rows_n.sum())
    rec[:, 0], rec[:, 1], rec[:, 2] = best, rows_n, nph
    rec[:, 3:] = ph.reshape(n, -1)
    if dist is not None:                                             # the one exchange of the path: RCCL all-gather
        rec_all = gather_records(dist, rec, world, cdev)
    else:
        rec_all = rec[None]
    fence()
    dt = time.perf_counter() - t0
    if trace:
        print(f"[trace] rank {rank}: enqueue {1e3 * (t_enq - t0):.2f} ms, close {1e3 * (t_close - t_enq):.2f} ms, gather+fence "
              f"{1e3 * (t0 + dt - t_close):.2f} ms", file=sys.stderr)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    overflow = int((ovf != 0).sum())

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        extras = extra_measurements(torch, pipe, frames, frames_np, n, XK, U, fbytes, stream, PH)
    roofline = None
    if rank == 0 and not args.no_roofline:
        roofline = roofline_block(pipe, frames, n, stream)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        cpu = cpu_baseline(args.cpu_frames)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        total_frames = K * n * world
        out = {
            "metric": "frames/sec end-to-end (detect+NMS+track), EfficientDet-Lite0 320x320",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic",
            "config": {"workload": "EfficientDet-Lite0 320x320 full-integer (int8 act/weights, int32 acc; arithmetic of tflite-runtime "
                                   "2.14's XNNPACK kernels), synthetic clips batched frame-wise, decode+NMS+OC-SORT on device every step, "
                                   "export-id selection + VelocityTracker on device at clip end (inside the timed region); frames resident in HBM",
                       "clips_per_gpu": n, "batch": n, "frames_per_clip": K, "model_file": os.path.basename(MODEL),
                       "weights": "seeded synthetic (PCG64), post-training int8 quantised", "parallelism": f"clip-sharded x{world}",
                       "pipeline_depth": pipe.depth},
            "rows_emitted_rank0": int(nrows), "tracker_overflow_rank0": int(overflow),
            "clips_with_result": int((rec_all[..., 1] > 0).sum()),
            "timed_region_ms": {"enqueue": 1e3 * (t_enq - t0), "clip_close": 1e3 * (t_close - t_enq), "total": 1e3 * dt},
            "settle_steps": int(args.settle_steps),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        out.update(extras)
        print(json.dumps(out))
    return 0


def gather_records(dist, rec, world, cdev):
    """The path's one exchange: every rank's fixed-size per-clip result records, one all-gather (RCCL on GPUs)."""
    import torch
    mine = torch.from_numpy(rec).to(cdev)
    allrec = torch.empty((world * mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=cdev)   # concatenated layout
    dist.all_gather_into_tensor(allrec, mine)
    return allrec.cpu().numpy().reshape(world, *rec.shape)


def extra_measurements(torch, pipe, frames, frames_np, n, K, U, fbytes, stream, PH):
    """Rank 0, N = 1, outside the contract's timed region: the SURVEY 8d metric (frames in pinned host memory -> rows on the
    host, H2D and D2H included) and the detect-only / track-only splits, each over the same K steps."""
    out = {}
    Kx = K

    def reset():
        torch.cuda.synchronize()
        pipe.reset()

    # ---- H2D-inclusive: uint8 frames in pinned host memory; DataFrame rows of every clip back on the host ----
    Uh = min(U, 16)
    host = torch.from_numpy(frames_np[:Uh]).pin_memory()                       # [Uh, n, S, S, 3]
    rows_host = torch.empty(n * pipe.tracker.rows_cap * 64, dtype=torch.uint8).pin_memory()
    reset()
    for i in range(2 * Uh):                     # every pinned slice once (first DMA from a pinned page is slow), twice for the staging ring
        pipe.step(host[i % Uh], stream)
    torch.cuda.synchronize()
    reset()
    t0 = time.perf_counter()
    for i in range(Kx):
        pipe.step(host[i % Uh], stream)
    best, rows_n, nph, ovf, ph = pipe.close(cap=PH)
    counts, rows = pipe.rows_all(out=rows_host)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(counts.sum()) == int(rows_n.sum())
    out["value_h2d_inclusive"] = Kx * n / dt
    out["h2d_inclusive"] = {"frames_per_s": Kx * n / dt, "ms_per_step": dt / Kx * 1e3, "steps": Kx,
                            "h2d_bytes_per_step": int(host[0].numel()), "rows_d2h_bytes": int(counts.max()) * 64 * n,
                            "note": "uint8 frames in pinned host memory -> hipMemcpyAsync on the slot's stream (overlapped by the depth-3 "
                                    "pipeline) -> detect+NMS+track -> clip close -> all DataFrame rows copied to pinned host memory"}
    # ---- splits ----
    reset()
    for i in range(3):
        pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Kx):
        pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=False)
    torch.cuda.synchronize()
    det_dt = time.perf_counter() - t0
    reset()
    pipe.step(frames.data_ptr(), stream)                                       # one real step: detections in slot 0's buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.tracker_only_steps(Kx)
    torch.cuda.synchronize()
    trk_dt = time.perf_counter() - t0
    out["splits"] = {"detect_only": {"frames_per_s": Kx * n / det_dt, "ms_per_step": det_dt / Kx * 1e3,
                                     "note": f"detector + decode + NMS, {pipe.depth} forwards in flight, no tracker"},
                     "track_only": {"clip_frames_per_s": Kx * n / trk_dt, "us_per_step": trk_dt / Kx * 1e6,
                                    "note": "OC-SORT step of all clips on one frame's detections, repeated"}}
    reset()
    return out


def roofline_block(pipe, frames, n, stream):
    """Dominant kernel family by HIP-event time on the launch stream (one forward in flight)."""
    from vbt_amd import _lib
    L = _lib.lib()
    stats = (_lib.KernelStat * 16)()
    cnt = ctypes.c_int()
    _lib.check(L.vbt_model_kernel_stats(pipe.interpreter.handle, n, stats, 16, ctypes.byref(cnt)))
    ms = (ctypes.c_double * 16)()
    _lib.check(L.vbt_model_profile(pipe.interpreter.handle, frames.data_ptr(), n, 10, stream, ms, 16))
    fam = max(range(cnt.value), key=lambda i: ms[i])
    s = stats[fam]
    name = s.name.decode()
    per_launch_s = ms[fam] * 1e-3 / s.launches
    achieved = (s.algorithmic_bytes / s.launches) / per_launch_s
    traffic = counters = None   # from the separate rocprofv3 --pmc passes of this same plan (profiles/, tools/make_profile_summary.py)
    try:
        tj = json.load(open(COUNTERS_JSON))
        fj = tj["families"].get(name)
        if tj.get("batch") == n and fj and fj["launches"] == s.launches:
            traffic = fj["hbm_bytes_per_launch"]
        counters = tj.get("derived")
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "hbm", "kernel": name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "traffic": traffic, "launches_per_step": s.launches,
            "avg_launch_us": per_launch_s * 1e6, "algorithmic_bytes_per_launch": s.algorithmic_bytes / s.launches,
            "hbm_traffic_GBps": (traffic / per_launch_s / 1e9) if traffic else None,
            "counters": counters,
            "families_ms_per_step": {stats[i].name.decode(): round(ms[i], 4) for i in range(cnt.value)},
            "whole_net_algorithmic_GBps": sum(stats[i].algorithmic_bytes for i in range(cnt.value)) /
            (sum(ms[i] for i in range(cnt.value)) * 1e-3) / 1e9}


if __name__ == "__main__":
    sys.exit(main() or 0)
