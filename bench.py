#!/usr/bin/env python3
"""Headline benchmark: frames/sec end-to-end (detect + NMS + track), EfficientDet-Lite0 320x320.

One STEP = one pass of the hot path over one batch: frame t of each of `--clips` (default 64)
synthetic clips per GPU -> int8 EfficientDet-Lite0 -> decode + NMS -> one OC-SORT step per clip,
all enqueued on HIP streams with the frames already resident in HBM.  After the K timed steps
the clips are closed inside the timed region too: export-id selection + preprocessing +
VelocityTracker on the device, then (N > 1) one RCCL all-gather of the per-clip result records.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; ranks own disjoint clips, no
   data-path collective -> "weak" scaling; the only exchange is the final result gather.)

Prints ONE JSON line on rank 0.  Which number is which:
  value                 the contract and nothing else: W warm-up steps, then exactly K steps + clip close between two fences,
                        frames resident in HBM (what the task's bench contract defines as `value`).  No other GPU work precedes
                        the W warm-up steps in this process.
  value_settled         (N = 1) the same W + K run repeated after `--settle-steps` detector-only steps: the GPU's clocks have
                        settled; informational.
  value_h2d_inclusive   SURVEY.md 8d's metric, measured with the SAME W and K in the same process: uint8 frames in pinned
                        host memory -> H2D -> detect + NMS + track -> clip close -> every DataFrame row back in pinned host
                        memory, all inside its timed region.
  configs               N = 1: BASELINE.json's other configurations (batch 1, batch 8, one clip time-batched, the 34-clip
                        corpus on one GPU, Lite2 448x448), each with its SURVEY 8d roofline fraction.
                        N > 1: BASELINE config 5 as written - the 34-clip corpus LPT-sharded over the ranks, one all-gather of
                        the result records (corpus_sharded) - and SURVEY 8e's frame-major mode of ONE long clip: every rank
                        detects a contiguous frame chunk, one all-gather of the 504-byte per-frame detection records, rank 0
                        tracks (one_clip_frame_major).
  roofline              dominant kernel family, timed as ONE HIP-event bracket around all of its launches (the way
                        rocprofv3 --kernel-trace sees them; profiles/), with the PMC traffic of the same plan beside it.
  cpu_baseline          the CPU oracle (a port: the reference's TFLite path cannot run here) on a bounded sample.
"""
import argparse
import ctypes
import gc
import json
import os
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")   # before torch initialises HIP (see vbt_amd/__init__.py)
# a pipeline whose busy streams share a hardware queue loses a third of its throughput: here that is an error, not a warning
# (vbt_amd/csrc/pipeline.hip: place_streams; the contract run falls back to the note on stderr and says so in `stream_placement`)
os.environ.setdefault("VBT_STRICT_PLACEMENT", "1")

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_I8_PEAK = 5.0e15      # op/s dense: int8 MFMA = 2x the bf16 rate per clock (same guide, matrix-core table: ~2.5 PF bf16 dense)
# Kernel plans tuned on an MI355X are pinned (profiles/plan_<model>.b<batch>.f0) so that every run and the committed
# rocprofv3 / PMC summaries under profiles/ execute the same kernels; a missing file re-tunes on the spot.
PLAN_LITE0 = os.path.join(ROOT, "profiles", "plan_lite0")
PLAN_LITE2 = os.path.join(ROOT, "profiles", "plan_lite2")
os.environ.setdefault("VBT_PLAN_FILE", PLAN_LITE0)
COUNTERS = [os.path.join(ROOT, "profiles", f) for f in ("r05_counters.json", "r04_counters.json", "r03_counters.json")]
# VBT_BENCH_MODEL: rehearsal knob; the contract line is always Lite0
MODEL = os.environ.get("VBT_BENCH_MODEL", os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm"))
MODEL_LITE2 = os.path.join(ROOT, "models", "efficientdet_lite2_synth.vbtm")
CORPUS_META = os.path.join(ROOT, "tests", "golden", "corpus_meta.json")   # real frame counts / fps of the reference's 34 clips
# algorithmic bytes per frame = activation elements (each op reads its inputs and writes its output once) +
# weight elements / batch, at the 1 byte per element of the full-integer graph
ALG_ELEMS = {0: (36.84e6, 3.27e6), 2: (115.56e6, 5.44e6)}
STEADY_STEPS = 1000        # length of the steady-state repeat of the host-fed pass when the contract's K is shorter


def alg_bytes_per_frame(arch, batch):
    a, w = ALG_ELEMS[arch]
    return a + w / batch


def roofline_frac_8d(frames_per_s, arch, batch):
    return frames_per_s * alg_bytes_per_frame(arch, batch) / HBM_PEAK


def make_frames(clip_seeds, t0, n_steps, size=320):
    """[n_steps, n_clips, S, S, 3] uint8: frame t0+i of every clip."""
    from vbt_amd import synth
    bgs = [synth.background(s, size) for s in clip_seeds]
    out = np.empty((n_steps, len(clip_seeds), size, size, 3), np.uint8)
    for i in range(n_steps):
        for c, bg in enumerate(bgs):
            out[i, c] = synth.render(bg, t0 + i)
    return out


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------------------------
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


# ----------------------------------------------------------------------------------------------------------------
# process plumbing
# ----------------------------------------------------------------------------------------------------------------
def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes (one per GPU, RCCL rendezvous on 127.0.0.1),
    started BEFORE this process makes any GPU call.  A rank that dies takes the others down (torch.distributed.run) and its
    exit code comes back; a rendezvous that never completes is bounded by VBT_BENCH_TIMEOUT_S in every rank."""
    import socket
    with socket.socket() as sk:                      # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def cold_start_child(args):
    """The contract run in a fresh process, before this one has touched the GPU, without the clock-settle phase."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--clips", str(args.clips),
           "--settle-steps", "0", "--contract-only"]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        j = json.loads(line)
        return {"value": j["value"], "ms_per_step": j["ms_per_step"], "settle_steps": 0,
                "note": "fresh process started before the parent touched the GPU; same --steps / --warmup, no clock-settle phase"}
    except Exception as e:   # the headline must not die with the diagnostic
        return {"value": None, "error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="frames per clip inside the timed region (the reference clips hold 700-3300 frames)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU = detector batch")
    ap.add_argument("--unique-steps", type=int, default=64, help="distinct frame sets kept in HBM and cycled")
    ap.add_argument("--cpu-frames", type=int, default=2048, help="frames of the all-cores CPU baseline leg (the 4-thread leg takes a quarter); about 25 s of CPU work in all (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the H2D-inclusive pass, the splits and the other configurations")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE's other configurations (b1, b8, clip1, corpus, Lite2)")
    ap.add_argument("--cold-start", action="store_true", help="also run the contract once more in a fresh process started before this one touches the GPU")
    ap.add_argument("--contract-only", action="store_true", help="the timed region and nothing else (what the cold-start child runs)")
    ap.add_argument("--settle-steps", type=int, default=200,
                    help="N = 1: detector-only steps run before the REPEAT of the contract run that is reported as value_settled (0 = no repeat); "
                         "`value` itself is never preceded by them")
    ap.add_argument("--seed-offset", type=int, default=0, help="rehearsal: run this rank on the clips another rank would own")
    args = ap.parse_args()
    if args.contract_only:
        args.no_extras = args.no_roofline = True
        args.cold_start = False
        args.cpu_frames = 0
        args.settle_steps = 0

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cold = None
    if world == 1 and args.cold_start:
        cold = cold_start_child(args)                       # before this process initialises HIP
    import torch
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
    rccl_ranks = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        tmo = datetime.timedelta(seconds=int(os.environ.get("VBT_BENCH_TIMEOUT_S", "180")))   # a missing rank fails the job instead of hanging it
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where collective buffers live
    if dist is not None and os.environ.get("VBT_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)   # rehearsal knob (tests): a rank that dies must fail the whole job, not hang it
    if dist is not None:
        ones = torch.ones(1, dtype=torch.float64, device=cdev)   # the collective really spans `world` ranks
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        if rccl_ranks != dist.get_world_size():
            raise SystemExit(f"all_reduce of ones gave {rccl_ranks}, world size is {dist.get_world_size()}")

    from vbt_amd.track import Pipeline
    n, K, W = args.clips, args.steps, args.warmup
    U = max(1, min(args.unique_steps, K + W))
    seeds = [(rank + args.seed_offset) * n + c for c in range(n)]   # ranks own disjoint clips
    from vbt_amd.container import Container
    size = int(Container(MODEL).header["image_size"])
    frames_np = make_frames(seeds, 0, U, size)
    frames = torch.from_numpy(frames_np).to(dev)                      # resident in HBM before timing
    placement = "every busy stream on its own hardware queue"
    try:
        pipe = Pipeline(MODEL, n, max_frames=max(K, STEADY_STEPS) + W + 8, fps=60.0, detection_treshold=0.5, device=local_rank, rows_per_frame=8)
    except Exception as e:
        if type(e).__name__ != "StreamPlacementError":
            raise
        os.environ["VBT_STRICT_PLACEMENT"] = "0"              # measure anyway, and say so
        placement = "FAILED: busy streams share a hardware queue (profiler attached, or GPU_MAX_HW_QUEUES too small)"
        pipe = Pipeline(MODEL, n, max_frames=max(K, STEADY_STEPS) + W + 8, fps=60.0, detection_treshold=0.5, device=local_rank, rows_per_frame=8)
    stream = torch.cuda.current_stream().cuda_stream
    fbytes = frames[0].numel()
    PH = 128                                                          # phases kept in the fixed-size result record (the synthetic detector's noisy tracks: up to ~60 short phases per 1000-frame clip)
    trace = os.environ.get("VBT_BENCH_TRACE") == "1"

    def run_steps(count, start, **kw):
        for i in range(count):
            pipe.step(frames.data_ptr() + ((start + i) % U) * fbytes, stream, **kw)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def contract_run():
        """W warm-up steps, fence, exactly K steps + clip close (+ the N > 1 gather), fence."""
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
        rec[:, 0], rec[:, 1], rec[:, 2] = best, rows_n, nph
        rec[:, 3:] = ph.reshape(n, -1)
        if dist is not None:                                             # the one exchange of the path: RCCL all-gather
            rec_all = gather_records(dist, rec, world, cdev)
        else:
            rec_all = rec[None]
        fence()
        return time.perf_counter() - t0, t0, t_enq, t_close, rows_n, ovf, rec_all

    dt_local, t0, t_enq, t_close, rows_n, ovf, rec_all = contract_run()      # `value`: nothing ran on the GPU before its warm-up steps
    nrows = int(rows_n.sum())
    dt = dt_local
    if trace:
        print(f"[trace] rank {rank}: enqueue {1e3 * (t_enq - t0):.2f} ms, close {1e3 * (t_close - t_enq):.2f} ms, gather+fence "
              f"{1e3 * (t0 + dt - t_close):.2f} ms", file=sys.stderr)
    per_rank = None
    if dist is not None:
        tall = torch.zeros(world, dtype=torch.float64, device=cdev)
        tall[rank] = dt_local
        dist.all_reduce(tall)                                         # every rank's own time; the job's time is the slowest
        per_rank = [K * n / float(t) for t in tall.tolist()]
        dt = float(tall.max().item())
    overflow = int((ovf != 0).sum())

    extras = {}
    if world == 1 and args.settle_steps > 0 and not args.contract_only:
        # the same run once more with the clocks settled (informational; ADVICE r03: the headline is the run above)
        pipe.reset()
        run_steps(args.settle_steps, 0, track=False)
        torch.cuda.synchronize()
        pipe.reset()
        dts = contract_run()[0]
        extras["value_settled"] = K * n / dts
        extras["settled"] = {"frames_per_s": K * n / dts, "ms_per_step": dts / K * 1e3, "settle_steps": int(args.settle_steps),
                             "note": "the contract run repeated in the same process after settle_steps detector-only steps"}
        pipe.reset()
    if rank == 0 and world == 1 and not args.no_extras:
        extras.update(extra_measurements(torch, pipe, frames, frames_np, n, K, W, U, fbytes, stream, PH))
    roofline = None
    if rank == 0 and not args.no_roofline:
        roofline = roofline_block(pipe, n, stream)
    del pipe
    gc.collect()
    configs = None
    if rank == 0 and world == 1 and not args.no_extras and not args.no_configs:
        configs = other_configs(torch, dev)
    if world > 1 and not args.no_configs:                    # every rank takes part: these are the sharded configurations
        configs = multi_gpu_configs(torch, dist, dev, cdev, rank, world, local_rank)
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
                       "pipeline_depth": int(os.environ.get("VBT_PIPELINE_DEPTH", "3"))},
            "metric_definition": "resident",
            "value_is": "the contract run and nothing before it: W warm-up steps, K timed steps + clip close, frames resident in HBM; value_settled = "
                        "the same run repeated after a clock-settle phase; value_h2d_inclusive = SURVEY 8d's host-to-host metric, same W and K",
            "roofline_frac_8d": roofline_frac_8d(total_frames / dt / world, 0, n),
            "rows_emitted_rank0": int(nrows), "tracker_overflow_rank0": int(overflow),
            "clips_with_result": int((rec_all[..., 1] > 0).sum()),
            "timed_region_ms": {"enqueue": 1e3 * (t_enq - t0), "clip_close": 1e3 * (t_close - t_enq), "total": 1e3 * dt},
            "cold_start": cold, "stream_placement": placement,
            "rccl_ranks": rccl_ranks, "per_rank_frames_per_s": per_rank,
            "roofline": roofline, "cpu_baseline": cpu, "configs": configs,
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


# ----------------------------------------------------------------------------------------------------------------
# measurements beside the contract's timed region (rank 0, N = 1)
# ----------------------------------------------------------------------------------------------------------------
def extra_measurements(torch, pipe, frames, frames_np, n, K, W, U, fbytes, stream, PH):
    """SURVEY 8d's metric (frames in pinned host memory -> rows on the host, H2D and D2H inside the timed region) with the
    contract's own W and K, and the detect-only / track-only splits."""
    out = {}
    Kx = min(K, pipe.tracker.rows_cap // 8 - W - 8)

    def reset():
        torch.cuda.synchronize()
        pipe.reset()

    # ---- H2D-inclusive: uint8 frames in pinned host memory; DataFrame rows of every clip back on the host ----
    Uh = min(U, 16)
    # (order matters on this stack: with the 33 MB row buffer pinned AFTER the frames, every H2D enqueue of a frame batch blocked the host
    #  for the length of the copy and the run lost 12 % - 82-87 k against 96-99 k frames/s at K = 20, tools/h2d_k20_probe.py,
    #  profiles/r04_h2d_pinned_order.md; pinned first, the copies are asynchronous as they should be)
    rows_host = torch.empty(n * pipe.tracker.rows_cap * 64, dtype=torch.uint8).pin_memory()
    host = torch.from_numpy(frames_np[:Uh]).pin_memory()                       # [Uh, n, S, S, 3]
    reset()
    for i in range(2 * Uh):                     # every pinned slice once (first DMA from a pinned page is slow), twice for the staging ring
        pipe.step(host[i % Uh], stream, track=False)
    pipe.step(host[0], stream)                  # ... and the row buffer's pages once: one tracked step, clip close, rows into the pinned buffer
    pipe.close(cap=PH)
    pipe.rows_all(out=rows_host)
    def host_fed(steps):
        reset()
        for i in range(W):                          # the contract's warm-up
            pipe.step(host[i % Uh], stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.step(host[(W + i) % Uh], stream)
        host_fed.enqueue_ms = (time.perf_counter() - t0) * 1e3      # host time spent enqueueing (a copy that blocks the host shows here)
        best, rows_n, nph, ovf, ph = pipe.close(cap=PH)
        host_fed.close_ms = (time.perf_counter() - t0) * 1e3 - host_fed.enqueue_ms
        counts, rows = pipe.rows_all(out=rows_host)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert int(counts.sum()) == int(rows_n.sum())
        return dt, int(counts.max()) * 64 * n

    dt, d2h = host_fed(Kx)
    out["value_h2d_inclusive"] = Kx * n / dt
    out["h2d_inclusive"] = {"frames_per_s": Kx * n / dt, "ms_per_step": dt / Kx * 1e3, "steps": Kx, "warmup": W,
                            "roofline_frac_8d": roofline_frac_8d(Kx * n / dt, 0, n),
                            "h2d_bytes_per_step": int(host[0].numel()), "rows_d2h_bytes": d2h, "enqueue_ms": host_fed.enqueue_ms, "close_ms": host_fed.close_ms,
                            "note": "uint8 frames in pinned host memory -> hipMemcpyAsync on the copy stream (two steps ahead of the "
                                    "forwards) -> detect+NMS+track -> clip close -> all DataFrame rows copied to pinned host memory; the first "
                                    "forward cannot start before its own 19.7 MB copy (0.36 ms at 54 GB/s) has landed, which a short run pays in full"}
    if Kx < STEADY_STEPS:                           # the same pass over 1000 steps: the pipeline-fill cost amortised
        dts, _ = host_fed(STEADY_STEPS)
        out["h2d_inclusive"]["steady"] = {"frames_per_s": STEADY_STEPS * n / dts, "ms_per_step": dts / STEADY_STEPS * 1e3, "steps": STEADY_STEPS,
                                          "roofline_frac_8d": roofline_frac_8d(STEADY_STEPS * n / dts, 0, n)}
    # ---- splits ----
    Ks = min(Kx, 400)
    reset()
    for i in range(3):
        pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Ks):
        pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=False)
    torch.cuda.synchronize()
    det_dt = time.perf_counter() - t0
    reset()
    pipe.step(frames.data_ptr(), stream)                                       # one real step: detections in slot 0's buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.tracker_only_steps(Ks)
    torch.cuda.synchronize()
    trk_dt = time.perf_counter() - t0
    out["splits"] = {"detect_only": {"frames_per_s": Ks * n / det_dt, "ms_per_step": det_dt / Ks * 1e3,
                                     "note": f"detector + decode + NMS, {pipe.depth} forwards in flight, no tracker"},
                     "track_only": {"clip_frames_per_s": Ks * n / trk_dt, "us_per_step": trk_dt / Ks * 1e6,
                                    "note": "OC-SORT step of all clips on one frame's detections, repeated"}}
    reset()
    return out


def _timed(torch, body, reps=2):
    """Best of `reps` runs of body() (which must end with everything on the host), seconds."""
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        body()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def other_configs(torch, dev):
    """BASELINE.json configs 2, 4 and 5 and the small-batch points, each as whole-clip runs (clip close + rows on the host
    inside the timed region), frames resident in HBM."""
    from vbt_amd import shard, synth
    from vbt_amd.track import Pipeline
    out = {}
    stream = torch.cuda.current_stream().cuda_stream

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as e:                                   # a diagnostic configuration must not cost the headline
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        gc.collect()

    # ---- config 2 as literally written: one clip, one frame per step (batch 1), and batch 8 ----
    def small_batch(nb, T):
        U = 64
        fr = torch.from_numpy(make_frames(list(range(nb)), 0, U)).to(dev)
        fb = fr[0].numel()
        pipe = Pipeline(MODEL, nb, max_frames=T, fps=60.0)

        enq = []

        def body():
            pipe.reset()
            t0 = time.perf_counter()
            for t in range(T):
                pipe.step(fr.data_ptr() + (t % U) * fb, stream)
            enq.append(time.perf_counter() - t0)
            pipe.close(cap=512)
            pipe.rows_all()
        dt = _timed(torch, body)
        fps = nb * T / dt
        # enqueue_ms: host time inside the T step() calls of a run (they return as soon as the work is queued; the host blocks only on back-pressure
        # from the HIP runtime's queues).  library_us_per_step is the part of it spent inside vbt_pipeline_step (vbt_pipeline_info.step_host_ns,
        # last run), the rest is the Python / ctypes wrapper.  enqueue_us_per_step well below ms_per_step = the GPU, not the host, paces the run.
        inf = pipe.info()
        lib_us = inf.step_host_ns / max(inf.step_calls, 1) / 1e3
        return {"frames_per_s": fps, "ms_per_step": dt / T * 1e3, "batch": nb, "frames_per_clip": T, "roofline_frac_8d": roofline_frac_8d(fps, 0, nb),
                "enqueue_ms": min(enq) * 1e3, "enqueue_us_per_step": min(enq) / T * 1e6, "library_us_per_step": lib_us,
                "wrapper_us_per_step": enq[-1] / T * 1e6 - lib_us, "host_paces_the_run": bool(min(enq) > 0.9 * dt),
                "note": f"{nb} clip(s), one frame per clip per step (the forward is a replayed hipGraph), {pipe.depth} forwards in flight"}
    guarded("b1", lambda: small_batch(1, 2048))
    guarded("b8", lambda: small_batch(8, 1024))

    # ---- INTEGRATION.md option A: the reference's loop UNCHANGED (track.py:159-234) over the drop-in objects: one host frame at a time
    #      through Interpreter (run_odt: resize, H2D, batch-1 forward, D2H) and OCSort.update / tracker.trackers (one stream-ordered
    #      read-back per frame) ----
    def dropin():
        from vbt_amd.interpreter import Interpreter
        from vbt_amd.track import track
        T = 400
        bg = synth.background(0)
        frames = [synth.render(bg, t) for t in range(T)]
        it = Interpreter(MODEL, max_batch=1)
        res = {}

        def body():
            res["rows"] = len(track(frames, it, detection_treshold=0.5, fps=60.0)["id"])
        dt = _timed(torch, body)
        # where a frame's time goes (one more, instrumented pass): the blocking C call of the detector (H2D + batch-1 forward + packed D2H),
        # the Python around it inside run_odt (preprocess_image, the list of dicts), OCSort.update (C call + packed read-back), the loop's own Python
        import vbt_amd.track as vt
        from vbt_amd.ocsort import OCSort
        acc = {"c_detect": 0.0, "run_odt": 0.0, "update": 0.0}
        o_odt, o_upd, o_det = vt.run_odt, OCSort.update, Interpreter.detect

        def timed(key, fn):
            def w(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    acc[key] += time.perf_counter() - t0
            return w
        vt.run_odt, OCSort.update, Interpreter.detect = timed("run_odt", o_odt), timed("update", o_upd), timed("c_detect", o_det)
        try:
            t0 = time.perf_counter()
            body()
            tot = time.perf_counter() - t0
        finally:
            vt.run_odt, OCSort.update, Interpreter.detect = o_odt, o_upd, o_det
        split = {"detector_c_call_us": acc["c_detect"] / T * 1e6, "run_odt_python_us": (acc["run_odt"] - acc["c_detect"]) / T * 1e6,
                 "ocsort_update_us": acc["update"] / T * 1e6, "loop_python_us": (tot - acc["run_odt"] - acc["update"]) / T * 1e6}
        return {"frames_per_s": T / dt, "ms_per_frame": dt / T * 1e3, "frames": T, "rows": res["rows"], "host_split_us_per_frame": split,
                "enqueue_ms": None, "enqueue_note": "every call of this loop blocks until its result is on the host: there is no enqueue phase; host_split_us_per_frame is the breakdown",
                "note": "vbt_amd.track.track(): the per-frame loop of reference track.py:159-234 with Interpreter / run_odt / OCSort swapped in "
                        "(INTEGRATION.md option A); every frame starts in host memory and every call returns host results, as in the reference"}
    guarded("dropin_per_frame", dropin)

    # ---- config 2's clip on the time-batched path: ONE clip of 4096 frames, 64 consecutive frames per detector batch ----
    def clip1():
        T, F, U = 4096, 64, 256
        bg = synth.background(0)
        base = np.stack([synth.render(bg, t) for t in range(U)])
        fr = torch.from_numpy(np.concatenate([base, base[:F]])).to(dev)          # a cycle of U frames: every run of F is contiguous
        pipe = Pipeline(MODEL, F, max_frames=T, fps=60.0, tracker_clips=1)
        res = {}

        def body():
            pipe.reset()
            for f0 in range(0, T, F):
                s = f0 % U
                pipe.step_runs(fr[s:s + F], [(0, 0, F, f0 + 1)])
            b, r, p, o, _ = pipe.close(cap=512)
            pipe.rows_all()
            res.update(rows=int(r.sum()), phases=int(p.sum()))
        dt = _timed(torch, body)
        fps = T / dt
        return {"frames_per_s": fps, "ms_per_step": dt / (T // F) * 1e3, "batch": F, "clips": 1, "frames_per_clip": T, "roofline_frac_8d": roofline_frac_8d(fps, 0, F),
                "rows": res["rows"], "phases": res["phases"],
                "note": "one 4096-frame clip, 64 consecutive frames per detector batch, OC-SORT walks the run in frame order inside one launch "
                        "(vbt_tracker_update_from_detections_seq): bound by the sequential tracker walk, not by the detector"}
    guarded("clip1_time_batched", clip1)

    # ---- the tracker on the REFERENCE's load: clip 001's own boxes (dfs_ocsort: <= 3 plates per frame, 2 ids) fed as detector outputs to the
    #      time-batched walk, 64 frames per launch like clip1_time_batched; the synthetic detector keeps ~10 tracks alive, the reference 2-3 ----
    def clip1_reference():
        from vbt_amd import _lib
        from vbt_amd.ocsort import MultiClipTracker
        L = _lib.lib()
        a = np.load(os.path.join(ROOT, "tests", "golden", "dfs_ocsort_all.npz"))
        fps = float(json.load(open(CORPUS_META))["001"][1])
        g = {k: a[f"c001_{k}"] for k in ("time", "x", "y", "norm_plate_height", "norm_plate_width")}
        fno = np.rint(g["time"] * fps).astype(np.int64)                  # 1-based frame numbers (time = frame_count / fps, track.py:161,169)
        T = int(fno.max())
        boxes = np.zeros((T, 25, 4), np.float32)
        scores = np.zeros((T, 25), np.float32)
        counts = np.zeros(T, np.int32)
        for f, x, y, h, w in zip(fno, g["x"], g["y"], g["norm_plate_height"], g["norm_plate_width"]):
            i = counts[f - 1]
            boxes[f - 1, i] = (y - h / 2, x - w / 2, y + h / 2, x + w / 2)   # detector layout ymin,xmin,ymax,xmax (odt.py:64-66)
            scores[f - 1, i] = 0.9
            counts[f - 1] = i + 1
        db, ds, dc = (torch.from_numpy(v).to(dev) for v in (boxes, scores, counts))
        trk = MultiClipTracker(1, 8 * T + 75, max_age=30, asso_func="diou", iou_threshold=0.1)
        F = 64
        res = {}

        def body():
            trk.reset()
            for f0 in range(0, T, F):
                nf = min(F, T - f0)
                run = (_lib.Run * 1)(_lib.Run(0, f0, 1, nf, f0 + 1, 1, fps))
                _lib.check(L.vbt_tracker_update_from_detections_seq(trk.handle, db.data_ptr(), ds.data_ptr(), dc.data_ptr(), T, run, 1, 0.5, stream))
            trk.finish(0.45, stream=stream)
            best, rows_n, nph, ovf, ph = trk.summary(cap=64)
            res.update(rows=int(rows_n.sum()), phases=int(nph.sum()), best_id=int(best[0]))
        dt = _timed(torch, body, reps=3)
        stepped = int((counts > 0).sum())
        return {"us_per_stepped_frame": dt / stepped * 1e6, "clip_frames_per_s": T / dt, "frames": T, "stepped_frames": stepped, "rows": res["rows"], "phases": res["phases"],
                "export_id": res["best_id"], "detections_per_stepped_frame": float(counts.sum()) / stepped,
                "note": "OC-SORT walk alone on the reference's own load: clip 001's dfs_ocsort boxes (tests/golden/dfs_ocsort_all.npz) as detector outputs, "
                        "64 frames per tracker launch + clip close; compare with clip1_time_batched, whose synthetic detector keeps ~10 tracks alive"}
    guarded("clip1_reference_boxes", clip1_reference)

    # ---- SURVEY 8b's fused entry point: vbt_track_clip - ONE call per clip, frames in pinned host memory, rows back - at the network
    #      resolution and at the reference's source resolution (1920 x 1080: row-pair upload + on-device resize) ----
    def track_clip_c_abi():
        from vbt_amd import mem
        res = {}
        for name, T, H, W in (("net_320x320", 2048, 320, 320), ("src_1920x1080", 192, 1920, 1080)):
            host = mem.pinned_empty((T, H, W, 3))
            if (H, W) == (320, 320):
                bg = synth.background(0)
                for t in range(T):
                    host[t] = synth.render(bg, t)
            else:
                rng = np.random.default_rng(7)
                base = np.repeat(np.repeat(rng.integers(0, 256, (8, H // 8, W // 8, 3), dtype=np.uint8), 8, 1), 8, 2)
                for t in range(T):
                    host[t] = base[t % 8]
            pipe = Pipeline(MODEL, 64, max_frames=T, fps=60.0, tracker_clips=1, rows_per_frame=25)
            out = {}

            def body():
                out["rows"] = len(pipe.track_clip(host, src_hw=None if (H, W) == (320, 320) else (H, W))["id"])
            body()
            dt = _timed(torch, body)
            res[name] = {"frames_per_s": T / dt, "ms_per_frame": dt / T * 1e3, "frames": T, "rows": out["rows"]}
            del pipe, host
            gc.collect()
        return {"frames_per_s": res["net_320x320"]["frames_per_s"], **res,
                "note": "vbt_track_clip (include/vbt_hip.h): the whole loop of reference track.py:129-260 for one clip in one C call - pinned host frames -> "
                        "copy stream -> 64 consecutive frames per detector batch -> OC-SORT walk on the device -> rows on the host; the 1920 x 1080 clip uploads "
                        "only the rows the bilinear resize reads"}
    guarded("track_clip_c_abi", track_clip_c_abi)

    # ---- 64 clips x 4 consecutive frames per detector batch (B = 256): what time-batching adds on top of the clip batch ----
    def b64x4():
        n, F, T = 64, 4, 512
        U = 12 * F
        fr = torch.from_numpy(make_frames(list(range(n)), 0, U)).to(dev).transpose(0, 1).contiguous()        # [clip][U]
        pipe = Pipeline(MODEL, n * F, max_frames=T + 8, fps=60.0, tracker_clips=n, rows_per_frame=8)

        def body():
            pipe.reset()
            pipe.frame_count = 0
            for t in range(0, T, F):
                s = t % U
                pipe.step_seq(fr[:, s:s + F].contiguous())
            pipe.close(cap=512)
            pipe.rows_all()
        dt = _timed(torch, body)
        fps = n * T / dt
        return {"frames_per_s": fps, "ms_per_step": dt / (T // F) * 1e3, "batch": n * F, "clips": n, "frames_per_clip": T, "roofline_frac_8d": roofline_frac_8d(fps, 0, n * F),
                "note": "frames t..t+3 of 64 clips as one detector batch of 256 (Pipeline.step_seq), OC-SORT walks each clip's 4 frames inside one launch"}
    guarded("b64x4_time_batched", b64x4)

    # ---- config 5's per-GPU shape: the 34-clip corpus (real frame counts / fps) on ONE GPU, time-batched ----
    def corpus():
        meta = json.load(open(CORPUS_META))
        keys = sorted(meta)
        lengths = np.array([meta[k][0] for k in keys])
        fps_c = np.array([meta[k][1] for k in keys])
        U, slots = 8, 64
        base = np.stack([np.stack([synth.render(synth.background(int(k), 320), 11 * u) for u in range(U)]) for k in keys])
        fr = torch.from_numpy(np.concatenate([base, base], axis=1)).to(dev)      # [clip][2U]
        steps = shard.run_schedule(lengths, slots)
        pipe = Pipeline(MODEL, slots, max_frames=int(lengths.max()), fps=fps_c, tracker_clips=len(keys))
        res = {}

        def body():
            pipe.reset()
            for step in steps:
                pipe.step_runs([fr[c, (f0 - 1) % U:(f0 - 1) % U + nf] for c, _, nf, f0 in step], step)
            b, r, p, o, _ = pipe.close(cap=512)
            pipe.rows_all()
            res.update(rows=int(r.sum()), overflow=int((o != 0).sum()))
        dt = _timed(torch, body)
        total = int(lengths.sum())
        fps = total / dt
        return {"frames_per_s": fps, "ms_per_step": dt / len(steps) * 1e3, "batch": slots, "clips": len(keys), "frames": total, "steps": len(steps),
                "roofline_frac_8d": roofline_frac_8d(fps, 0, slots), "rows": res["rows"], "tracker_overflow": res["overflow"],
                "longest_run": max(nf for s in steps for _, _, nf, _ in s),
                "note": "34 synthetic clips with the reference corpus' frame counts (699...3243) and frame rates; every step deals the 64 "
                        "detector slots to the clips in proportion to the frames they have left (shard.run_schedule)"}
    guarded("corpus_1gpu", corpus)

    # ---- row N2: frames at SOURCE resolution (the reference's clips are ~1080x1920 portrait): BGR -> RGB +
    #      bilinear resize + truncating cast on the device (odt.py:10-19, track.py:171) in front of every forward ----
    def n2_source():
        nb, T, H, W = 64, 24, 1920, 1080
        rng = np.random.default_rng(5)
        src = torch.from_numpy(rng.integers(0, 256, (2, nb, H, W, 3), dtype=np.uint8))
        dev_src = src.to(dev)
        host_src = src.pin_memory()
        pipe = Pipeline(MODEL, nb, max_frames=T + 4, fps=30.0)
        res = {}
        for name, s_ in (("resident", dev_src), ("host_fed", host_src)):
            def body():
                pipe.reset()
                for t in range(T):
                    pipe.step(s_[t % 2], stream, src_hw=(H, W), swap_rb=True)
                pipe.close(cap=64)
                pipe.rows_all()
            body()                                             # first touch of the pinned pages / staging buffers
            dt = _timed(torch, body)
            res[name] = {"frames_per_s": nb * T / dt, "ms_per_step": dt / T * 1e3}
        bytes_step = nb * H * W * 3
        up0 = pipe.info().h2d_bytes
        pipe.reset()
        pipe.step(host_src[0], stream, src_hw=(H, W), swap_rb=True)
        pipe.close(cap=64)
        uploaded = int(pipe.info().h2d_bytes - up0)               # only the row pairs the bilinear resize reads travel (vbt_amd/csrc/pipeline.hip)
        return {"frames_per_s": res["host_fed"]["frames_per_s"], "resident_frames_per_s": res["resident"]["frames_per_s"],
                "ms_per_step": res["host_fed"]["ms_per_step"], "resident_ms_per_step": res["resident"]["ms_per_step"], "batch": nb,
                "source_hw": [H, W], "source_bytes_per_step": bytes_step, "h2d_bytes_per_step": uploaded,
                "h2d_GBps": uploaded / (res["host_fed"]["ms_per_step"] * 1e-3) / 1e9,
                "roofline_frac_8d": roofline_frac_8d(res["host_fed"]["frames_per_s"], 0, nb),
                "note": "64 uint8 1920x1080x3 BGR frames per step (398 MB in host memory): host-fed = pinned host memory -> H2D of the 2 x 320 source rows per frame "
                        "the bilinear resize of odt.py:15-16 reads (one strided copy, a third of the bytes) -> resize on the device -> detect + NMS + track; "
                        "resident = the same frames already in HBM"}
    guarded("n2_1080x1920", n2_source)

    # ---- config 4: EfficientDet-Lite2 448x448 + OC-SORT, 64 clips per step ----
    def lite2():
        nb, T, U = 64, 120, 4
        os.environ["VBT_PLAN_FILE"] = PLAN_LITE2
        try:
            fr = torch.from_numpy(make_frames(list(range(nb)), 0, U, 448)).to(dev)
            fb = fr[0].numel()
            pipe = Pipeline(MODEL_LITE2, nb, max_frames=T, fps=30.0)
        finally:
            os.environ["VBT_PLAN_FILE"] = PLAN_LITE0

        def body():
            pipe.reset()
            for t in range(T):
                pipe.step(fr.data_ptr() + (t % U) * fb, stream)
            pipe.close(cap=64)
            pipe.rows_all()
        dt = _timed(torch, body)
        fps = nb * T / dt
        return {"frames_per_s": fps, "ms_per_step": dt / T * 1e3, "batch": nb, "frames_per_clip": T, "roofline_frac_8d": roofline_frac_8d(fps, 2, nb),
                "model_file": os.path.basename(MODEL_LITE2), "plan": os.path.basename(PLAN_LITE2) + f".b{nb}.f0",
                "algorithmic_bytes_per_frame": alg_bytes_per_frame(2, nb)}
    if os.path.exists(MODEL_LITE2):
        guarded("lite2_448", lite2)
    return out


def multi_gpu_configs(torch, dist, dev, cdev, rank, world, local_rank):
    """N > 1 (every rank calls this): BASELINE config 5 as written and SURVEY 8e's frame-major mode, each one timed region between
    two barriers with its one collective inside; rank 0 returns the dict that goes into the line."""
    out = {}
    for name, fn in (("corpus_sharded", corpus_sharded), ("one_clip_frame_major", one_clip_frame_major)):
        try:
            res = fn(torch, dist, dev, cdev, rank, world, local_rank)
        except Exception as e:                                # a rank that fails here would hang the others at the next collective: leave
            print(f"[bench] rank {rank}: {name} failed: {type(e).__name__}: {e}", file=sys.stderr)
            raise
        if rank == 0:
            out[name] = res
        gc.collect()
    return out if rank == 0 else None


def corpus_sharded(torch, dist, dev, cdev, rank, world, local_rank, slots=64, U=8, PH=32):
    """BASELINE config 5 (reference track.py:85-126 runs the clips one after the other): the 34-clip corpus - a synthetic stand-in with
    the REAL per-clip frame counts and frame rates of dfs_ocsort (tests/golden/corpus_meta.json) - clip-sharded over the ranks by
    longest-processing-time packing and run TIME-BATCHED on every rank (shard.run_schedule deals the 64 detector slots of a step to
    the rank's clips in proportion to the frames they have left; OC-SORT walks a clip's run inside one launch); one all-gather of the
    fixed-size per-clip result records at the end."""
    from vbt_amd import shard, synth
    from vbt_amd.track import Pipeline
    clips = {k: (int(v[0]), float(v[1])) for k, v in json.load(open(CORPUS_META)).items()}
    shards = shard.shard_clips({k: v[0] for k, v in clips.items()}, world)
    mine = shards[rank]
    n = len(mine)
    lengths = np.array([clips[k][0] for k in mine])
    fps = np.array([clips[k][1] for k in mine])
    base = np.stack([np.stack([synth.render(synth.background(int(k), 320), 11 * u) for u in range(U)]) for k in mine])
    frames = torch.from_numpy(np.concatenate([base] * (2 + slots // U), axis=1)).to(dev)      # [clip][cycle]: any run of <= 64 frames is contiguous
    steps = shard.run_schedule(lengths, slots)
    pipe = Pipeline(MODEL, slots, max_frames=int(lengths.max()), fps=fps, detection_treshold=0.5, tracker_clips=n, device=local_rank)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for rep in range(2):                                             # the first pass warms plans, pinned buffers and clocks
        pipe.reset()
        fence()
        t0 = time.perf_counter()
        for step in steps:
            pipe.step_runs([frames[c, (f0 - 1) % U:(f0 - 1) % U + nf] for c, _, nf, f0 in step], step)
        best, rows, nph, ovf, ph = pipe.close(cap=512)                # (the synthetic detector's noisy tracks give up to ~200 short phases)
        ph = ph[:, :PH]                                               # the fixed-size record keeps the first PH
        rec = np.zeros((n, 4 + PH * 6), np.float64)
        rec[:, 0] = [int(k) for k in mine]
        rec[:, 1], rec[:, 2], rec[:, 3] = best, rows, nph
        rec[:, 4:] = ph.reshape(n, -1)
        allrec = shard.gather_records(torch.from_numpy(rec).to(cdev), dist, pad_to=max(len(sh) for sh in shards)) if dist is not None else rec
        fence()
        dt = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        tall = torch.zeros(world, dtype=torch.float64, device=cdev)
        tall[rank] = dt
        dist.all_reduce(tall)
        per_rank = [float(t) for t in tall.tolist()]
        dt = max(per_rank)
    total = sum(v[0] for v in clips.values())
    del pipe
    return {"frames_per_s": total / dt, "seconds": dt, "clips": int(len(allrec)), "frames": total, "n_gpus": world, "per_rank_seconds": per_rank,
            "rank0": {"clips": n, "frames": int(lengths.sum()), "steps": len(steps), "longest_run": max(nf for st in steps for _, _, nf, _ in st),
                      "rows": int(rows.sum()), "overflow": int((ovf != 0).sum())},
            "note": "34 synthetic clips with the reference corpus' frame counts (699...3243) and frame rates, LPT-sharded over the ranks, "
                    "time-batched per rank, one all-gather of the per-clip result records inside the timed region"}


def one_clip_frame_major(torch, dist, dev, cdev, rank, world, local_rank, T=4096, F=64, U=256):
    """SURVEY 8e's second mode / the north_star's "gather of per-frame boxes": ONE long clip (config 2's 4096 frames).  The detector has no
    state, so rank r detects the contiguous frame chunk r (64 frames per forward); ONE all-gather of the fixed-size per-frame records
    (25 x (4 box + 1 score) f32 + count = 504 bytes per frame); the tracker is sequential per clip, so rank 0 alone walks all frames
    (on the device, straight from the gathered records) and closes the clip.  Timed: barrier .. detect .. gather .. track + close +
    rows on the host (rank 0) .. barrier."""
    from vbt_amd import _lib, shard, synth
    from vbt_amd.ocsort import MultiClipTracker
    from vbt_amd.track import Pipeline
    L = _lib.lib()
    chunks = shard.frame_chunks(T, world)
    s0, s1 = chunks[rank]
    nmine = s1 - s0
    per = max(e - b_ for b_, e in chunks)
    bg = synth.background(0)
    base = np.stack([synth.render(bg, t) for t in range(U)])
    fr = torch.from_numpy(np.concatenate([base, base[:F]])).to(dev)          # a cycle of U frames: every run of F is contiguous
    pipe = Pipeline(MODEL, F, max_frames=8, fps=60.0, tracker_clips=1, device=local_rank)
    boxes = torch.zeros((per, 25, 4), dtype=torch.float32, device=dev)
    scores = torch.zeros((per, 25), dtype=torch.float32, device=dev)
    classes = torch.zeros((per, 25), dtype=torch.float32, device=dev)
    counts = torch.zeros((per,), dtype=torch.int32, device=dev)
    trk = MultiClipTracker(1, 8 * T + 75, max_age=30, asso_func="diou", iou_threshold=0.1, device=local_rank) if rank == 0 else None
    st = torch.cuda.current_stream().cuda_stream
    res = {}

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for rep in range(2):
        if trk is not None:
            trk.reset()
        fence()
        t0 = time.perf_counter()
        for f0 in range(s0, s1, F):
            nf = min(F, s1 - f0)
            o = f0 - s0
            s = f0 % U
            pipe.step_runs(fr[s:s + nf], [(0, 0, nf, f0 + 1)], track=False,
                           outputs=(boxes[o:o + nf], scores[o:o + nf], classes[o:o + nf], counts[o:o + nf]))
        pipe.join_detectors()
        rec = torch.cat([boxes.reshape(per, 100), scores, counts.to(torch.float32)[:, None]], dim=1)     # [per, 126] f32 = 504 B per frame
        if dist is not None:
            block = rec.to(cdev)
            allrec = torch.empty((world * per, shard.RECORD_FLOATS), dtype=torch.float32, device=cdev)
            dist.all_gather_into_tensor(allrec, block)                # the one exchange of this mode
            allrec = allrec.to(dev)
        else:
            allrec = rec
        if rank == 0:
            gb = allrec[:, :100].contiguous()
            gs = allrec[:, 100:125].contiguous()
            gc_ = allrec[:, 125].to(torch.int32).contiguous()
            for r_, (b_, e_) in enumerate(chunks):                    # one run per rank's chunk, in frame order (calls on a stream are ordered)
                if e_ > b_:
                    run = (_lib.Run * 1)(_lib.Run(0, r_ * per, 1, e_ - b_, b_ + 1, 1, 60.0))
                    _lib.check(L.vbt_tracker_update_from_detections_seq(trk.handle, gb.data_ptr(), gs.data_ptr(), gc_.data_ptr(), world * per,
                                                                        run, 1, 0.5, st))
            trk.finish(0.45, stream=st)
            best, rows_n, nph, ovf, ph = trk.summary(cap=512)
            cnt_rows, rows = trk.rows_all(stream=st)
            res.update(rows=int(rows_n.sum()), phases=int(nph.sum()), best_id=int(best[0]), checksum=zlib.crc32(rows[0, :cnt_rows[0]].tobytes()))
        fence()
        dt = time.perf_counter() - t0
    if dist is not None:
        tall = torch.zeros(world, dtype=torch.float64, device=cdev)
        tall[rank] = dt
        dist.all_reduce(tall)
        dt = float(tall.max().item())
    out = None
    if rank == 0:
        # the same clip on ONE rank's time-batched pipeline (outside the timed region): the rows must be the same rows
        ref = Pipeline(MODEL, F, max_frames=T, fps=60.0, tracker_clips=1, device=local_rank)
        for f0 in range(0, T, F):
            s = f0 % U
            ref.step_runs(fr[s:s + F], [(0, 0, F, f0 + 1)])
        rb, rr, rn, ro, _ = ref.close(cap=512)
        rc, rrows = ref.rows_all()
        same = int(rr.sum()) == res["rows"] and int(rb[0]) == res["best_id"] and zlib.crc32(rrows[0, :rc[0]].tobytes()) == res["checksum"]
        out = {"frames_per_s": T / dt, "seconds": dt, "frames": T, "n_gpus": world, "frames_per_rank": per, "record_bytes_per_frame": 4 * shard.RECORD_FLOATS,
               "gather_bytes": 4 * shard.RECORD_FLOATS * per * world, "rows": res["rows"], "phases": res["phases"],
               "rows_equal_single_gpu_time_batched": bool(same),
               "note": "one 4096-frame clip: every rank detects a contiguous frame chunk (64 frames per forward), one all-gather of the 504-byte "
                       "per-frame detection records, rank 0 walks all frames on the device and closes the clip; bound by rank 0's sequential walk"}
        del ref
    del pipe
    return out


def roofline_block(pipe, n, stream):
    """Dominant kernel family of the forward: all of its launches timed as ONE HIP-event bracket on the launch stream (one
    forward in flight, pipeline idle), i.e. without an event pair around every short launch; the committed rocprofv3
    --kernel-trace --stats summary of the same plan (profiles/) gives the same average launch duration.
    achieved = SURVEY 8d algorithmic bytes per launch / that duration; traffic = PMC HBM bytes per launch of the same plan."""
    from vbt_amd import _lib
    L = _lib.lib()
    stats = (_lib.KernelStat * 16)()
    cnt = ctypes.c_int()
    _lib.check(L.vbt_model_kernel_stats(pipe.interpreter.handle, n, stats, 16, ctypes.byref(cnt)))
    ms = (ctypes.c_double * 16)()
    _lib.check(L.vbt_model_profile_families(pipe.interpreter.handle, n, 20, stream, ms, 16))
    fam = max(range(cnt.value), key=lambda i: ms[i])
    s = stats[fam]
    name = s.name.decode()
    per_launch_s = ms[fam] * 1e-3 / s.launches
    achieved = (s.algorithmic_bytes / s.launches) / per_launch_s
    traffic = counters = prof_us = src = None   # from the separate rocprofv3 --pmc passes of this same plan (tools/make_profile_summary.py)
    for path in COUNTERS:
        try:
            tj = json.load(open(path))
            fj = tj["families"].get(name)
            if tj.get("batch") == n and fj and fj["launches"] == s.launches:
                traffic = fj["hbm_bytes_per_launch"]
                prof_us = fj["kernel_us_per_forward"] / fj["launches"]
                counters = tj.get("derived", {}).get("per_family", {}).get(name)
                src = os.path.basename(path)
                break
        except (OSError, ValueError, KeyError):
            continue
    limiter = None
    if counters:
        # which roof the counters point at: neither HBM nor MFMA is the limiter when both sit far below their peaks
        hbm_frac = (traffic / per_launch_s / HBM_PEAK) if traffic else 0.0
        mfma = counters.get("mfma_busy_frac", 0.0)
        limiter = {"hbm_traffic_frac": hbm_frac, "mfma_busy_frac": mfma, "wait_any_frac": counters.get("wait_any_frac"),
                   "active_inst_frac": counters.get("active_inst_frac"),
                   "reading": "vector-ALU issue + waits between phases; neither the HBM nor the MFMA roof" if max(hbm_frac, mfma) < 0.5
                   else ("hbm" if hbm_frac >= mfma else "mfma")}
    # which roof bounds the family: the roofline model's own rule - attainable = min(MFMA peak, arithmetic intensity x HBM peak) - on the
    # family's algorithmic intensity (ops per algorithmic byte).  The int8 EfficientDet blocks sit far left of the ridge
    # (625 op/B), so the memory roof is the binding one; `mfma` reports the other roof's figures beside it.
    ops_per_launch = 2.0 * s.macs / s.launches
    ai = ops_per_launch / (s.algorithmic_bytes / s.launches)
    bound = "hbm" if ai * HBM_PEAK < MFMA_I8_PEAK else "mfma"
    mfma_achieved = ops_per_launch / per_launch_s
    mfma_blk = {"achieved": mfma_achieved / 1e12, "peak": MFMA_I8_PEAK / 1e12, "unit": "TOP/s", "frac": mfma_achieved / MFMA_I8_PEAK,
                "arithmetic_intensity_op_per_byte": ai, "ridge_op_per_byte": MFMA_I8_PEAK / HBM_PEAK}
    if bound == "mfma":
        return {"bound": "mfma", "kernel": name, "achieved": mfma_blk["achieved"], "peak": mfma_blk["peak"], "unit": "TFLOP/s", "frac": mfma_blk["frac"],
                "traffic": traffic, "hbm": {"achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK},
                "launches_per_step": s.launches, "avg_launch_us": per_launch_s * 1e6, "limiter": limiter}
    return {"bound": bound, "kernel": name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "traffic": traffic, "mfma": mfma_blk,
            "frac_traffic": (traffic / per_launch_s / HBM_PEAK) if traffic else None,
            "launches_per_step": s.launches, "avg_launch_us": per_launch_s * 1e6, "profiles_avg_launch_us": prof_us, "profiles_source": src,
            "algorithmic_bytes_per_launch": s.algorithmic_bytes / s.launches,
            "bound_note": "`bound` = the roof the roofline model says binds this family (algorithmic ops per algorithmic byte against the ridge "
                          "MFMA peak / HBM peak); `frac` = algorithmic bytes per launch / launch time / HBM peak (SURVEY 8d); `limiter` is what the "
                          "PMC counters of this family say actually limits it",
            "limiter": limiter,
            "families_ms_per_step": {stats[i].name.decode(): round(ms[i], 4) for i in range(cnt.value) if ms[i] > 0},
            "whole_net_algorithmic_GBps": sum(stats[i].algorithmic_bytes for i in range(cnt.value)) /
            (sum(ms[i] for i in range(cnt.value)) * 1e-3) / 1e9}


if __name__ == "__main__":
    sys.exit(main() or 0)
