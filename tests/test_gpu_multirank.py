"""The N > 1 path of bench.py rehearsed on ONE GPU (both ranks on device 0, gloo for the gather: a 1-GPU box cannot run RCCL
across devices): the line carries the whole-job value, the number of ranks the collective really spanned and every rank's
own rate; a rank that dies fails the job instead of hanging it."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
ENV = dict(os.environ, VBT_BENCH_SAME_DEVICE="1", VBT_BENCH_BACKEND="gloo", VBT_BENCH_TIMEOUT_S="60")


def test_two_rank_rehearsal_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--no-roofline", "--no-configs"],
                       env=ENV, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and len(j["per_rank_frames_per_s"]) == 2 and j["scaling"] == "weak"
    assert j["clips_with_result"] == 128                      # both ranks' clips arrived through the one all-gather
    assert abs(j["value"] - 2 * 64 * 8 / (j["ms_per_step"] * 8e-3)) < 1e-6 * j["value"]
    assert j["value"] <= sum(j["per_rank_frames_per_s"]) * (1 + 1e-9)   # the job's time is the slowest rank's


def test_two_rank_line_carries_the_sharded_configs():
    """N > 1: BASELINE config 5 (the 34-clip corpus LPT-sharded over the ranks, one all-gather of the result records) and SURVEY 8e's
    frame-major mode of one long clip (one all-gather of the 504-byte per-frame detection records, rank 0 tracks) are part of the
    line; the frame-major rows equal the single-GPU time-batched run of the same clip."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-roofline"],
                       env=ENV, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    c = j["configs"]
    cs, fm = c["corpus_sharded"], c["one_clip_frame_major"]
    assert cs["clips"] == 34 and cs["n_gpus"] == 2 and len(cs["per_rank_seconds"]) == 2 and cs["frames"] > 50000 and cs["rank0"]["overflow"] == 0
    assert abs(cs["frames_per_s"] - cs["frames"] / max(cs["per_rank_seconds"])) < 1e-6 * cs["frames_per_s"]
    assert fm["frames"] == 4096 and fm["n_gpus"] == 2 and fm["record_bytes_per_frame"] == 504 and fm["rows"] > 1000
    assert fm["rows_equal_single_gpu_time_batched"] is True


def test_a_dead_rank_fails_the_job():
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-roofline"],
                       env=dict(ENV, VBT_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and time.time() - t0 < 300
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
