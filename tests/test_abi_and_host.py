"""CPU-side checks: the C-ABI library loads and exports every symbol include/*.h declares
(no compute without a GPU), loud failure without a GPU, export logic, clip sharding, gloo gather."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _declared_in_header():
    names = set()
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):          # vbt_hip.h (the boundary) + vbt_hip_diag.h (measurement / tests)
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(vbt_[a-z_0-9]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from vbt_amd import _lib
    L = _lib.lib()
    names = _declared_in_header()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"libvbt_hip.so does not export {n}"
    assert sorted(_lib.declared_symbols()) == names        # the ctypes table binds exactly the header


def test_no_cpu_fallback_without_gpu(model_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vbt_amd import _lib
    from vbt_amd.interpreter import Interpreter
    from vbt_amd.ocsort import OCSort
    from vbt_amd.velocity import analyze_rows
    with pytest.raises(_lib.VbtError):
        Interpreter(model_path)
    with pytest.raises(_lib.VbtError):
        OCSort(max_age=30, asso_func="diou", iou_threshold=0.1)
    with pytest.raises(_lib.VbtError):
        analyze_rows(np.zeros((3, 7)))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vbt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libvbt_oracle" not in src and "detector_ref" not in src and "ocsort_np" not in src, f


def test_export_matches_reference_files():
    """reference track.py:103-126 on the reference's own rows: sort, retained index, id in the file name."""
    from vbt_amd.track import export_dataframe
    full = np.load(os.path.join(GOLDEN, "dfs_ocsort_full.npz"))
    cols = ("id", "time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width")
    for clip, name in (("001", "001_squat_6reps"), ("008", "008_sdl_9reps")):
        idx = full[f"c{clip}_index"]
        order = np.argsort(idx)                                     # emission order = original row labels
        data = {k: full[f"c{clip}_{k}"][order].tolist() for k in cols}
        df, best, path = export_dataframe(data, f"/videos/{name}.mp4", "models/efficientdet_lite0_whole.tflite", df_dir=None, write=False)
        assert path == f"{name}_id1_efficientdet_lite0_whole.pkl.gz" and best == 1
        assert list(df.columns) == list(cols) and df["id"].dtype == np.int64 and df["time"].dtype == np.float64
        assert np.array_equal(df.index.to_numpy(), idx)            # sort_values keeps the labels (1,3,5,... for id 1)
        assert np.array_equal(df["dx"].to_numpy(), full[f"c{clip}_dx"])


def test_clip_sharding_lpt():
    from vbt_amd.shard import shard_clips
    import json
    with open(os.path.join(GOLDEN, "phases_ocsort.json")) as f:
        ph = json.load(f)
    rows = {k: v["rows"] for k, v in ph.items() if len(k) == 3}
    assert len(rows) == 34
    shards = shard_clips(rows, 8)
    assert sorted(c for s in shards for c in s) == sorted(rows)
    loads = [sum(rows[c] for c in s) for s in shards]
    assert max(loads) <= 1.25 * (sum(loads) / 8)                   # longest-processing-time packing is near-balanced
    assert shard_clips(rows, 1) == [sorted(rows, key=lambda c: (-rows[c], c))]


WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["VBT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from vbt_amd.shard import shard_clips, gather_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
work = {f"c{i:02d}": 100 + 37 * i for i in range(7)}
mine = shard_clips(work, world)[rank]
rec = np.array([[int(c[1:]), work[c], rank] for c in mine], np.float64)
allrec = gather_records(torch.from_numpy(rec), dist, pad_to=8)
if rank == 0:
    got = sorted((int(r[0]), int(r[1]), int(r[2])) for r in allrec)
    print(json.dumps(got))
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_gather_over_gloo(tmp_path):
    """N > 1 path on CPU: clips are sharded, each rank produces its records, one all-gather collects them."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, VBT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    assert [g[0] for g in got] == list(range(7)) and {g[2] for g in got} == {0, 1}


WORKER_FRAMES = r"""
import os, sys, json
sys.path.insert(0, os.environ["VBT_ROOT"])
import numpy as np, torch.distributed as dist
from vbt_amd.shard import frame_chunks, pack_detection_records, gather_detection_records, unpack_detection_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
T = 37
s, e = frame_chunks(T, world)[rank]
f = np.arange(s, e)
boxes = (f[:, None, None] + np.arange(25)[None, :, None] * 0.01 + np.arange(4)[None, None, :] * 0.001).astype(np.float32)
scores = (f[:, None] * 0.5 + np.arange(25)[None, :] / 256.0).astype(np.float32)
counts = (f % 26).astype(np.int32)
allrec = gather_detection_records(pack_detection_records(boxes, scores, counts), T, dist)
b, sc, c = unpack_detection_records(allrec)
ok = len(c) == T and np.array_equal(c, np.arange(T) % 26) and np.allclose(b[:, 0, 0], np.arange(T)) and np.allclose(sc[:, 3], np.arange(T) * 0.5 + 3 / 256)
if rank == 0:
    print(json.dumps({"ok": bool(ok), "frames": int(len(c))}))
dist.barrier()
dist.destroy_process_group()
"""


def test_frame_major_detection_records_over_gloo(tmp_path):
    """SURVEY 8e, one long clip: ranks detect contiguous frame chunks, one all-gather returns the 504-byte
    per-frame records in frame order (uneven chunks are padded)."""
    from vbt_amd.shard import frame_chunks, records_to_tracker_inputs, pack_detection_records
    assert frame_chunks(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)] and frame_chunks(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    script = tmp_path / "worker_frames.py"
    script.write_text(WORKER_FRAMES)
    env = dict(os.environ, VBT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29654", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    assert json.loads(outs[0][0].strip().splitlines()[-1]) == {"ok": True, "frames": 37}
    # threshold + reorder on the owner (reference odt.py:70-75,102-118)
    boxes = np.zeros((2, 25, 4), np.float32); boxes[0, 0] = (0.1, 0.2, 0.3, 0.4); boxes[0, 1] = (0.5, 0.6, 0.7, 0.8)
    scores = np.zeros((2, 25), np.float32); scores[0, :2] = (0.9, 0.4)
    dets, cnt, times = records_to_tracker_inputs(pack_detection_records(boxes, scores, np.array([2, 0])), fps=30.0)
    assert cnt.tolist() == [[1], [0]] and np.allclose(dets[0, 0, 0], (0.2, 0.1, 0.4, 0.3, 0.9, 0.0)) and np.allclose(times[:, 0], (1 / 30, 2 / 30))


WORKER_BENCH = r"""
import os, sys, json
sys.path.insert(0, os.environ["VBT_ROOT"])
import numpy as np, torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, PH = 5, 32
rec = np.zeros((n, 3 + PH * 6), np.float64)
rec[:, 0] = rank * 100 + np.arange(n)              # best ids
rec[:, 1] = 10 + rank                               # rows
rec[:, 3:] = rank + np.arange(PH * 6)[None, :] / 1000.0
allrec = bench.gather_records(dist, rec, world, torch.device("cpu"))
ok = allrec.shape == (world, n, 3 + PH * 6) and all(np.array_equal(allrec[r, :, 0], r * 100 + np.arange(n)) and
                                                    np.all(allrec[r, :, 1] == 10 + r) and allrec[r, 2, 3 + 7] == r + 0.007 for r in range(world))
if rank == 0:
    print(json.dumps({"ok": bool(ok), "clips_with_result": int((allrec[..., 1] > 0).sum())}))
dist.barrier()
dist.destroy_process_group()
"""


def test_bench_gather_layout_two_ranks_over_gloo(tmp_path):
    """bench.py's result exchange (one all_gather_into_tensor of fixed-size per-clip records, concatenated layout) with two
    ranks on CPU: rank r's records land in block r, in clip order."""
    script = tmp_path / "worker_bench.py"
    script.write_text(WORKER_BENCH)
    env = dict(os.environ, VBT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    assert json.loads(outs[0][0].strip().splitlines()[-1]) == {"ok": True, "clips_with_result": 10}


def test_bench_self_launches_its_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher (the driver's command shape) must start N fresh rank processes through
    torch.distributed.run on 127.0.0.1 before touching the GPU, and return their exit code."""
    import bench
    calls = {}

    class P:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"] = cmd, env
        return P()
    import subprocess as sp
    monkeypatch.setattr(sp, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    import torch
    before = torch.cuda.is_initialized()
    assert bench.main() == 7
    assert torch.cuda.is_initialized() == before                     # the parent made no GPU call
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _build_c_host(tmp_path):
    import subprocess
    exe = str(tmp_path / "track_clip")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "examples", "track_clip.c"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + os.path.join(ROOT, "vbt_amd"), "-lvbt_hip", "-Wl,-rpath," + os.path.join(ROOT, "vbt_amd"), "-o", exe])
    return exe


def test_plain_c_host_builds_against_the_header_and_fails_loudly_without_a_gpu(tmp_path, model_path):
    """The boundary is a C ABI, not a Python module: examples/track_clip.c - the clip loop of reference track.py:129-260 from a C host
    (vbt_host_alloc, vbt_pipeline_create, vbt_track_clip) - compiles warning-free against include/vbt_hip.h and links against
    libvbt_hip.so; without a GPU it exits non-zero with the library's message (no CPU fallback)."""
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    exe = _build_c_host(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run is covered by tests/test_gpu_cli.py")
    raw = tmp_path / "clip.raw"
    np.zeros((2, 320, 320, 3), np.uint8).tofile(raw)
    p = subprocess.run([exe, model_path, str(raw), "2", "320", "320", "30"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and ("HIP" in p.stderr or "hip" in p.stderr), p.stderr
