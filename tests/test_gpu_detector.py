"""GPU parity: HIP detector vs the CPU oracle, bit-exact (integer pipeline) through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    from vbt_amd import synth
    return np.concatenate([synth.clip_frames(s, 11 * s, 2) for s in range(3)])   # 6 frames, 3 clips


@pytest.fixture(scope="module")
def oracle_run(oracle_lib, model_path, frames):
    det = oracle_lib.OracleDetector(model_path)
    outs, tensors = [], []
    for f in frames:
        outs.append(det.run(f))
        tensors.append([det.tensor(t) for t in range(1, det.num_tensors - 1)])
    return outs, tensors


# plan flags (include/vbt_hip.h): 1 = one kernel per graph op; 8 = no autotuning -> the most fused alternative of
# every group (incl. whole BiFPN nodes; low-resolution MBConv blocks as whole-image expand+depthwise + projection GEMM
# with the residual in its epilogue; BiFPN nodes and head layers on row bands); 8|4096 / 8|8192 = the same with those
# blocks / layers on the 64-pixel tile kernel; 8|128 = one launch per head layer and level; 8|2 = dw+project fused,
# expand separate; 8|16 = no BiFPN node fusion; 8|32768 = the P6 conv, its two max pools and the five lateral convs as eight launches instead
# of one (pw_multi_kernel); 0 = autotuned mix (whatever is fastest on this GPU)
@pytest.mark.parametrize("flags", [1, 8, 8 | 4096 | 8192, 8 | 8192, 8 | 2, 8 | 16, 8 | 128 | 8192, 8 | 128, 8 | 256, 8 | 512 | 4096, 8 | 2048 | 4096, 8 | 16384, 8 | 16384 | 2048, 8 | 32768, 0])
def test_every_tensor_bit_exact(model_path, frames, oracle_run, flags):
    """Every plan must reproduce the oracle bit for bit: all 250 tensors when unfused, every tensor that still
    reaches HBM otherwise (fused MBConv / SeparableConv blocks keep their intermediates in LDS)."""
    from vbt_amd.interpreter import Interpreter
    outs, tensors = oracle_run
    B = len(frames)
    fuse = flags != 1
    it = Interpreter(model_path, max_batch=B, flags=flags)
    boxes, scores, classes, counts = it.detect(frames)
    bad = []
    checked = 0
    if flags == 1:
        assert it.num_launches() == 251 and all(it.materialized(t) for t in range(1, it.num_tensors() - 1))
    if flags == 8:
        assert it.num_launches() < 70
        assert Interpreter(model_path, max_batch=B, flags=8 | 32768).num_launches() == it.num_launches() + 7   # 6 convs + 2 pools -> 1 launch
    for tid in range(1, it.num_tensors() - 1):
        if not it.materialized(tid):
            continue
        checked += 1
        got = it.read_tensor(tid, B)
        for b in range(B):
            if not np.array_equal(got[b], tensors[b][tid - 1]):
                d = np.abs(got[b].astype(int) - tensors[b][tid - 1].astype(int))
                bad.append((tid, b, int(d.max()), float((d > 0).mean())))
                break
    assert not bad, f"first mismatching tensors (id, frame, max|diff|, frac): {bad[:8]}"
    assert checked == 250 if flags == 1 else checked > 60
    for b in range(B):
        ob, os_, oc, on = outs[b]
        assert counts[b] == on
        assert np.array_equal(scores[b], os_)
        assert np.array_equal(boxes[b], ob)
        assert np.array_equal(classes[b], oc)


@pytest.mark.parametrize("env", [{"VBT_PW_VARIANT": "3"}, {"VBT_PW_VARIANT": "4"}, {"VBT_PW_VARIANT": "2"}, {"VBT_PW_VARIANT": "5"}, {"VBT_PW_VARIANT": "6"}, {"VBT_XD_VARIANT": "1"}, {"VBT_XD_VARIANT": "3"},
                                 {"VBT_XD_VARIANT": "101"}, {"VBT_XD_VARIANT": "103"}, {"VBT_XD_VARIANT": "106", "VBT_PW_VARIANT": "4"},
                                 {"VBT_XD_VARIANT": "201"}, {"VBT_XD_VARIANT": "203"},
                                 {"VBT_SUBSTREAMS": "2"}, {"VBT_SUBSTREAMS": "3", "VBT_GRAPH_MAX_BATCH": "0"}])
def test_forced_kernel_variants_bit_exact(model_path, frames, oracle_run, env):
    """Kernel variants the autotuner may or may not pick on a given day, forced through the test-only environment overrides (read
    once per process, hence a child process per case): the large-K pointwise conv with its weights shared through LDS (3 / 4; 5 / 6: the block's whole weight panel in LDS and a K loop without barriers) or split
    over the waves (2), and both forms of the expand + depthwise kernel (chunks per workgroup; 100 + n = the second form on 8 waves, 200 + n on 16).  Every
    materialised tensor and every detection must equal the oracle's.
    VBT_SUBSTREAMS: the batch split over side streams (the merged lateral-conv launch holds whole-batch pointers and has to be
    left unmerged there, like the row-band kernels)."""
    import os, subprocess, sys, tempfile, pickle
    outs, tensors = oracle_run
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "in.pkl")
        with open(src, "wb") as f:
            pickle.dump((model_path, frames), f)
        code = (
            "import pickle, sys, numpy as np\n"
            "from vbt_amd.interpreter import Interpreter\n"
            "model, frames = pickle.load(open(sys.argv[1], 'rb'))\n"
            "B = len(frames)\n"
            "it = Interpreter(model, max_batch=B, flags=8)\n"
            "det = it.detect(frames)\n"
            "ten = {t: it.read_tensor(t, B) for t in range(1, it.num_tensors() - 1) if it.materialized(t)}\n"
            "pickle.dump((det, ten), open(sys.argv[2], 'wb'))\n")
        dst = os.path.join(td, "out.pkl")
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        subprocess.run([sys.executable, "-c", code, src, dst], check=True, cwd=root, env={**os.environ, **env}, timeout=300)
        (boxes, scores, classes, counts), ten = pickle.load(open(dst, "rb"))
    assert len(ten) > 60
    for tid, got in ten.items():
        for b in range(len(frames)):
            assert np.array_equal(got[b], tensors[b][tid - 1]), f"tensor {tid} of frame {b} differs under {env}"
    for b in range(len(frames)):
        ob, os_, oc, on = outs[b]
        assert counts[b] == on and np.array_equal(scores[b], os_) and np.array_equal(boxes[b], ob) and np.array_equal(classes[b], oc)


def test_batch_tail_and_single_frame(model_path, frames, oracle_run):
    """B=1 (the reference's call shape) and a batch smaller than max_batch."""
    from vbt_amd.interpreter import Interpreter
    outs, _ = oracle_run
    it = Interpreter(model_path, max_batch=4)
    run = it.get_signature_runner()
    out = run(images=frames[2:3])
    ob, os_, oc, on = outs[2]
    assert int(np.squeeze(out["output_0"])) == on
    assert np.array_equal(np.squeeze(out["output_1"]), os_)
    assert np.array_equal(np.squeeze(out["output_3"]), ob)
    b3 = it.detect(frames[:3])
    for b in range(3):
        assert b3[3][b] == outs[b][3] and np.array_equal(b3[0][b], outs[b][0])


def test_explicit_clamps_bit_exact(tmp_path, model_path, oracle_lib, frames):
    """The synthetic model's activation ranges all end at the int8 limits, where the requantisation leaves the clamp
    to the saturating u8 conversion.  Narrow the range of two convs in three so the explicit-clamp path runs too."""
    from vbt_amd.container import Container
    from vbt_amd.interpreter import Interpreter
    raw = bytearray(open(model_path, "rb").read())
    c = Container(model_path)
    ops = np.frombuffer(raw, dtype=c.ops.dtype, count=len(c.ops), offset=128 + 32 * len(c.tensors))
    n = 0
    for i, r in enumerate(ops):
        if int(r["type"]) in (1, 2, 3) and i % 3 != 0:
            r["act_min"], r["act_max"] = max(int(r["act_min"]), -101 + i % 7), min(int(r["act_max"]), 96 - i % 5)
            n += 1
    assert n > 100
    path = str(tmp_path / "clamped.vbtm")
    open(path, "wb").write(bytes(raw))
    det = oracle_lib.OracleDetector(path)
    for flags in (1, 8, 8 | 4096 | 8192, 0):
        it = Interpreter(path, max_batch=2, flags=flags)
        boxes, scores, classes, counts = it.detect(frames[:2])
        for b in range(2):
            ob, os_, oc, on = det.run(frames[b])
            assert counts[b] == on and np.array_equal(scores[b], os_) and np.array_equal(boxes[b], ob)
            for tid in range(1, it.num_tensors() - 1):
                if it.materialized(tid):
                    assert np.array_equal(it.read_tensor(tid, 2)[b], det.tensor(tid)), (flags, tid)


def test_both_accumulator_flavours_bit_exact(tmp_path, model_path, oracle_lib, frames, monkeypatch):
    """A conv whose int32 accumulators provably stay inside (-2^22, 2^22) starts them at bias + 0x4B400000 and reads them back as
    floats (two packed adds instead of four v_cvt_f32_i32); the proof is 255 * sum|w| + |bias| < 2^22 per output
    channel.  (a) VBT_NO_KBIAS=1: no conv takes that form; (b) a container with |bias| = 5e6 on some channels of every third conv:
    the proof fails there (and the outputs saturate), the other convs keep the biased form.  Every materialised tensor and every
    detection equals the oracle's in the fused plan modes either way."""
    from vbt_amd.container import Container
    from vbt_amd.interpreter import Interpreter
    raw = bytearray(open(model_path, "rb").read())
    c = Container(model_path)
    bo = int(c.header["blob_offset"])
    n = 0
    for i, r in enumerate(c.ops):
        if int(r["type"]) in (1, 2, 3) and i % 3 == 0:
            cout = int(c.tensors[int(r["output"])]["c"])
            b = np.frombuffer(raw, dtype="<i4", count=cout, offset=bo + int(r["b_off"]))
            b[::5] = np.where(np.arange(len(b[::5])) % 2 == 0, 5_000_000, -5_000_000)
            n += 1
    assert n > 30
    big = str(tmp_path / "bigbias.vbtm")
    open(big, "wb").write(bytes(raw))
    for path, env in ((model_path, "1"), (big, None), (big, "1")):
        if env is None:
            monkeypatch.delenv("VBT_NO_KBIAS", raising=False)
        else:
            monkeypatch.setenv("VBT_NO_KBIAS", env)
        det = oracle_lib.OracleDetector(path)
        for flags in (8, 0):
            it = Interpreter(path, max_batch=2, flags=flags)
            boxes, scores, classes, counts = it.detect(frames[:2])
            for b in range(2):
                ob, os_, oc, on = det.run(frames[b])
                assert counts[b] == on and np.array_equal(scores[b], os_) and np.array_equal(boxes[b], ob), (path, env, flags)
                for tid in range(1, it.num_tensors() - 1):
                    if it.materialized(tid):
                        assert np.array_equal(it.read_tensor(tid, 2)[b], det.tensor(tid)), (path, env, flags, tid)


def test_extreme_frames_bit_exact(model_path, oracle_lib):
    """All-black, all-white, uniform noise and a 1-pixel checkerboard: inputs that drive many activations into the
    saturating ends of their int8 ranges (the requantisation leaves that clamp to the u8 conversion)."""
    from vbt_amd.interpreter import Interpreter
    rng = np.random.Generator(np.random.PCG64(11))
    S = 320
    chk = ((np.add.outer(np.arange(S), np.arange(S)) & 1) * 255).astype(np.uint8)
    frames = np.stack([np.zeros((S, S, 3), np.uint8), np.full((S, S, 3), 255, np.uint8),
                       rng.integers(0, 256, (S, S, 3), dtype=np.uint8), np.repeat(chk[:, :, None], 3, axis=2)])
    det = oracle_lib.OracleDetector(model_path)
    want = [det.run(f) for f in frames]
    tens = [[det.tensor(t) for t in range(1, det.num_tensors - 1)] for f in frames if det.run(f) is not None]
    for flags in (8, 8 | 2048, 0):
        it = Interpreter(model_path, max_batch=len(frames), flags=flags)
        boxes, scores, classes, counts = it.detect(frames)
        for b in range(len(frames)):
            ob, os_, oc, on = want[b]
            assert counts[b] == on and np.array_equal(scores[b], os_) and np.array_equal(boxes[b], ob), (flags, b)
        for tid in range(1, it.num_tensors() - 1):
            if it.materialized(tid):
                got = it.read_tensor(tid, len(frames))
                for b in range(len(frames)):
                    assert np.array_equal(got[b], tens[b][tid - 1]), (flags, tid, b)


@pytest.fixture(scope="module")
def tie_model(tmp_path_factory):
    """Lite0 with every ADD's scale ratios forced to exactly 0.5 / 1.0 (tools/make_model.py --tie_adds): the rounding of
    XNNPACK's integer ADD (half towards +infinity, where the float form this build used before rounds to even) is then hit
    on about half of all elements of every residual add, partial sum and BiFPN sum."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = str(tmp_path_factory.mktemp("models") / "efficientdet_lite0_tie.vbtm")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_model.py"), "--arch", "0", "--out", out, "--calib", "4", "--tie_adds"])
    return out


@pytest.mark.parametrize("flags", [1, 8, 8 | 4096 | 8192, 8 | 2, 8 | 16, 8 | 512 | 4096, 8 | 16384 | 2048, 0])
def test_integer_add_ties_bit_exact_in_every_fused_path(tie_model, oracle_lib, frames, flags):
    """add_kernel, the residual epilogues (tile and whole-image MBConv), the node load stage (2-input and chained 3-input
    sums) and the node chain all evaluate the same integer ADD: every tensor equals the oracle on the tie model."""
    from vbt_amd.container import Container
    from vbt_amd.interpreter import Interpreter
    c = Container(tie_model)
    adds = [r for r in c.ops if int(r["type"]) == 4]
    assert len(adds) == 42 and all(int(r["n_inputs"]) == 2 for r in adds)
    assert {(int(r["add_q"][1]), int(r["add_q"][2]), int(r["add_q"][3])) for r in adds} == {(2 ** 19, 2 ** 20, 20), (2 ** 20, 2 ** 19, 20), (2 ** 20, 2 ** 20, 21)}
    det = oracle_lib.OracleDetector(tie_model)
    B = 3
    want, tens = [], []
    for f in frames[:B]:
        want.append(det.run(f))
        tens.append([det.tensor(t) for t in range(1, det.num_tensors - 1)])
    # the ties are really there: in a 0.5/0.5 add an odd a + b (zero points aside) sits exactly between two outputs
    r = next(r for r in adds if int(r["add_q"][3]) == 21)
    a, b = tens[0][int(r["inputs"][0]) - 1].astype(int), tens[0][int(r["inputs"][1]) - 1].astype(int)
    za, zb = int(c.tensors[int(r["inputs"][0])]["zero_point"]), int(c.tensors[int(r["inputs"][1])]["zero_point"])
    assert 0.3 < np.mean(((a - za) + (b - zb)) % 2 == 1) < 0.7
    it = Interpreter(tie_model, max_batch=B, flags=flags)
    boxes, scores, classes, counts = it.detect(frames[:B])
    n = 0
    for tid in range(1, it.num_tensors() - 1):
        if it.materialized(tid):
            got = it.read_tensor(tid, B)
            for k in range(B):
                assert np.array_equal(got[k], tens[k][tid - 1]), (flags, tid, k)
            n += 1
    assert n > 60
    for k in range(B):
        ob, os_, oc, on = want[k]
        assert counts[k] == on and np.array_equal(scores[k], os_) and np.array_equal(boxes[k], ob)


def test_plan_file_with_a_foreign_kernel_name_is_refused(model_path, tmp_path, monkeypatch):
    """Plan files name the kernel family of every step (format 2): a file whose names do not fit this build of the planner - tuned for
    another set of alternatives - must not select kernels by bare index; the library refuses it, re-tunes and re-writes it.  A format-1
    file (indices only) and the pinned format-2 file both load unchanged."""
    import shutil
    from conftest import ROOT
    from vbt_amd.interpreter import Interpreter
    pinned = os.path.join(ROOT, "profiles", "plan_lite0.b8.f0")
    good = open(pinned).read()
    assert good.startswith("VBTPLAN2 ")
    prefix = str(tmp_path / "plan")
    monkeypatch.setenv("VBT_PLAN_FILE", prefix)
    shutil.copy(pinned, prefix + ".b8.f0")
    Interpreter(model_path, max_batch=8)
    assert open(prefix + ".b8.f0").read() == good                       # accepted as it is
    bad = good.replace("fused_mbconv:", "fused_mbconv_of_another_build:", 1)
    open(prefix + ".b8.f0", "w").write(bad)
    it = Interpreter(model_path, max_batch=8)
    after = open(prefix + ".b8.f0").read()
    assert after != bad and after.startswith("VBTPLAN2 ") and "of_another_build" not in after and it.num_launches() < 70
    legacy = "\n".join([good.split("\n")[0].split()[1]] + [" ".join(t.rsplit(":", 1)[-1] for t in ln.split()) for ln in good.split("\n")[1:]])
    open(prefix + ".b8.f0", "w").write(legacy)
    Interpreter(model_path, max_batch=8)
    assert open(prefix + ".b8.f0").read() == legacy                     # format 1 still loads (and is left alone without VBT_PLAN_CONVERT)
