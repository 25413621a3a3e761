"""The CPU restatement of the detector (oracle/detector.c) under AddressSanitizer + UBSan (sanitizers run on
the CPU build only): `make -C oracle asan`, then a child Python with libasan preloaded runs the oracle on two frames, single and
batched, and its detections must equal the ones the ordinary build gives - with no sanitizer report on stderr."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_detector_is_clean_under_asan_and_ubsan(model_path, oracle_lib, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    from vbt_amd import synth
    frames = np.stack([synth.render(synth.background(3), t) for t in (0, 9)])
    want = oracle_lib.run_batch(model_path, frames, threads=2)
    np.save(tmp_path / "f.npy", frames)
    code = (
        "import sys, numpy as np\n"
        "from oracle import detector_ref\n"
        "fr = np.load(sys.argv[1])\n"
        "det = detector_ref.OracleDetector(sys.argv[2])\n"
        "one = det.run(fr[0])\n"
        "b, s, c, n = detector_ref.run_batch(sys.argv[2], fr, threads=2)\n"
        "assert np.array_equal(one[0], b[0]) and one[3] == n[0]\n"
        "np.savez(sys.argv[3], b=b, s=s, c=c, n=n)\n")
    env = dict(os.environ, LD_PRELOAD=libasan, VBT_ORACLE_LIB=os.path.join(ROOT, "oracle", "libvbt_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-c", code, str(tmp_path / "f.npy"), model_path, str(tmp_path / "o.npz")], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    got = np.load(tmp_path / "o.npz")
    for k, w in zip(("b", "s", "c", "n"), want):
        assert np.array_equal(got[k], w), k
