"""Pins the round-1 GPU fault fix (out-of-bounds read of packed project weights in the NBP = 5 instantiation, exposed by
creating a Lite2 pipeline after a Lite0 one): the exact sequence runs once in a FRESH process with VBT_DEBUG_FENCE=1, where
every model buffer ends at the end of its own 2 MiB-granular allocation, so a read past the documented slack faults instead
of silently hitting a neighbour.  (The environment variable must be set before the library loads: hence the child.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["VBT_ROOT"])
import numpy as np, torch
from vbt_amd import synth
from vbt_amd.interpreter import Interpreter
lite2, lite0 = sys.argv[1], sys.argv[2]
for path, size in ((lite2, 448), (lite0, 320), (lite2, 448)):          # Lite2 -> Lite0 -> Lite2, each through the autotuner
    it = Interpreter(path, max_batch=4)                                 # flags 0: autotuned plan (every alternative is launched)
    frames = np.stack([synth.render(synth.background(90 + i, size), 3 * i) for i in range(4)])
    b, s, c, n = it.detect(frames)
    assert b.shape == (4, 25, 4) and np.all(n >= 0) and np.all(n <= 25)
    it2 = Interpreter(path, max_batch=4, flags=8 | 512)                 # heuristic plan with whole-image blocks
    b2, s2, c2, n2 = it2.detect(frames)
    assert np.array_equal(b, b2) and np.array_equal(s, s2) and np.array_equal(n, n2)
    del it, it2
torch.cuda.synchronize()
print("fence ok")
"""


def test_lite2_then_lite0_under_the_electric_fence(tmp_path, model_path):
    lite2 = str(tmp_path / "efficientdet_lite2_synth.vbtm")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_model.py"), "--arch", "2", "--out", lite2, "--calib", "2"])
    script = tmp_path / "fence_child.py"
    script.write_text(CHILD)
    env = dict(os.environ, VBT_ROOT=ROOT, VBT_DEBUG_FENCE="1")
    env.pop("VBT_PLAN_FILE", None)
    r = subprocess.run([sys.executable, str(script), lite2, model_path], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "fence ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
