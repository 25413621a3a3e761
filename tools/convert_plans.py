#!/usr/bin/env python3
"""Re-write the pinned plan files (profiles/plan_*.b<batch>.f0) in format 2 - kernel families by name - on the GPU box: every plan is
loaded by the library it was tuned for and saved again (VBT_PLAN_CONVERT).  Output under gpurun_out/plans_v2/; copy into profiles/."""
import os
import re
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "plans_v2")
os.makedirs(OUT, exist_ok=True)
os.environ["VBT_PLAN_CONVERT"] = "1"
from vbt_amd.interpreter import Interpreter  # noqa: E402

MODELS = {"plan_lite0": "efficientdet_lite0_synth.vbtm", "plan_lite2": "efficientdet_lite2_synth.vbtm"}
for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
    mt = re.match(r"(plan_lite\d)\.b(\d+)\.f0$", f)
    if not mt:
        continue
    shutil.copy(os.path.join(ROOT, "profiles", f), OUT)
    before = open(os.path.join(OUT, f)).read()
    os.environ["VBT_PLAN_FILE"] = os.path.join(OUT, mt.group(1))
    it = Interpreter(os.path.join(ROOT, "models", MODELS[mt.group(1)]), max_batch=int(mt.group(2)))
    after = open(os.path.join(OUT, f)).read()
    same = [ln.split()[:2] for ln in before.split("\n")[1:] if ln] == [ln.split()[:2] for ln in after.split("\n")[1:] if ln]
    print(f, it.num_launches(), "launches;", "format 2, same alternatives" if after.startswith("VBTPLAN2") and same else "RE-TUNED (the old file no longer fitted)", flush=True)
    del it
