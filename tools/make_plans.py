#!/usr/bin/env python3
"""Autotune and save the execution plans bench.py pins (profiles/plan_<model>.b<batch>.f0) for the configurations that do
not have one yet.  Run on the GPU box; writes under gpurun_out/plans/ (copy the files into profiles/ afterwards).
usage: python tools/make_plans.py [--retune PREFIX ...]     (--retune plan_lite2: tune those plans again instead of starting from the
pinned ones under profiles/ - needed after a kernel alternative was added: an old plan still loads, it just never picks it)"""
import os
import shutil
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "plans")
os.makedirs(OUT, exist_ok=True)
retune = [a for a in sys.argv[1:] if not a.startswith("--")]
for f in os.listdir(OUT):
    if any(f.startswith(r) for r in retune):
        os.remove(os.path.join(OUT, f))
for f in os.listdir(os.path.join(ROOT, "profiles")):
    if f.startswith("plan_") and not any(f.startswith(r) for r in retune):
        shutil.copy(os.path.join(ROOT, "profiles", f), OUT)
from vbt_amd.interpreter import Interpreter  # noqa: E402

for name, model, batches in (("plan_lite0", "efficientdet_lite0_synth.vbtm", (1, 8, 64, 256)), ("plan_lite2", "efficientdet_lite2_synth.vbtm", (64,))):
    os.environ["VBT_PLAN_FILE"] = os.path.join(OUT, name)
    for b in batches:
        t0 = time.time()
        it = Interpreter(os.path.join(ROOT, "models", model), max_batch=b)
        print(name, b, f"{time.time() - t0:.1f} s", flush=True)
        del it
print(sorted(os.listdir(OUT)))
