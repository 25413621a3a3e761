#!/usr/bin/env python3
"""Developer probe: A/B of HIP runtime knobs on the small-batch pipeline (batch 1 / 8, four forwards in flight).  Every setting runs
tools/b1_probe.py in a FRESH process (the knobs are read when the HIP runtime initialises)."""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SETTINGS = [
    {},
    {"HIP_FORCE_DEV_KERNARG": "1"},
    {"HIP_FORCE_DEV_KERNARG": "0"},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"},
    {"HIP_FORCE_DEV_KERNARG": "1", "DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
    {"DEBUG_HIP_KERNARG_COPY_OPT": "1"},
    {"ROC_USE_FGS_KERNARG": "0"},
    {"AMD_OPT_FLUSH": "0"},
    {"DEBUG_CLR_SKIP_RELEASE_SCOPE": "1"},
    {"GPU_STREAMOPS_CP_WAIT": "1"},
    {"ROC_ACTIVE_WAIT_TIMEOUT": "100"},
    {"DEBUG_HIP_DYNAMIC_QUEUES": "0"},
    {"GPU_MAX_HW_QUEUES": "5"},
    {"GPU_MAX_HW_QUEUES": "8"},
]
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,8").split(",")]
for s in SETTINGS:
    for nb in batches:
        env = dict(os.environ, DEPTHS=os.environ.get("DEPTHS", "4"), **s)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "b1_probe.py"), str(nb)], env=env, capture_output=True, text=True, timeout=300)
        rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
        out = {"env": s, "batch": nb, "rc": p.returncode}
        for r in rows:
            out[("track" if r["track"] else "det") + f"_d{r['depth']}"] = round(r["frames_per_s"])
        if p.returncode:
            out["err"] = p.stderr[-300:]
        print(json.dumps(out), flush=True)
