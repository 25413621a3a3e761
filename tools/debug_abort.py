import ctypes, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from click.testing import CliRunner
from vbt_amd import synth
from vbt_amd.cli import main
root = os.path.join(os.path.dirname(__file__), "..")
model = os.path.join(root, "models", "efficientdet_lite0_synth.vbtm")
tmp = tempfile.mkdtemp()
l2 = os.path.join(tmp, "l2.vbtm")
subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_model.py"), "--arch", "2", "--out", l2, "--calib", "4"], stdout=subprocess.DEVNULL)
np.save(os.path.join(tmp, "c.npy"), synth.clip_frames(12, 0, 12, size=416))
res = CliRunner().invoke(main, ["track", os.path.join(tmp, "c.npy"), "--model", model, "--fps", "60", "--detection_treshold", "0.3"])
print("cli:", res.exit_code, res.output.strip()[:100], flush=True)
from vbt_amd.track import Pipeline
print("creating lite2 pipeline", flush=True)
pipe = Pipeline(l2, 2, max_frames=5, fps=30.0, depth=int(os.environ.get("DBG_DEPTH", "3")))
print("created", flush=True)
