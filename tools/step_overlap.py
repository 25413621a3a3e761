"""Developer probe: which plan steps can the other forwards in flight hide?  Every step `reps` times on one stream vs on three
streams at once (vbt_model_profile_overlap); conc / single = 1 -> the kernel saturates some resource, 0.33 -> pure latency."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vbt_amd import _lib, spec
from vbt_amd.interpreter import Interpreter

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = os.environ.get("VBT_MODEL", os.path.join(os.path.dirname(__file__), "..", "models", "efficientdet_lite0_synth.vbtm"))
it = Interpreter(model, max_batch=B, flags=int(os.environ.get("VBT_FLAGS", "0")))
size = int(it.get_input_details()[0]["shape"][1])
g = spec.build_graph({320: 0, 384: 1, 448: 2}[size])
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
fd = torch.zeros((B, size, size, 3), dtype=torch.uint8, device="cuda:0")
out = (_lib.StepTime * 512)()
n = ctypes.c_int()
_lib.check(L.vbt_model_profile_steps(it.handle, fd.data_ptr(), B, 2, st, out, 512, ctypes.byref(n)))   # names + one warm forward
single = np.zeros(512, np.float32)
conc = np.zeros(512, np.float32)
_lib.check(L.vbt_model_profile_overlap(it.handle, B, 20, NS, single.ctypes.data, conc.ctypes.data, 512, ctypes.byref(n)))
ts = tc = 0.0
fam = {}
for i in range(n.value):
    s = out[i]
    a, b = g.ops[s.first_op], g.ops[s.op]
    ts += single[i]; tc += conc[i]
    f = fam.setdefault(s.family.decode(), [0, 0.0, 0.0])
    f[0] += 1; f[1] += single[i]; f[2] += conc[i]
    print(f"{i:3d} {s.family.decode():20s} v{s.variant:<3d} {a.name:>18s}..{b.name:<18s} single {single[i] * 1e3:7.1f} us  x{NS} {conc[i] * 1e3:7.1f} us/launch  ratio {conc[i] / single[i]:.2f}")
print(f"sum: single {ts:.3f} ms, {NS} streams {tc:.3f} ms per forward")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:22s} n{v[0]:3d} single {v[1] * 1e3:7.1f} us  x{NS} {v[2] * 1e3:7.1f} us  ratio {v[2] / v[1]:.2f}")
