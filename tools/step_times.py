"""Developer probe: per-launch times of the detector plan (one forward in flight), with graph op names."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vbt_amd import _lib, spec, synth
from vbt_amd.interpreter import Interpreter

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = os.environ.get("VBT_MODEL", os.path.join(os.path.dirname(__file__), "..", "models", "efficientdet_lite0_synth.vbtm"))
flags = int(os.environ.get("VBT_FLAGS", "0"))
it = Interpreter(model, max_batch=B, flags=flags)
size = int(it.get_input_details()[0]["shape"][1])
g = spec.build_graph({320: 0, 384: 1, 448: 2}[size])
frames = np.stack([synth.render(synth.background(s, size), 3 * s) for s in range(min(B, 16))])
frames = np.concatenate([frames] * ((B + len(frames) - 1) // len(frames)))[:B]
fd = torch.from_numpy(frames).to("cuda:0")
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
out = (_lib.StepTime * 512)()
n = ctypes.c_int()
for _ in range(2):
    _lib.check(L.vbt_model_profile_steps(it.handle, fd.data_ptr(), B, 10, st, out, 512, ctypes.byref(n)))
tot = 0.0
for i in range(n.value):
    s = out[i]
    a, b = g.ops[s.first_op], g.ops[s.op]
    to = g.tensors[b.output]
    tot += s.ms
    print(f"{i:3d} {s.family.decode():20s} v{s.variant:<3d} {a.name:>18s}..{b.name:<18s} out {to.h:3d}x{to.w:<3d}x{to.c:<4d} {s.ms * 1e3:8.1f} us  "
          f"{s.macs / 1e6:9.1f} MMAC  {s.macs * 2 / (s.ms * 1e-3) / 1e12 if s.ms > 0 else 0:7.1f} Top/s  alg {s.algorithmic_bytes / 1e6:7.1f} MB "
          f"{s.algorithmic_bytes / (s.ms * 1e-3) / 1e9 if s.ms > 0 else 0:7.0f} GB/s")
print(f"total {tot:.3f} ms over {n.value} launches")
