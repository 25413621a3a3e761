"""Developer probe (not the contract bench): detector-only throughput + per-family HIP-event split."""
import ctypes, sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vbt_amd import _lib, synth
from vbt_amd.interpreter import Interpreter

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
model = os.environ.get("VBT_MODEL", os.path.join(os.path.dirname(__file__), "..", "models", "efficientdet_lite0_synth.vbtm"))
it = Interpreter(model, max_batch=B)
print('launches per forward:', it.num_launches())
size = int(it.get_input_details()[0]["shape"][1])
frames = np.stack([synth.render(synth.background(s, size), 3 * s) for s in range(min(B, 16))])
frames = np.concatenate([frames] * ((B + len(frames) - 1) // len(frames)))[:B]
dev = torch.device("cuda:0")
fd = torch.from_numpy(frames).to(dev)
boxes = torch.empty((B, 25, 4), dtype=torch.float32, device=dev)
scores = torch.empty((B, 25), dtype=torch.float32, device=dev)
classes = torch.empty((B, 25), dtype=torch.float32, device=dev)
counts = torch.empty((B,), dtype=torch.int32, device=dev)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def step():
    _lib.check(L.vbt_detect_async(it.handle, fd.data_ptr(), B, st, boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(), counts.data_ptr()))
for _ in range(3): step()
torch.cuda.synchronize()
t = time.time()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = (time.time() - t) / steps
print(f"B={B}: {dt*1e3:.3f} ms/step, {B/dt:.0f} frames/s")
stats = (_lib.KernelStat * 16)(); n = ctypes.c_int()
_lib.check(L.vbt_model_kernel_stats(it.handle, B, stats, 16, ctypes.byref(n)))
ms = (ctypes.c_double * 16)()
_lib.check(L.vbt_model_profile(it.handle, fd.data_ptr(), B, 5, st, ms, 16))
tot = sum(ms[i] for i in range(n.value))
for i in range(n.value):
    s = stats[i]
    gbs = s.algorithmic_bytes / (ms[i] * 1e-3) / 1e9 if ms[i] > 0 else 0
    print(f"  {s.name.decode():20s} launches {s.launches:3d}  {ms[i]:8.3f} ms  alg {s.algorithmic_bytes/1e6:9.1f} MB  {gbs:8.1f} GB/s  {s.macs/1e9:7.2f} GMAC")
print(f"  sum of event-bracketed families: {tot:.3f} ms")
