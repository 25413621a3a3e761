import sys, os, pickle
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from vbt_amd import synth
from vbt_amd.track import Pipeline
model = os.path.join(os.path.dirname(__file__), "..", "models", "efficientdet_lite0_synth.vbtm")
n, T = 6, 24
frames = np.stack([np.stack([synth.render(synth.background(40 + c), 9 * c + t) for c in range(n)]) for t in range(T)])
fd = torch.from_numpy(frames).to("cuda:0")
pipe = Pipeline(model, n, max_frames=T, fps=60.0)
st = torch.cuda.current_stream().cuda_stream
dets = []
for t in range(T):
    pipe.step(fd[t].data_ptr(), st)
    dets.append(pipe.detections())
pipe.finish(st)
pickle.dump({"dets": dets, "rows": [pipe.rows(c) for c in range(n)], "status": [pipe.tracker.status(c) for c in range(n)]}, open("gpurun_out/pipe_dump.pkl", "wb"))
print("ok")
