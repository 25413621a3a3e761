cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, time
x = torch.empty(64*320*320*3, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device='cuda')
s = torch.cuda.Stream()
for n in (1, 3):
    torch.cuda.synchronize(); t=time.perf_counter()
    with torch.cuda.stream(s):
        for i in range(20): d.copy_(x, non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print('H2D pinned 19.66 MB x20:', dt/20*1e3, 'ms each', x.numel()*20/dt/1e9, 'GB/s')
y = torch.empty(64*320*320*3, dtype=torch.uint8)
torch.cuda.synchronize(); t=time.perf_counter()
for i in range(5): d.copy_(y)
torch.cuda.synchronize(); dt=time.perf_counter()-t
print('H2D pageable:', y.numel()*5/dt/1e9, 'GB/s')
PY
python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['value_h2d_inclusive'], d['h2d_inclusive']['ms_per_step'], d['splits']['detect_only']['frames_per_s'])"
