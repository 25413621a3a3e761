"""Measured device copy / read bandwidth (the practical HBM ceiling next to the 8 TB/s nominal figure)."""
import time
import torch
for mb in (64, 256, 1024, 4096):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(1)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t = time.time()
    reps = 20
    for _ in range(reps): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.time() - t) / reps
    ai = a.view(torch.int32)
    for _ in range(3): ai.sum()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): ai.sum()
    torch.cuda.synchronize(); dr = (time.time() - t) / reps
    print(f"{mb:5d} MiB: copy {2*n/dt/1e12:.2f} TB/s (read+write), read-only sum {n/dr/1e12:.2f} TB/s")
