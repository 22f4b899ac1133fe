"""PCIe yardstick for the host-path rows: pinned 32 MB buffers, H2D alone, D2H alone, both at once on two streams, and both as 8 x 4 MB pieces (what the
library's piece pipeline issues).  python tools/mb_link.py"""
import torch, time, ctypes
n = 32 << 20
hp = torch.empty(n, dtype=torch.uint8, pin_memory=True); hq = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1): d1.copy_(hp, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2): hq.copy_(d2, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return n / dt / 1e9
for _ in range(2):
    print("H2D alone %.1f GB/s   D2H alone %.1f GB/s   both: %.1f GB/s each direction" % (run(1, 0), run(0, 1), run(1, 1)), flush=True)
# small pieces: 4 MB copies back to back
m = 4 << 20
def run_small(reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for k in range(8):
            with torch.cuda.stream(s1): d1[k*m:(k+1)*m].copy_(hp[k*m:(k+1)*m], non_blocking=True)
            with torch.cuda.stream(s2): hq[k*m:(k+1)*m].copy_(d2[k*m:(k+1)*m], non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return n / dt / 1e9
print("both directions as 8 x 4 MB pieces: %.1f GB/s each direction" % run_small(), flush=True)
