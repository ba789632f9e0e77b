"""D2H copy bandwidth from pinned memory: one stream vs the copy split over two / four streams (development)."""
import time
import torch
dev = torch.device("cuda", 0)
n = 64 << 20
src = torch.empty(4 * n, dtype=torch.uint8, device=dev).fill_(7)
dst = torch.empty(4 * n, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream() for _ in range(4)]
def run(k, reps=20):
    part = 4 * n // k
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                dst[i * part:(i + 1) * part].copy_(src[i * part:(i + 1) * part], non_blocking=True)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("D2H %d MB in %d piece(s) on %d stream(s): %.1f GB/s" % (4 * n >> 20, k, k, 4 * n * reps / dt / 1e9))
for k in (1, 2, 4):
    run(k)
# and with an H2D copy running against it
h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device=dev)
def duplex(k, reps=20):
    part = 4 * n // k
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(streams[3]):
            for _ in range(2):
                d.copy_(h, non_blocking=True)
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                dst[i * part:(i + 1) * part].copy_(src[i * part:(i + 1) * part], non_blocking=True)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("D2H in %d piece(s) with 128 MB of H2D alongside: %.1f GB/s D2H" % (k, 4 * n * reps / dt / 1e9))
duplex(1); duplex(2)
