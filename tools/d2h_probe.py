"""How long the 1080p fp64 frame takes to reach host memory: pageable (fresh / reused) vs pinned."""
import time, torch, numpy as np
n = 1920 * 1080 * 3
dev = torch.zeros(n, dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
def clock(f, k=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
print("fresh pageable  %.2f ms" % clock(lambda: dev.cpu()))
host = torch.empty(n, dtype=torch.float64)
print("reused pageable %.2f ms" % clock(lambda: host.copy_(dev)))
pin = torch.empty(n, dtype=torch.float64).pin_memory()
print("pinned          %.2f ms" % clock(lambda: pin.copy_(dev)))
t0 = time.perf_counter(); p2 = torch.empty(n, dtype=torch.float64).pin_memory(); print("pin_memory() of 50 MB %.2f ms" % ((time.perf_counter() - t0) * 1e3))
a = np.empty(n); t0 = time.perf_counter(); a[:] = pin.numpy(); print("memcpy pinned -> fresh numpy %.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); a[:] = pin.numpy(); print("memcpy pinned -> touched numpy %.2f ms" % ((time.perf_counter() - t0) * 1e3))
