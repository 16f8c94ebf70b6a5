"""What does a large device allocation cost (the round-2 bench discrepancy)?  Times torch.empty of large
buffers that go to hipMalloc — from an empty cache (memory this process had before), and of memory this process has NOT had
before (held simultaneously) — plus the first touch of each.  Run it FIRST on a fresh box."""
import time

import torch

t0 = time.perf_counter()
torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
print(f"context up after {time.perf_counter() - t0:.2f} s")


def alloc(gb, label):
    n0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = torch.empty(int(gb * 2 ** 30), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x.fill_(1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    x.fill_(2)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{label}: torch.empty of {gb} GiB {1e3 * (t1 - t0):8.1f} ms ({torch.cuda.memory_stats()['num_device_alloc'] - n0} device allocation), "
          f"first fill {1e3 * (t2 - t1):7.1f} ms, second fill {1e3 * (t3 - t2):6.1f} ms")
    return x


held = []
for i in range(6):
    held.append(alloc(8.0, f"new memory #{i} (earlier ones still held)"))
del held
torch.cuda.empty_cache()
for i in range(2):
    x = alloc(8.0, "after empty_cache (memory this process had before)")
    del x
    torch.cuda.empty_cache()
big = alloc(100.0, "100 GiB, mostly new memory")
