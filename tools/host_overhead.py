"""Host-side cost of the shim per call (small launches are host bound): microseconds per forward / backward enqueue at a
shape whose kernels take a few microseconds, and of the Python primitives the shim uses."""
import sys
import time
import timeit

sys.path.insert(0, "flashattention-pytorch_amd")
import torch
import flashattention_lab_cuda as ext

dev = torch.device("cuda", 0)
q, k, v, do = (torch.randn((8, 512, 128), device=dev, dtype=torch.bfloat16) for _ in range(4))
o, lse = ext.forward(q, k, v, False, 0.088, 64, 128)
ext.backward(q, k, v, o, do, lse, False, 0.088, 64, 128)
torch.cuda.synchronize()


def rate(fn, n=3000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n


print("forward  enqueue / wall per call (us): %.1f / %.1f" % rate(lambda: ext.forward(q, k, v, False, 0.088, 64, 128)))
print("backward enqueue / wall per call (us): %.1f / %.1f" % rate(lambda: ext.backward(q, k, v, o, do, lse, False, 0.088, 64, 128)))
for name, stmt in (("torch.cuda.is_current_stream_capturing()", lambda: torch.cuda.is_current_stream_capturing()),
                   ("torch.cuda.current_stream(dev).cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream),
                   ("torch.cuda.device(dev) context", lambda: torch.cuda.device(dev).__enter__()),
                   ("torch.empty_like(q)", lambda: torch.empty_like(q)),
                   ("fa_backward_workspace_bytes_fast (ctypes)", lambda: ext._lib.fa_backward_workspace_bytes_fast(8, 512, 128, 2, 0)),
                   ("torch.cuda.mem_get_info(dev)", lambda: torch.cuda.mem_get_info(dev)),
                   ("q.contiguous()", lambda: q.contiguous()),
                   ("q.data_ptr()", lambda: q.data_ptr())):
    n = 2000
    t = timeit.timeit(stmt, number=n)
    print(f"   {name:50s} {1e6 * t / n:7.2f} us")
