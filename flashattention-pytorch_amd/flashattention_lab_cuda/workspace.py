"""Grow-only workspace buffers of the extension shim, one per (device, stream).

The reference's backward allocates its scratch on every call (`/root/reference/csrc/fa2/fa2_bwd.cu:53-57`: three fp32
tensors through ATen's allocator).  Here the scratch is the C library's caller-owned workspace (include/fa_mi355x.h), and a
per-call `torch.empty` of it puts the caching allocator into every backward: it may split a cached block for another tensor and
send the workspace request to the device allocator inside a training step, a failed multi-GB request makes torch synchronise the
device and empty its cache before the retry, and a transient multi-GB block is what fragments a tight pool (DESIGN.md section 0:
round 2's driver-timed step spent 11.85 of its 19.28 ms on the host side of exactly this path).  So the shim keeps ONE buffer per
(device, stream), sized by the largest request so far, handed to every call on that stream (calls on one stream are ordered, so
they can share it; calls on different streams get different buffers), and released only by `release()`.

While a stream is being captured into a graph the cache is bypassed: the buffer then has to belong to the graph's own pool.

Nothing here computes; the allocator is injectable so the grow / reuse / release logic runs on a CPU in the tests.
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, Optional, Tuple


class WorkspaceCache:
    def __init__(self, alloc: Callable[[int, object], object], granule: int = 1 << 20):
        """alloc(nbytes, device) -> an object that owns nbytes of device memory and has .data_ptr()."""
        self._alloc = alloc
        self._granule = int(granule)
        self._bufs: Dict[Tuple[object, int], Tuple[int, object]] = {}
        self._lock = threading.Lock()
        self.allocations = 0   # times a buffer was (re)allocated
        self.hits = 0          # calls served from an existing buffer

    def get(self, device, stream: int, nbytes: int):
        """A buffer of >= nbytes for calls enqueued on `stream` of `device` (never smaller than the previous one)."""
        nbytes = max(int(nbytes), 1)
        key = (device, int(stream))
        with self._lock:
            have = self._bufs.get(key)
            if have is not None and have[0] >= nbytes:
                self.hits += 1
                return have[1]
            size = (nbytes + self._granule - 1) // self._granule * self._granule
            # drop the old buffer first: kernels already enqueued on this stream keep it alive through the allocator's
            # stream ordering (torch frees are stream-ordered on the allocating stream), and two live copies would double the peak
            self._bufs.pop(key, None)
            buf = self._alloc(size, device)
            self._bufs[key] = (size, buf)
            self.allocations += 1
            return buf

    def capacity(self, device, stream: int) -> int:
        with self._lock:
            have = self._bufs.get((device, int(stream)))
            return have[0] if have is not None else 0

    def total_bytes(self) -> int:
        with self._lock:
            return sum(size for size, _ in self._bufs.values())

    def release(self, device=None) -> int:
        """Drop the buffers of one device (all devices if None); returns the bytes given back to the allocator."""
        with self._lock:
            keys = [k for k in self._bufs if device is None or k[0] == device]
            freed = sum(self._bufs[k][0] for k in keys)
            for k in keys:
                del self._bufs[k]
            return freed


def plan_backward_workspace(minimum: int, fast: int, have: int, free_bytes: Optional[int]) -> int:
    """How many bytes to hand the backward: `fast` (room for the dS tiles) when it is already there or can be had without
    pressing the device, else `minimum` (the library then takes its recomputing dQ pass).  Decided BEFORE allocating: an
    allocation that fails inside torch first synchronises the device and empties the allocator's cache.

    have       -- bytes of the buffer this (device, stream) already owns
    free_bytes -- what the device could still give (driver-free + the allocator's cached-but-unused bytes); None = unknown
    """
    if fast <= minimum or have >= fast:
        return max(fast, minimum)
    if free_bytes is None:
        return fast
    grow = fast - have
    # keep a quarter of what is free for the caller's own tensors
    return fast if grow <= (free_bytes * 3) // 4 else minimum
