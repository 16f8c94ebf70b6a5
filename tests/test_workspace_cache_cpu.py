"""CPU: the shim's grow-only workspace cache (flashattention_lab_cuda/workspace.py) — grow / reuse / release logic with an
injected allocator, and the decision how much workspace a backward call gets.  Replaces the per-call `torch.empty` of the
backward workspace (the reference allocates its scratch per call: /root/reference/csrc/fa2/fa2_bwd.cu:53-57)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location(
    "fa_workspace", os.path.join(ROOT, "flashattention-pytorch_amd", "flashattention_lab_cuda", "workspace.py"))
ws = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ws)   # the module has no dependency on the HIP library: loadable on any machine


class Buf:
    live = 0

    def __init__(self, n, dev):
        self.n, self.dev = n, dev
        Buf.live += n

    def __del__(self):
        Buf.live -= self.n


def make():
    Buf.live = 0
    return ws.WorkspaceCache(lambda n, d: Buf(n, d), granule=1024)


def test_reuse_and_grow_only():
    c = make()
    a = c.get("dev0", 7, 1000)
    assert a.n == 1024 and c.allocations == 1 and c.capacity("dev0", 7) == 1024
    assert c.get("dev0", 7, 10) is a and c.get("dev0", 7, 1024) is a and c.hits == 2 and c.allocations == 1
    b = c.get("dev0", 7, 5000)       # grows: rounded up to the granule, the old buffer is dropped
    assert b is not a and b.n == 5120 and c.allocations == 2
    del a
    assert Buf.live == 5120
    assert c.get("dev0", 7, 1) is b  # never shrinks
    assert c.total_bytes() == 5120


def test_one_buffer_per_device_and_stream():
    c = make()
    a, b, d = c.get("dev0", 1, 100), c.get("dev0", 2, 100), c.get("dev1", 1, 100)
    assert a is not b and a is not d and b is not d and c.allocations == 3
    assert c.get("dev0", 1, 50) is a and c.get("dev0", 2, 50) is b and c.get("dev1", 1, 50) is d
    assert c.total_bytes() == 3 * 1024


def test_release():
    c = make()
    c.get("dev0", 1, 100); c.get("dev0", 2, 3000); c.get("dev1", 1, 100)
    assert c.release("dev0") == 1024 + 3072 and c.total_bytes() == 1024 and c.capacity("dev0", 1) == 0
    assert Buf.live == 1024
    assert c.release() == 1024 and c.total_bytes() == 0 and Buf.live == 0
    assert c.get("dev0", 1, 100).n == 1024 and c.allocations == 4   # allocates again after a release


def test_zero_byte_request_still_yields_a_pointer():
    c = make()
    assert c.get("dev0", 0, 0).n == 1024


def test_backward_workspace_plan():
    plan = ws.plan_backward_workspace
    assert plan(8, 8, 0, None) == 8                      # no hand-over for this call
    assert plan(8, 1000, 2000, 0) == 1000                # already owned: no question asked
    assert plan(8, 1000, 0, None) == 1000                # unknown headroom: ask for it
    assert plan(8, 1000, 0, 10_000) == 1000              # fits comfortably
    assert plan(8, 1000, 0, 1200) == 8                   # would take more than 3/4 of what is free: recomputing pass
    assert plan(8, 1000, 600, 600) == 1000               # only the growth counts
    assert plan(8, 1000, 0, 0) == 8
