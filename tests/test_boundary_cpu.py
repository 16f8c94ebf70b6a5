"""CPU: the drop-in boundary — the C-ABI library loads and exports every symbol include/fa_mi355x.h
declares, and the host-side wrappers keep the reference's names, argument meaning and error behaviour
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "fa_mi355x.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fa[0-9_a-z]*)\s*\(", src)))


def test_header_declares_the_six_reference_entry_points():
    names = _declared_symbols()
    for n in ("fa1_forward", "fa1_backward", "fa2_forward", "fa2_backward", "fa3_forward", "fa3_backward"):
        assert n in names


def test_library_exports_every_declared_symbol():
    import flashattention_lab_cuda as ext

    lib = ctypes.CDLL(ext.LIBRARY_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in fa_mi355x.h but not exported"
    assert set(ext.EXPORTED_C_SYMBOLS) == set(_declared_symbols())
    assert "gfx950" in ext.version()


def test_shim_exports_reference_python_names():
    # csrc/common/torch.extension.cpp:73-83
    import flashattention_lab_cuda as ext

    for n in ("fa1_forward", "fa1_backward", "forward", "backward", "fa3_forward", "fa3_backward"):
        assert callable(getattr(ext, n))


def test_workspace_query_and_argument_validation_without_gpu():
    import flashattention_lab_cuda as ext

    lib = ext._lib
    assert lib.fa_backward_workspace_bytes(4, 128, 64, 2) >= 4 * 128 * 4
    assert lib.fa_backward_workspace_bytes(0, 0, 64, 2) > 0
    # the dS hand-over's size: N * N * 2 bytes per (b,h) on top of the minimum where it serves the call (d = 128, 16-bit, no
    # mask, launches of > 256 row tiles), at most 4 GiB (equal chunks of (b,h) units); the minimum everywhere else
    base = lib.fa_backward_workspace_bytes(256, 4096, 128, 2)
    assert lib.fa_backward_workspace_bytes_fast(256, 4096, 128, 2, 0) == base + 128 * 4096 * 4096 * 2   # two chunks of 128 units
    # causal (round 3): the hand-over too, while a chunk holds 16 units (or the whole launch), else the recomputing backward
    assert lib.fa_backward_workspace_bytes_fast(256, 4096, 128, 2, 1) == base + 128 * 4096 * 4096 * 2
    assert lib.fa_backward_workspace_bytes_fast(128, 8192, 128, 2, 1) == lib.fa_backward_workspace_bytes(128, 8192, 128, 2) + 32 * 8192 * 8192 * 2
    assert lib.fa_backward_workspace_bytes_fast(32, 16384, 128, 2, 1) == lib.fa_backward_workspace_bytes(32, 16384, 128, 2)    # 8 units per chunk: O(BH N) bytes
    assert lib.fa_backward_workspace_bytes_fast(8, 16384, 128, 2, 1) == lib.fa_backward_workspace_bytes(8, 16384, 128, 2) + 8 * 16384 * 16384 * 2
    assert lib.fa_backward_workspace_bytes_fast(512, 1024, 128, 2, 1) == lib.fa_backward_workspace_bytes(512, 1024, 128, 2) + 512 * 1024 * 1024 * 2   # short rows too
    assert lib.fa_backward_workspace_bytes_fast(32, 1024, 128, 2, 1) == lib.fa_backward_workspace_bytes(32, 1024, 128, 2)   # short rows, < 160 row tiles: the small-launch kernels
    assert lib.fa_backward_workspace_bytes_fast(2, 4096, 128, 2, 1) == lib.fa_backward_workspace_bytes(2, 4096, 128, 2) + 2 * 4096 * 4096 * 2   # rows of 4096: always
    assert lib.fa_backward_workspace_bytes_fast(32, 4096, 128, 2, 0) == lib.fa_backward_workspace_bytes(32, 4096, 128, 2) + 32 * 4096 * 4096 * 2
    assert lib.fa_backward_workspace_bytes_fast(2048, 4096, 128, 1, 0) == lib.fa_backward_workspace_bytes(2048, 4096, 128, 1) + 128 * 4096 * 4096 * 2
    assert lib.fa_backward_workspace_bytes_fast(300, 4096, 128, 1, 0) == lib.fa_backward_workspace_bytes(300, 4096, 128, 1) + 100 * 4096 * 4096 * 2   # 3 x 100, not 128 + 128 + 44
    assert lib.fa_backward_workspace_bytes_fast(64, 32768, 128, 2, 0) == lib.fa_backward_workspace_bytes(64, 32768, 128, 2) + 2 * 32768 * 32768 * 2   # 2 units of 2 GiB per chunk
    assert lib.fa_backward_workspace_bytes_fast(64, 65536, 128, 2, 0) == lib.fa_backward_workspace_bytes(64, 65536, 128, 2)   # one unit alone is over the bound
    for bh_, n_ in ((1, 4096), (256, 4096), (4096, 4096), (100, 16384), (7, 40000), (3, 100000)):    # bounded whatever BH and N are
        assert lib.fa_backward_workspace_bytes_fast(bh_, n_, 128, 2, 0) - lib.fa_backward_workspace_bytes(bh_, n_, 128, 2) <= 4 << 30
    assert lib.fa_backward_workspace_bytes_fast(80, 1000, 128, 2, 0) == lib.fa_backward_workspace_bytes(80, 1000, 128, 2) + 80 * 32 * 32 * 2048  # ragged N: whole tiles
    assert lib.fa_backward_workspace_bytes_fast(2, 512, 128, 2, 0) == lib.fa_backward_workspace_bytes(2, 512, 128, 2) + 2 * 16 * 16 * 2048   # small launches too (round 3)
    assert lib.fa_backward_workspace_bytes_fast(256, 4096, 64, 2, 0) == lib.fa_backward_workspace_bytes(256, 4096, 64, 2)  # d = 64
    assert lib.fa_backward_workspace_bytes_fast(256, 4096, 128, 0, 0) == lib.fa_backward_workspace_bytes(256, 4096, 128, 0)  # fp32
    # validation happens before any HIP call: bad dtype / head_dim > 256 / null pointers
    rc = lib.fa2_forward(None, None, None, None, None, 1, 8, 64, 7, 0, 0.125, 128, 128, None)
    assert rc == -1 and b"dtype" in lib.fa_last_error()
    rc = lib.fa2_forward(None, None, None, None, None, 1, 8, 512, 2, 0, 0.125, 128, 128, None)
    assert rc == -2 and b"head_dim" in lib.fa_last_error()
    rc = lib.fa2_forward(None, None, None, None, None, 1, 8, 64, 2, 0, 0.125, 128, 128, None)
    assert rc == -1 and b"null" in lib.fa_last_error()
    rc = lib.fa2_forward(None, None, None, None, None, 0, 8, 64, 2, 0, 0.125, 128, 128, None)
    assert rc == 0  # empty problem is a no-op
    assert lib.fa_set_kernel_mode(5) == -1
    assert lib.fa_set_kernel_mode(2) == 0 and lib.fa_set_kernel_mode(0) == 2


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_wrappers_keep_reference_names_and_errors(algo):
    op = __import__(f"fa{algo}.op", fromlist=["x"])
    impl = __import__(f"fa{algo}.cuda.impl", fromlist=["x"])
    spec = __import__(f"fa{algo}.spec", fromlist=["x"])
    attn = getattr(op, f"fa{algo}_attention")
    pick = getattr(spec, f"pick_fa{algo}_spec")
    # spec table: src/fa2/spec.py:9-12, src/fa3/spec.py:10-13
    assert (pick(64).br, pick(64).bc, pick(64).num_warps) == (128, 128, 8)
    assert (pick(128).br, pick(128).bc) == (64, 128)
    if algo == 3:
        assert pick(64).stages == 2
    with pytest.raises(Exception):  # frozen dataclass
        pick(64).br = 1
    q = torch.randn(2, 3, 8, 16)
    with pytest.raises(ValueError):
        attn(q, q, q, backend="nope")
    with pytest.raises(NotImplementedError):
        attn(q, q, q, backend="torch")
    with pytest.raises(NotImplementedError):
        attn(q, q, q, backend="triton")
    with pytest.raises(RuntimeError, match="CUDA tensors"):  # src/fa2/cuda/impl.py:43-44; auto never falls back
        attn(q, q, q, backend="auto")
    assert impl._load_ext().__name__ == "flashattention_lab_cuda"
    qb, shp = impl._merge_bh(q)
    assert qb.shape == (6, 8, 16) and shp == (2, 3)
    assert impl._merge_bh(qb) == (qb, None) or impl._merge_bh(qb)[1] is None
    assert impl._split_bh(qb, shp).shape == q.shape
    assert impl._split_bh_lse(torch.zeros(6, 8), shp).shape == (2, 3, 8)


def test_shim_rejects_cpu_tensors_loudly():
    import flashattention_lab_cuda as ext

    q = torch.randn(2, 8, 16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.forward(q, q, q, False, 0.25, 128, 128)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "flashattention-pytorch_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
