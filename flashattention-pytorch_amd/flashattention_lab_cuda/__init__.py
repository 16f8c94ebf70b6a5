"""`flashattention_lab_cuda` — the extension module the reference's wrappers look up by name
(`/root/reference/src/fa2/cuda/impl.py:10`), re-created as a thin ctypes shim over the C-ABI
library `libfa_mi355x.so` (declared in `include/fa_mi355x.h`, built from `csrc/`).

Exports exactly the six names of `/root/reference/csrc/common/torch.extension.cpp:73-83` with the
same positional signatures:

    fa1_forward / forward (q, k, v, causal, softmax_scale, br, bc)              -> (o, lse)
    fa1_backward / backward (q, k, v, o, do_, lse, causal, softmax_scale, br, bc) -> (dq, dk, dv)
    fa3_forward (q, k, v, causal, softmax_scale, br, bc, stages, fp8)           -> (o, lse)
    fa3_backward(q, k, v, o, do_, lse, causal, softmax_scale, br, bc, stages, fp8) -> (dq, dk, dv)

As in the reference (`csrc/fa2/fa2_fwd.cu:38-54`, `fa2_bwd.cu:112-115`): inputs are (BH, N, d) device
tensors, `o` and the gradients come back in the input dtype, `lse` is float32, nothing is recorded by
autograd, inputs are never modified, and shape errors raise RuntimeError.  PyTorch is used only to
allocate the outputs / workspace and to name the current HIP stream; all compute is in the HIP library.
There is NO CPU fallback: a missing library or a non-device tensor is an error.
"""
from __future__ import annotations

import ctypes
import os

import torch

from .workspace import WorkspaceCache, plan_backward_workspace

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfa_mi355x.so")

_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}

EXPORTED_C_SYMBOLS = (
    "fa1_forward", "fa1_backward", "fa2_forward", "fa2_backward", "fa3_forward", "fa3_backward",
    "fa_backward_workspace_bytes", "fa_backward_workspace_bytes_fast", "fa3_forward_workspace_bytes", "fa3_backward_workspace_bytes", "fa_last_error", "fa_version",
    "fa_set_kernel_mode", "fa_set_option", "fa_debug_trace_buffer", "fa_device_is_gfx950", "fa_profile_enable", "fa_profile_report",
    "fa_ex_forward", "fa_ex_backward", "fa_ex_backward_workspace_bytes", "fa_ex_backward_workspace_bytes_fast",
)


def _load_library() -> ctypes.CDLL:
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} not found: build it with `make -C flashattention-pytorch_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`)"
        )
    lib = ctypes.CDLL(_LIB_PATH)
    vp, i64, ci, dbl, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    fwd = [vp, vp, vp, vp, vp, i64, i64, i64, ci, ci, dbl, i64, i64, vp]
    bwd = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, ci, dbl, i64, i64, vp, sz, vp]
    for name in ("fa1_forward", "fa2_forward"):
        getattr(lib, name).argtypes = fwd
        getattr(lib, name).restype = ci
    for name in ("fa1_backward", "fa2_backward"):
        getattr(lib, name).argtypes = bwd
        getattr(lib, name).restype = ci
    lib.fa3_forward.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, ci, ci, dbl, i64, i64, i64, ci, vp, sz, vp]
    lib.fa3_forward.restype = ci
    lib.fa3_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, ci, dbl, i64, i64, i64, ci, vp, sz, vp]
    lib.fa3_backward.restype = ci
    lib.fa_backward_workspace_bytes.argtypes = [i64, i64, i64, ci]
    lib.fa_backward_workspace_bytes.restype = sz
    lib.fa_backward_workspace_bytes_fast.argtypes = [i64, i64, i64, ci, ci]
    lib.fa_backward_workspace_bytes_fast.restype = sz
    lib.fa3_forward_workspace_bytes.argtypes = [i64, i64, i64, ci, ci]
    lib.fa3_forward_workspace_bytes.restype = sz
    lib.fa3_backward_workspace_bytes.argtypes = [i64, i64, i64, ci, ci]
    lib.fa3_backward_workspace_bytes.restype = sz
    lib.fa_last_error.restype = ctypes.c_char_p
    lib.fa_version.restype = ctypes.c_char_p
    lib.fa_set_kernel_mode.argtypes = [ci]
    lib.fa_set_kernel_mode.restype = ci
    lib.fa_debug_trace_buffer.argtypes = [vp]
    lib.fa_debug_trace_buffer.restype = ci
    lib.fa_set_option.argtypes = [ctypes.c_char_p, ci]
    lib.fa_set_option.restype = ci
    lib.fa_device_is_gfx950.argtypes = [ci]
    lib.fa_device_is_gfx950.restype = ci
    lib.fa_profile_enable.argtypes = [ci]
    lib.fa_profile_enable.restype = ci
    lib.fa_profile_report.argtypes = [ctypes.c_char_p, sz]
    lib.fa_profile_report.restype = ci
    u64 = ctypes.c_uint64
    lib.fa_ex_forward.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, ci, ci, dbl, vp, i64, vp, i64, i64, dbl, u64, vp]
    lib.fa_ex_forward.restype = ci
    lib.fa_ex_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, ci, ci, dbl, vp, i64, vp, i64, i64, dbl, u64,
                                   vp, sz, vp]
    lib.fa_ex_backward.restype = ci
    lib.fa_ex_backward_workspace_bytes.argtypes = [i64, i64, i64, i64, ci]
    lib.fa_ex_backward_workspace_bytes.restype = sz
    lib.fa_ex_backward_workspace_bytes_fast.argtypes = [i64, i64, i64, i64, ci, ci, ci]
    lib.fa_ex_backward_workspace_bytes_fast.restype = sz
    return lib


_lib = _load_library()
LIBRARY_PATH = _LIB_PATH


def version() -> str:
    return _lib.fa_version().decode()


def set_kernel_mode(mode: int) -> int:
    """0 = auto (16-bit MFMA kernels where they apply), 1 = force the exact-f32 kernels. Returns the old mode."""
    return _lib.fa_set_kernel_mode(int(mode))


def set_option(name: str, value: int) -> None:
    """Tuning knob for sweeps / A-B runs (fwd_kb, fwd_stag, fwd_tpw, dq_tpw, dkdv_tpw, ...); see csrc/fa_kernels.h."""
    _check(_lib.fa_set_option(name.encode(), int(value)))


def debug_trace_buffer(t) -> None:
    """Debug: register a CUDA int64 tensor of >= 4096 elements for the staggered forward's phase timestamps (None = off)."""
    _check(_lib.fa_debug_trace_buffer(ctypes.c_void_p(t.data_ptr() if t is not None else 0)))


def profile_enable(on: bool) -> None:
    """Start (and clear) or stop per-kernel HIP-event timing inside the library."""
    _lib.fa_profile_enable(int(bool(on)))


def profile_report() -> dict:
    """{kernel_name: (launches, total_ms)} for the launches since profile_enable(True); waits for the events."""
    buf = ctypes.create_string_buffer(4096)
    n = _lib.fa_profile_report(buf, 4096)
    if n < 0:
        raise RuntimeError(_lib.fa_last_error().decode())
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(_lib.fa_last_error().decode())


def _check_inputs(who, *tensors):
    t0 = tensors[0]
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(f"{who}: tensors must be on the GPU (HIP device); there is no CPU path")
    if t0.dim() != 3:
        raise RuntimeError(f"{who}: q must be 3-D (BH, N, d), got {tuple(t0.shape)}")  # fa2_fwd.cu:40
    for t in tensors:
        if t.shape != t0.shape or t.dtype != t0.dtype or t.device != t0.device:
            raise RuntimeError(f"{who}: q, k, v (o, do) must share shape, dtype and device")  # fa2_fwd.cu:41-45
    if t0.dtype not in _DTYPE_CODE:
        raise RuntimeError(f"{who}: unsupported dtype {t0.dtype}")
    return _DTYPE_CODE[t0.dtype]


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ---- workspace: one grow-only buffer per (device, stream), see workspace.py ----
_workspaces = WorkspaceCache(lambda nbytes, device: torch.empty((nbytes,), dtype=torch.uint8, device=device))


def _workspace(device, nbytes: int):
    """The buffer a call on `device`'s current stream works in.  Inside a graph capture it is a plain allocation of the
    capture's own pool (a cached buffer allocated during capture would outlive the pool it came from)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)
    return _workspaces.get(device, _stream_ptr(device), nbytes)


def _device_headroom(device) -> int:
    """Bytes the device could still give this process: driver-free memory + what torch's allocator holds unused."""
    free, _total = torch.cuda.mem_get_info(device)
    return int(free) + max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))


def release_workspace(device=None) -> int:
    """Give the shim's workspace buffers back to torch's allocator (all devices if None).  Returns the bytes released.
    Call it between phases of a program that no longer runs attention backward; the next call allocates again."""
    if device is not None:
        device = torch.device(device)
        if device.index is None:
            device = torch.device(device.type, torch.cuda.current_device())
    return _workspaces.release(device)


def workspace_stats() -> dict:
    return {"bytes": _workspaces.total_bytes(), "allocations": _workspaces.allocations, "hits": _workspaces.hits}


def _forward(cfn, who, q, k, v, causal, softmax_scale, br, bc, extra=None):
    code = _check_inputs(who, q, k, v)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    bh, n, d = q.shape
    with torch.cuda.device(q.device):
        o = torch.empty_like(q)
        lse = torch.empty((bh, n), dtype=torch.float32, device=q.device)
        args = [q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), bh, n, d, code,
                int(bool(causal)), float(softmax_scale), int(br), int(bc)]
        ws = None
        if extra is not None:
            stages, fp8 = extra
            nbytes = _lib.fa3_forward_workspace_bytes(bh, n, d, code, int(bool(fp8)))
            ws = _workspace(q.device, nbytes)
            args += [int(stages), int(bool(fp8)), ws.data_ptr(), int(nbytes)]
        args.append(_stream_ptr(q.device))
        _check(cfn(*args))
    return o, lse


def _backward(cfn, who, q, k, v, o, do_, lse, causal, softmax_scale, br, bc, extra=None):
    code = _check_inputs(who, q, k, v, o, do_)
    q, k, v, o, do_ = (t.contiguous() for t in (q, k, v, o, do_))
    bh, n, d = q.shape
    if lse.shape != (bh, n) or lse.dtype != torch.float32 or not lse.is_cuda:
        raise RuntimeError(f"{who}: lse must be a float32 device tensor of shape (BH, N)")
    lse = lse.contiguous()
    with torch.cuda.device(q.device):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        # minimum = row constants (+ FA3's round-tripped Q, K); fast = + room for the dS tiles where the hand-over serves the
        # call (bounded by the library's chunk size whatever BH is).  The size only selects speed: with the minimum the
        # library runs its recomputing dQ pass.
        small = int(_lib.fa_backward_workspace_bytes(bh, n, d, code))
        fast = int(_lib.fa_backward_workspace_bytes_fast(bh, n, d, code, int(bool(causal))))
        if extra is not None:
            fp8_slabs = int(_lib.fa3_backward_workspace_bytes(bh, n, d, code, int(bool(extra[1])))) - small
            small, fast = small + fp8_slabs, fast + fp8_slabs
        capturing = torch.cuda.is_current_stream_capturing()
        have = 0 if capturing else _workspaces.capacity(q.device, _stream_ptr(q.device))
        # (no device query while a graph is being captured: the buffer then comes from the capture's pool anyway)
        nbytes = plan_backward_workspace(small, fast, have, None if (have >= fast or capturing) else _device_headroom(q.device))
        ws = _workspace(q.device, nbytes)
        nbytes = max(nbytes, 0 if capturing else _workspaces.capacity(q.device, _stream_ptr(q.device)))
        args = [q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do_.data_ptr(), lse.data_ptr(),
                dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), bh, n, d, code, int(bool(causal)),
                float(softmax_scale), int(br), int(bc)]
        if extra is not None:
            stages, fp8 = extra
            args += [int(stages), int(bool(fp8))]
        args += [ws.data_ptr(), nbytes, _stream_ptr(q.device)]
        _check(cfn(*args))
    return dq, dk, dv


# ---- the six names of csrc/common/torch.extension.cpp:73-83 ----

def fa1_forward(q, k, v, causal, softmax_scale, br, bc):
    return _forward(_lib.fa1_forward, "fa1_forward", q, k, v, causal, softmax_scale, br, bc)


def fa1_backward(q, k, v, o, do_, lse, causal, softmax_scale, br, bc):
    return _backward(_lib.fa1_backward, "fa1_backward", q, k, v, o, do_, lse, causal, softmax_scale, br, bc)


def forward(q, k, v, causal, softmax_scale, br, bc):
    return _forward(_lib.fa2_forward, "forward", q, k, v, causal, softmax_scale, br, bc)


def backward(q, k, v, o, do_, lse, causal, softmax_scale, br, bc):
    return _backward(_lib.fa2_backward, "backward", q, k, v, o, do_, lse, causal, softmax_scale, br, bc)


def fa3_forward(q, k, v, causal, softmax_scale, br, bc, stages, fp8):
    return _forward(_lib.fa3_forward, "fa3_forward", q, k, v, causal, softmax_scale, br, bc, extra=(stages, fp8))


def fa3_backward(q, k, v, o, do_, lse, causal, softmax_scale, br, bc, stages, fp8):
    return _backward(_lib.fa3_backward, "fa3_backward", q, k, v, o, do_, lse, causal, softmax_scale, br, bc,
                     extra=(stages, fp8))


# ---- extended attention (include/fa_mi355x.h: fa_ex_forward / fa_ex_backward) ----

def _ex_common(who, q, k, v, mask, block_mask, br, bc):
    for t in (q, k, v):
        if not t.is_cuda:
            raise RuntimeError(f"{who}: tensors must be on the GPU (HIP device); there is no CPU path")
    if q.dim() != 3 or k.dim() != 3 or v.shape != k.shape or q.shape[0] != k.shape[0] or q.shape[2] != k.shape[2]:
        raise RuntimeError(f"{who}: q must be (BH, Nq, d), k and v (BH, Nk, d); got {tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
    if q.dtype not in _DTYPE_CODE or k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError(f"{who}: q, k, v must share a supported dtype")
    bh, nq, d = q.shape
    nk = k.shape[1]
    mptr, mstride = 0, 0
    if mask is not None:
        mask = mask.to(device=q.device, dtype=torch.uint8).contiguous()   # 0 = masked
        if tuple(mask.shape) == (nq, nk):
            mstride = 0
        elif tuple(mask.shape) == (bh, nq, nk):
            mstride = nq * nk
        else:
            raise RuntimeError(f"{who}: mask must be (Nq, Nk) or (BH, Nq, Nk), got {tuple(mask.shape)}")
        mptr = mask.data_ptr()
    bptr = 0
    if block_mask is not None:
        block_mask = block_mask.to(device=q.device, dtype=torch.uint8).contiguous()
        br, bc = max(int(br), 1), max(int(bc), 1)   # (an empty side: Br = min(block_size, 0))
        want = ((nq + br - 1) // br, (nk + bc - 1) // bc)
        if tuple(block_mask.shape) != want:
            raise RuntimeError(f"{who}: block_sparse_mask must be {want} for br={br}, bc={bc}, got {tuple(block_mask.shape)}")
        bptr = block_mask.data_ptr()
    return bh, nq, nk, d, _DTYPE_CODE[q.dtype], mask, mptr, mstride, block_mask, bptr


def ex_forward(q, k, v, causal, softmax_scale, mask=None, block_mask=None, br=128, bc=128, dropout_p=0.0, seed=0):
    """(o, lse) of attention with Nq != Nk (causal aligned bottom-right), dense mask (0 = masked), block-sparse mask
    (0 = tile skipped) and dropout; see include/fa_mi355x.h."""
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    bh, nq, nk, d, code, mask, mptr, mstride, block_mask, bptr = _ex_common("ex_forward", q, k, v, mask, block_mask, br, bc)
    with torch.cuda.device(q.device):
        o = torch.empty_like(q)
        lse = torch.empty((bh, nq), dtype=torch.float32, device=q.device)
        _check(_lib.fa_ex_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), bh, nq, nk, d, code,
                                  int(bool(causal)), float(softmax_scale), mptr, mstride, bptr, int(br), int(bc), float(dropout_p),
                                  int(seed) & (2 ** 64 - 1), _stream_ptr(q.device)))
    return o, lse


def ex_backward(q, k, v, o, do_, lse, causal, softmax_scale, mask=None, block_mask=None, br=128, bc=128, dropout_p=0.0, seed=0):
    q, k, v, o, do_, lse = (t.contiguous() for t in (q, k, v, o, do_, lse))
    bh, nq, nk, d, code, mask, mptr, mstride, block_mask, bptr = _ex_common("ex_backward", q, k, v, mask, block_mask, br, bc)
    if o.shape != q.shape or do_.shape != q.shape or lse.shape != (bh, nq) or lse.dtype != torch.float32:
        raise RuntimeError("ex_backward: o, do must be (BH, Nq, d) and lse (BH, Nq) float32")
    with torch.cuda.device(q.device):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        small = int(_lib.fa_ex_backward_workspace_bytes(bh, nq, nk, d, code))
        extras = int(mask is not None or block_mask is not None or dropout_p > 0.0)
        fast = int(_lib.fa_ex_backward_workspace_bytes_fast(bh, nq, nk, d, code, int(bool(causal)), extras))
        capturing = torch.cuda.is_current_stream_capturing()
        have = 0 if capturing else _workspaces.capacity(q.device, _stream_ptr(q.device))
        nbytes = plan_backward_workspace(small, fast, have, None if (have >= fast or capturing) else _device_headroom(q.device))
        ws = _workspace(q.device, nbytes)
        nbytes = max(nbytes, 0 if capturing else _workspaces.capacity(q.device, _stream_ptr(q.device)))
        _check(_lib.fa_ex_backward(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do_.data_ptr(), lse.data_ptr(),
                                   dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), bh, nq, nk, d, code, int(bool(causal)),
                                   float(softmax_scale), mptr, mstride, bptr, int(br), int(bc), float(dropout_p),
                                   int(seed) & (2 ** 64 - 1), ws.data_ptr(), nbytes, _stream_ptr(q.device)))
    return dq, dk, dv
