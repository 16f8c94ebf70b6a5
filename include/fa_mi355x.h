/*
 * fa_mi355x.h — C ABI of the MI355X (gfx950) FlashAttention forward/backward library.
 *
 * This is the drop-in boundary for the hot path of PeTeRr0/FlashAttention-pytorch.
 * The six `fa*_forward/backward` entry points below are what the reference's Python
 * wrappers call on its native extension module (reference file:line, relative to the
 * reference tree):
 *
 *   csrc/common/torch.extension.cpp:74  m.def("fa1_forward",  &fa1_forward)   -> fa1_forward
 *   csrc/common/torch.extension.cpp:75  m.def("fa1_backward", &fa1_backward)  -> fa1_backward
 *   csrc/common/torch.extension.cpp:78  m.def("forward",      &fa2_forward)   -> fa2_forward
 *   csrc/common/torch.extension.cpp:79  m.def("backward",     &fa2_backward)  -> fa2_backward
 *   csrc/common/torch.extension.cpp:81  m.def("fa3_forward",  &fa3_forward)   -> fa3_forward
 *   csrc/common/torch.extension.cpp:82  m.def("fa3_backward", &fa3_backward)  -> fa3_backward
 *
 * The reference's functions take and return `torch::Tensor`; here every tensor is a
 * plain device pointer plus sizes, the caller owns all memory (inputs, outputs and the
 * backward workspace), and work is enqueued on the caller's `hipStream_t` (passed as
 * `void*`) without synchronising the device.  The Python shim
 * `flashattention-pytorch_amd/flashattention_lab_cuda/__init__.py` re-creates the
 * reference's tensor-level signatures on top of these via ctypes.
 *
 * Tensor layout (same as the reference, csrc/fa2/fa2_fwd.cu:40-54):
 *   q, k, v, o, do, dq, dk, dv : contiguous row-major (BH, N, d), element type `dtype`
 *   lse                        : contiguous (BH, N) float32, natural-log logsumexp
 * Results do not depend on the tile hints `br`, `bc`, `stages`.
 *
 * Every function returns FA_OK (0) or a negative error code; `fa_last_error()` returns a
 * thread-local message for the last failure.
 */
#ifndef FA_MI355X_H
#define FA_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_OK 0
#define FA_ERR_INVALID_ARGUMENT (-1) /* bad shape / dtype / null pointer (reference: TORCH_CHECK -> RuntimeError) */
#define FA_ERR_UNSUPPORTED (-2)      /* head_dim > 256, or no gfx950 device */
#define FA_ERR_WORKSPACE (-3)        /* workspace too small */
#define FA_ERR_LAUNCH (-4)           /* HIP launch error */

/* element type of q/k/v/o/do/dq/dk/dv */
#define FA_DTYPE_F32 0
#define FA_DTYPE_F16 1
#define FA_DTYPE_BF16 2

/* Which implementation the dispatcher picks (fa_set_kernel_mode): AUTO = MFMA bf16/f16 kernels
 * when dtype is 16-bit and d is a multiple of 8 up to 256 (64 / 128 / 256 wide tiles, narrower rows zero-padded in the
 * kernel), exact-f32 MFMA kernels otherwise.
 * The mode and the fa_set_option knobs are PROCESS-GLOBAL tuning state (sweeps, A/B runs): set them before the
 * streams start working, not concurrently with calls from other threads. */
#define FA_MODE_AUTO 0
#define FA_MODE_F32_GENERIC 1
#define FA_MODE_BWD_ATOMIC 2 /* as AUTO, but the 16-bit backward is the single-kernel variant with float-atomic dQ */

/* --- FlashAttention-1 names (replaces csrc/fa1/fa1_fwd.cu:30, csrc/fa1/fa1_bwd.cu:30) --- */
int fa1_forward(const void* q, const void* k, const void* v, void* o, float* lse,
                int64_t bh, int64_t n, int64_t d, int dtype,
                int causal, double softmax_scale, int64_t br, int64_t bc, void* stream);

int fa1_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                 void* dq, void* dk, void* dv,
                 int64_t bh, int64_t n, int64_t d, int dtype,
                 int causal, double softmax_scale, int64_t br, int64_t bc,
                 void* workspace, size_t workspace_bytes, void* stream);

/* --- FlashAttention-2 (Python names `forward` / `backward`; replaces csrc/fa2/fa2_fwd.cu:30, fa2_bwd.cu:30) --- */
int fa2_forward(const void* q, const void* k, const void* v, void* o, float* lse,
                int64_t bh, int64_t n, int64_t d, int dtype,
                int causal, double softmax_scale, int64_t br, int64_t bc, void* stream);

int fa2_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                 void* dq, void* dk, void* dv,
                 int64_t bh, int64_t n, int64_t d, int dtype,
                 int causal, double softmax_scale, int64_t br, int64_t bc,
                 void* workspace, size_t workspace_bytes, void* stream);

/* --- FlashAttention-3 (replaces csrc/fa3/fa3_fwd.cu:172, csrc/fa3/fa3_bwd.cu:104).
 * fp8 != 0: where the e4m3 kernel exists (f16/bf16 tensors, d = 128) Q and K are quantised to OCP e4m3 with one
 * scale per (bh, 64-row block) and QK^T runs on the fp8 MFMA; V, P and all accumulation stay 16/32-bit, and the
 * forward needs a workspace (fa3_forward_workspace_bytes, 0 when the kernel does not apply).  Other shapes take the
 * regular 16/32-bit path.  The fp8 backward differentiates the function the forward evaluated (attention of the
 * e4m3-round-tripped Q, K; cf. csrc/fa3/fa3_bwd.cu:134-146) and needs fa3_backward_workspace_bytes. */
int fa3_forward(const void* q, const void* k, const void* v, void* o, float* lse,
                int64_t bh, int64_t n, int64_t d, int dtype,
                int causal, double softmax_scale, int64_t br, int64_t bc, int64_t stages, int fp8,
                void* workspace, size_t workspace_bytes, void* stream);

int fa3_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                 void* dq, void* dk, void* dv,
                 int64_t bh, int64_t n, int64_t d, int dtype,
                 int causal, double softmax_scale, int64_t br, int64_t bc, int64_t stages, int fp8,
                 void* workspace, size_t workspace_bytes, void* stream);

/* --- Extended attention: the extras the reference's notebook model wires around its tiled attention,
 * src/fa3/torch/flashattention_pytorch.py (MultiHeadAttention._block_sparse_flash_attention :94-174, look_ahead_mask_ :176-190,
 * dense branch :80-87; src/common/dropout.py:3-15), as kernel features behind one entry point.
 *   q, o, do, dq : (BH, Nq, d)      k, v, dk, dv : (BH, Nk, d)      lse : (BH, Nq) float32
 *   causal != 0      : key j is visible to query i iff j <= i + (Nk - Nq)          (look_ahead_mask_, bottom-right aligned)
 *   mask             : Nq x Nk bytes, 0 = masked (masked_fill(mask == 0, -inf)); mask_bh_stride = 0 shares one mask over all
 *                      (b,h), Nq*Nk gives every (b,h) its own; NULL = none
 *   block_mask       : ceil(Nq/br) x ceil(Nk/bc) bytes, 0 = the tile is skipped (Algorithm 5 line 8); NULL = none
 *   dropout_p, seed  : standard dropout of the attention probabilities, scale 1/(1-p); an element is kept where its 16
 *                      uniform bits — a counter-based generator of (seed, b*h, i, j): one splitmix64 value per 2 x 2 quad of
 *                      (row, key) elements — are >= floor(65536 p) + 1, so the backward regenerates the same mask; 0 = none
 *   softmax_scale    : includes the model's temperature tau
 * A query row with no visible key returns o = 0, lse = -inf, dq = 0 (the reference's softmax of an all -inf row is NaN).
 * Kernels: f16 / bf16 tensors with d % 8 == 0, d <= 128, softmax_scale > 0 and block-mask blocks that are multiples of 32 run
 * on 16-bit MFMA kernels (a call without mask and dropout takes the plain fa2 kernels: square as it is; Nq != Nk at d = 128,
 * causal only with Nk >= Nq); everything else — f32, d <= 256 — on exact-f32
 * kernels.  Same results contract either way. */
int fa_ex_forward(const void* q, const void* k, const void* v, void* o, float* lse,
                  int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype,
                  int causal, double softmax_scale,
                  const uint8_t* mask, int64_t mask_bh_stride,
                  const uint8_t* block_mask, int64_t br, int64_t bc,
                  double dropout_p, uint64_t dropout_seed, void* stream);

int fa_ex_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                   void* dq, void* dk, void* dv,
                   int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype,
                   int causal, double softmax_scale,
                   const uint8_t* mask, int64_t mask_bh_stride,
                   const uint8_t* block_mask, int64_t br, int64_t bc,
                   double dropout_p, uint64_t dropout_seed,
                   void* workspace, size_t workspace_bytes, void* stream);

size_t fa_ex_backward_workspace_bytes(int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype);
/* The size with which a call WITHOUT mask, block-sparse mask and dropout (extras = 0) hands dS from its dK/dV kernel to its dQ
 * kernel, as fa_backward_workspace_bytes_fast does for the square backward and by the same rule (d = 128, 16-bit tensors; Nq != Nk
 * included, under the causal mask with Nk >= Nq; at most 4 GiB more).  Equals fa_ex_backward_workspace_bytes where that does not apply. */
size_t fa_ex_backward_workspace_bytes_fast(int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype, int causal, int extras);

/* --- support entry points (no reference counterpart: the reference allocates inside the callee) --- */
/* bytes for the CURRENT kernel mode: two float row constants per query row (+ an fp32 dQ scratch of bh*n*d floats in
 * FA_MODE_BWD_ATOMIC only); ask again after changing the mode */
size_t fa_backward_workspace_bytes(int64_t bh, int64_t n, int64_t d, int dtype);
/* The size that lets the backward hand dS from its dK/dV kernel to its dQ kernel instead of recomputing S and dP there
 * (d = 128, 16-bit tensors: N * N * 2 bytes per (b,h), at most 4 GiB whatever BH and N are — the (b,h) units are worked through
 * in equal chunks of that size; option ds_chunk_mb moves the bound).  Without the causal mask that serves every launch; under it
 * rows of 4096 and more, or launches of 160 and more 256-row tiles, while a chunk holds 16 units or the whole launch (option
 * dq = 6 forces it elsewhere).  Equals fa_backward_workspace_bytes where it does not apply.  A backward call given less than this
 * (but at least fa_backward_workspace_bytes) runs the recomputing dQ pass: same results up to summation order, 4 - 17 % more
 * backward time depending on the launch (profiles/r03_bwd_variants.md, r03_ds_chunk_sweep.md). */
size_t fa_backward_workspace_bytes_fast(int64_t bh, int64_t n, int64_t d, int dtype, int causal);
size_t fa3_forward_workspace_bytes(int64_t bh, int64_t n, int64_t d, int dtype, int fp8);
size_t fa3_backward_workspace_bytes(int64_t bh, int64_t n, int64_t d, int dtype, int fp8);
const char* fa_last_error(void);
const char* fa_version(void);
int fa_set_kernel_mode(int mode);      /* FA_MODE_*; returns the previous mode */
/* debug only: device buffer of 4096 int64 for the phase timestamps of the staggered forward kernel (NULL = off) */
int fa_debug_trace_buffer(void* device_ptr);
int fa_set_option(const char* name, int value); /* tuning knobs for sweeps: fwd_kb, fwd_stag, fwd_tpw, dq_tpw, dkdv_tpw, ... (csrc/fa_kernels.h) */
int fa_device_is_gfx950(int device);   /* 1 if `device` reports gcnArchName gfx950, 0 otherwise, <0 on HIP error */
/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline figure).
 * fa_profile_enable(1) starts collecting (and clears old records), fa_profile_report waits for the events and
 * writes one "kernel_name launches total_ms" line per kernel; returns bytes written or a negative code. */
int fa_profile_enable(int on);
int fa_profile_report(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* FA_MI355X_H */
