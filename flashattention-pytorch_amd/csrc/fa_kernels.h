// Internal launcher interface between the C-ABI layer (fa_capi.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fa {

struct FwdArgs {
    const void *q, *k, *v;
    void* o;
    float* lse;
    int64_t bh, n, d;
    int dtype;  // FA_DTYPE_*
    int causal;
    float scale;
    int64_t nk = 0;   // keys (0: = n, the query rows).  Only the kernels fwd_nqnk_supported() names take nk != n.
};

struct BwdArgs {
    const void *q, *k, *v, *o, *dout;
    const float* lse;
    void *dq, *dk, *dv;
    int64_t bh, n, d;
    int dtype;
    int causal;
    float scale;
    void* workspace;
    size_t workspace_bytes;
    int fused_dq;  // 1: single kernel, dQ by global float atomics; 0: dK/dV kernel + dQ kernel (deterministic)
    int64_t nk = 0;   // keys (0: = n, the query rows).  Only the stream kernels take nk != n (causal: nk >= n).
};

// Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the
// roofline figure; off by default, costs nothing when off).
enum KernelId { K_FWD_F32 = 0, K_BWD_DELTA, K_BWD_DKDV_F32, K_BWD_DQ_F32, K_FWD_MFMA, K_BWD_MFMA, K_BWD_DQ_CVT, K_BWD_DQ_MFMA,
                K_FP8_QUANT, K_FWD_FP8, K_EX_FWD, K_EX_BWD, K_COUNT };
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);
struct ProfScope {
    int id; hipStream_t st;
    ProfScope(int i, hipStream_t s) : id(i), st(s) { prof_begin(id, st); }
    ~ProfScope() { prof_end(id, st); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size instead of before every launch (a host
// call of a few microseconds: visible on the small launches).  Device-wide attribute; thread safe.
hipError_t ensure_dynamic_smem(const void* kernel, int bytes);

// Tuning knobs (tile sweep / A-B runs).  Each starts from an environment variable of the same upper-case name with
// an FA_ prefix (FA_FWD_KB, FA_FWD_STAG, FA_DKDV (4|8), FA_DQ_KT, FA_FWD_RS, FA_DKDV_KREG, FA_FWD_EAGER, FA_FWD_HS, FA_FWD_TPW, FA_DQ_TPW, FA_DKDV_TPW, FA_DQ_NLF, FA_DQ_W4, FA_FWD_ABL, FA_SMALL_GRID, FA_FP8_ROT, FA_DKDV_STG) and can be changed at run time through
// fa_set_option() so that variants can be interleaved in one process.
enum OptionId { OPT_FWD_KB = 0, OPT_FWD_STAG, OPT_DKDV, OPT_DQ_KT, OPT_FWD_RS, OPT_DKDV_KREG, OPT_FWD_EAGER, OPT_FWD_HS, OPT_FWD_TPW, OPT_DQ_TPW, OPT_DKDV_TPW, OPT_DQ_NLF, OPT_DQ_W4, OPT_FWD_ABL, OPT_SMALL_GRID, OPT_FP8_ROT, OPT_DKDV_STG, OPT_DKDV_ABL, OPT_DQ, OPT_DQ_ABL, OPT_EX_PATH, OPT_DS_CHUNK_MB, OPT_FP8_PV, OPT_FWD_RD, OPT_FWD_W2, OPT_COUNT };
int option(int id);
int set_option(const char* name, int value);   // returns 0, or -1 for an unknown name

// exact-f32 kernels (fa_generic.hip): any dtype, d <= 256
hipError_t launch_fwd_generic(const FwdArgs& a, hipStream_t st);
hipError_t launch_bwd_generic(const BwdArgs& a, hipStream_t st);
size_t bwd_generic_workspace_bytes(int64_t bh, int64_t n);

// 16-bit MFMA kernels (fa_fwd_mfma.hip / fa_bwd_mfma.hip): f16/bf16, d in {64, 128}
bool fwd_mfma_supported(int dtype, int64_t d);
bool small_grid(int64_t bh, int64_t n, bool backward, int64_t d = 128);   // the 4-wave / 128-row kernels serve the launch better than 256-row tiles
hipError_t set_trace_buffer(void* device_ptr);   // debug: phase timestamps of the staggered forward (fa_fwd_mfma.hip)
hipError_t launch_fwd_mfma(const FwdArgs& a, hipStream_t st);
// Nq != Nk on the plain path's d = 128 kernels (staggered forward, stream backward): 16-bit tensors, Nk % 64 == 0 not needed,
// causal only with Nk >= Nq, launches big enough for the 256-row tiles
bool nqnk_mfma_supported(int dtype, int64_t d, int64_t bh, int64_t nq, int64_t nk, int causal);
hipError_t launch_fwd_nqnk(const FwdArgs& a, hipStream_t st);
bool bwd_mfma_supported(int dtype, int64_t d);
hipError_t launch_bwd_mfma(const BwdArgs& a, hipStream_t st);
size_t bwd_mfma_workspace_bytes(int64_t bh, int64_t n, int64_t d, bool atomic_variant);   // fp32 dQ scratch only for the single-kernel variant
// bytes the dS hand-over (fa_bwd_dq_ds.hip) wants on top of that; 0 where it does not serve the call
size_t bwd_ds_extra_bytes(int64_t bh, int64_t n, int64_t d, int dtype, bool causal, bool atomic_variant, int64_t nk = 0);
// row constants + the chunk loop [dK/dV with dS stores, dQ product]; a.nk keys (0: = a.n), ds: bwd_ds_extra_bytes of room
hipError_t launch_bwd_handover(const BwdArgs& a, float* nlse, float* ndelta, void* ds, hipStream_t st);
hipError_t launch_bwd_dkdv_mfma(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st);  // 8-wave dK/dV
hipError_t launch_bwd_dq_mfma(const BwdArgs& a, float* nlse, float* ndelta, hipStream_t st);   // d > 64: also WRITES nlse / ndelta
inline bool dq_makes_row_constants(int64_t d) { return d > 64; }
// one wave per SIMD, 64 keys per wave, hand-ordered MFMA stream (fa_bwd_dkdv_w4.hip): d = 128
bool bwd_dkdv_w4_supported(int dtype, int64_t d);
// ds != null: the kernel also stores the packed dS tiles there (ds_workspace_bytes) for launch_bwd_dq_ds
hipError_t launch_bwd_dkdv_w4(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st, void* ds = nullptr);
// dS hand-over between the dK/dV stream kernel and the dQ product kernel (fa_bwd_dq_ds.hip): 2-KiB tiles of 32 queries x
// 32 keys, (b,h)-major, then 32-query block, then 32-key block; the key blocks are padded to the dK/dV kernel's 256-key tiles
inline int ds_tile_rows(int64_t nq) { return (int)((nq + 31) / 32); }
inline int ds_tile_cols(int64_t nk) { return (int)(8 * ((nk + 255) / 256)); }
inline size_t ds_workspace_bytes(int64_t bh, int64_t nq, int64_t nk) { return (size_t)bh * ds_tile_rows(nq) * ds_tile_cols(nk) * 2048; }
// dQ = scale * dS K from the stored dS tiles: d = 128, one pass over dS at HBM rate (no recomputation of S and dP)
hipError_t launch_bwd_dq_ds(const BwdArgs& a, const void* ds, hipStream_t st);

// the dQ pass in the same shape (fa_bwd_dq_w4.hip): d = 128; writes nlse / ndelta like launch_bwd_dq_mfma at d > 64
bool bwd_dq_w4_supported(int dtype, int64_t d);
hipError_t launch_bwd_dq_w4(const BwdArgs& a, float* nlse, float* ndelta, hipStream_t st);

// FA3-style fp8 forward (fa_fwd_fp8.hip): Q/K quantised to e4m3 per 64-row block, S on the fp8 MFMA
bool fwd_fp8_supported(int dtype, int64_t d);
// workspace: fwd_fp8_workspace_bytes; vslab: room for one 16-bit (bh, n, d) tensor (the round-tripped V of the 16-bit P.V kernel)
hipError_t launch_fwd_fp8(const FwdArgs& a, void* workspace, void* vslab, hipStream_t st);
size_t fwd_fp8_workspace_bytes(int64_t bh, int64_t n, int64_t d);
// e4m3 round trip (one scale per 64-row block) of q, k (rotated around the quantisation for power-of-two d) and v into 16-bit
// tensors, any d % 8 == 0 up to 256; null sources are skipped
hipError_t launch_fp8_roundtrip(const void* q, const void* k, const void* v, void* qt, void* kt, void* vt, int64_t bh, int64_t n,
                                int64_t d, int dtype, hipStream_t st);

// Extended attention: Nq != Nk with a bottom-right aligned causal mask, dense mask, block-sparse mask, dropout.
// fa_ex.hip: exact-f32 kernels, any dtype, d <= 256.  fa_ex_mfma.hip: bf16 / f16, d % 8 == 0 up to 128, block-sparse
// blocks that are multiples of 32 — the default where it applies (option ex_path: 1 = always exact f32,
// 2 = MFMA or fail, 3 = the extended MFMA kernels or fail, also where the plain kernels would do: square, no extras).
struct ExArgs {
    const void *q, *k, *v;        // q: (bh, nq, d); k, v: (bh, nk, d)
    void* o;                      // (bh, nq, d)
    float* lse;                   // (bh, nq)
    const void* dout;             // backward: (bh, nq, d)
    void *dq, *dk, *dv;           // backward outputs
    int64_t bh, nq, nk, d;
    int dtype, causal;
    float scale;
    const uint8_t* mask;          // (nq, nk) bytes, 0 = masked; null = none
    int64_t mask_bh_stride;       // 0 = one mask shared by all (b,h), nq * nk = one per (b,h)
    const uint8_t* block_mask;    // (ceil(nq/br), ceil(nk/bc)) bytes, 0 = tile skipped; null = none
    int64_t br, bc;
    double dropout_p;
    uint64_t seed;
    void* workspace;              // backward: ex_backward_workspace_bytes
    size_t workspace_bytes = 0;   // what the caller really gave (more than the minimum lets plain calls hand dS over)
};
hipError_t launch_ex(const ExArgs& a, bool backward, hipStream_t st);
bool ex_mfma_supported(const ExArgs& a);
hipError_t launch_ex_mfma(const ExArgs& a, bool backward, hipStream_t st);
size_t ex_backward_workspace_bytes(int64_t bh, int64_t nq);

}  // namespace fa
