// C-ABI entry points declared in include/fa_mi355x.h: argument validation, kernel dispatch,
// error reporting.  No torch types, no allocation, no device synchronisation: everything is
// enqueued on the caller's stream.
#include "../../include/fa_mi355x.h"
#include "fa_kernels.h"
#include <mutex>
#include <unordered_map>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <cstdint>

namespace {

thread_local char g_err[512] = "";
std::atomic<int> g_mode{FA_MODE_AUTO};

// ---- per-kernel event timing -------------------------------------------------------------------
struct ProfRec { int id; hipEvent_t a, b; };
std::atomic<int> g_prof_on{0};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_free;
hipEvent_t g_prof_open[fa::K_COUNT];
const char* const kKernelNames[fa::K_COUNT] = {"fwd_f32", "bwd_delta", "bwd_dkdv_f32", "bwd_dq_f32", "fwd_mfma",
                                               "bwd_mfma", "bwd_dq_cvt", "bwd_dq_mfma", "fp8_quant", "fwd_fp8", "ex_fwd", "ex_bwd"};

hipEvent_t prof_get_event() {
    if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// ---- tuning knobs ---------------------------------------------------------------------------------
constexpr const char* kOptNames[fa::OPT_COUNT] = {"fwd_kb", "fwd_stag", "dkdv", "dq_kt", "fwd_rs", "dkdv_kreg", "fwd_eager", "fwd_hs", "fwd_tpw", "dq_tpw", "dkdv_tpw", "dq_nlf", "dq_w4", "fwd_abl", "small_grid", "fp8_rot", "dkdv_stg", "dkdv_abl", "dq", "dq_abl", "ex_path", "ds_chunk_mb", "fp8_pv", "fwd_rd", "fwd_w2"};
constexpr const char* kOptEnv[fa::OPT_COUNT] = {"FA_FWD_KB", "FA_FWD_STAG", "FA_DKDV", "FA_DQ_KT", "FA_FWD_RS", "FA_DKDV_KREG", "FA_FWD_EAGER", "FA_FWD_HS", "FA_FWD_TPW", "FA_DQ_TPW", "FA_DKDV_TPW", "FA_DQ_NLF", "FA_DQ_W4", "FA_FWD_ABL", "FA_SMALL_GRID", "FA_FP8_ROT", "FA_DKDV_STG", "FA_DKDV_ABL", "FA_DQ", "FA_DQ_ABL", "FA_EX_PATH", "FA_DS_CHUNK_MB", "FA_FP8_PV", "FA_FWD_RD", "FA_FWD_W2"};
// a name added to OptionId without its two strings here would leave a null at the end of a table
template <size_t N> constexpr bool all_set(const char* const (&t)[N]) {
    for (size_t i = 0; i < N; ++i)
        if (!t[i]) return false;
    return true;
}
static_assert(all_set(kOptNames) && all_set(kOptEnv), "kOptNames / kOptEnv: one string per OptionId");
std::atomic<int> g_opts[fa::OPT_COUNT];
std::once_flag g_opts_once;
void init_opts() {
    for (int i = 0; i < fa::OPT_COUNT; ++i) {
        const char* e = getenv(kOptEnv[i]);
        int v = 0;
        if (e) v = (e[0] == 'w') ? atoi(e + 1) : atoi(e);   // FA_DKDV=w4 / w8 are accepted as 4 / 8
        g_opts[i].store(v);
    }
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_common(const char* who, int64_t bh, int64_t n, int64_t d, int dtype, double scale) {
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_F16 && dtype != FA_DTYPE_BF16)
        return fail(FA_ERR_INVALID_ARGUMENT, "%s: unknown dtype code %d", who, dtype);
    if (bh < 0 || n < 0 || d <= 0)
        return fail(FA_ERR_INVALID_ARGUMENT, "%s: bad shape (BH=%lld, N=%lld, d=%lld)", who, (long long)bh, (long long)n,
                    (long long)d);
    if (d > 256) return fail(FA_ERR_UNSUPPORTED, "%s: head_dim %lld > 256 is not supported", who, (long long)d);
    if (n > (int64_t)1 << 24) return fail(FA_ERR_UNSUPPORTED, "%s: N=%lld too large", who, (long long)n);
    if (bh * ((n + 63) / 64) >= ((int64_t)1 << 31))
        return fail(FA_ERR_UNSUPPORTED, "%s: BH*N too large for one launch", who);
    if (!(scale == scale)) return fail(FA_ERR_INVALID_ARGUMENT, "%s: softmax_scale is NaN", who);
    return FA_OK;
}

// FA_MODE_BWD_ATOMIC (or the environment variable FA_BWD_VARIANT=atomic) selects the single-kernel backward whose
// dQ tiles are summed with global float atomics (5 GEMMs, not bitwise reproducible); the default is the split
// backward: dK/dV kernel + dQ kernel (7 GEMMs, no atomics, deterministic).
bool bwd_atomic_variant() {
    static const int env = [] { const char* e = getenv("FA_BWD_VARIANT"); return (e && !strcmp(e, "atomic")) ? 1 : 0; }();
    return env != 0 || g_mode.load() == FA_MODE_BWD_ATOMIC;
}
// the 16-bit MFMA kernels fold softmax_scale into the exp2 argument and need it finite and > 0
bool scale_ok(double s) { return s > 1e-20 && s < 1e20; }
// The 16-bit MFMA kernels read 16 bytes per lane and address each (b,h) slab with 32-bit byte offsets:
// every tensor must be 16-byte aligned and N*d*2 must stay below 2^31.  Anything else takes the exact-f32 kernels.
bool aligned16(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    return true;
}
bool slab_ok(int64_t n, int64_t d) { return n * d * 2 < ((int64_t)1 << 31) - 65536; }
bool use_mfma_fwd(int dtype, int64_t n, int64_t d, double s, std::initializer_list<const void*> ps) {
    return g_mode.load() != FA_MODE_F32_GENERIC && scale_ok(s) && slab_ok(n, d) && aligned16(ps) &&
           fa::fwd_mfma_supported(dtype, d);
}
bool use_mfma_bwd(int dtype, int64_t n, int64_t d, double s, std::initializer_list<const void*> ps) {
    return g_mode.load() != FA_MODE_F32_GENERIC && scale_ok(s) && slab_ok(n, d) && aligned16(ps) &&
           fa::bwd_mfma_supported(dtype, d);
}

// Does an fa3 call with fp8 = 1 take the e4m3 path?  ONE predicate for fa3_forward and fa3_backward, over arguments that are the
// same in both calls (the workspace is not: a misaligned one is an error there), so that the backward always differentiates the
// function the forward evaluated.  It holds wherever the 16-bit MFMA kernels serve the call (f16 / bf16 tensors, head dims that
// are multiples of 8 up to 256): Q, K and V then go through OCP e4m3 with one scale per 64-row block, as the reference's wiring
// has it (csrc/fa3/fa3_fwd.cu:196-208) — at d = 128 on the e4m3 MFMA kernel, elsewhere as a round trip ahead of the 16-bit kernels.
bool fp8_path(int dtype, int64_t n, int64_t d, double s, const void* q, const void* k, const void* v, const void* o) {
    return fa::fwd_mfma_supported(dtype, d) && fa::bwd_mfma_supported(dtype, d) && scale_ok(s) && g_mode.load() != FA_MODE_F32_GENERIC &&
           slab_ok(n, d) && aligned16({q, k, v, o});
}
size_t slab_bytes(int64_t bh, int64_t n, int64_t d) { return ((size_t)bh * n * d * 2 + 255) & ~(size_t)255; }   // one round-tripped 16-bit tensor

int forward_impl(const char* who, const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh,
                 int64_t n, int64_t d, int dtype, int causal, double scale, void* stream) {
    int rc = check_common(who, bh, n, d, dtype, scale);
    if (rc != FA_OK) return rc;
    if (bh == 0 || n == 0) return FA_OK;  // empty problem: nothing to write
    if (!q || !k || !v || !o || !lse) return fail(FA_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", who);
    fa::FwdArgs a{q, k, v, o, lse, bh, n, d, dtype, causal ? 1 : 0, (float)scale};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = use_mfma_fwd(dtype, n, d, scale, {q, k, v, o}) ? fa::launch_fwd_mfma(a, st) : fa::launch_fwd_generic(a, st);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "%s: HIP error %d (%s)", who, (int)e, hipGetErrorString(e));
    return FA_OK;
}

int backward_impl(const char* who, const void* q, const void* k, const void* v, const void* o, const void* dout,
                  const float* lse, void* dq, void* dk, void* dv, int64_t bh, int64_t n, int64_t d, int dtype,
                  int causal, double scale, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(who, bh, n, d, dtype, scale);
    if (rc != FA_OK) return rc;
    if (bh == 0 || n == 0) return FA_OK;
    if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv)
        return fail(FA_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", who);
    const size_t need = fa_backward_workspace_bytes(bh, n, d, dtype);
    if (!ws || ws_bytes < need)
        return fail(FA_ERR_WORKSPACE, "%s: workspace of %zu bytes needed, %zu given", who, need, ws_bytes);
    fa::BwdArgs a{q, k, v, o, dout, lse, dq, dk, dv, bh, n, d, dtype, causal ? 1 : 0, (float)scale, ws, ws_bytes,
                   bwd_atomic_variant() ? 1 : 0};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = use_mfma_bwd(dtype, n, d, scale, {q, k, v, o, dout, dq, dk, dv, ws}) ? fa::launch_bwd_mfma(a, st) : fa::launch_bwd_generic(a, st);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "%s: HIP error %d (%s)", who, (int)e, hipGetErrorString(e));
    return FA_OK;
}

}  // namespace

namespace fa {
int option(int id) {
    std::call_once(g_opts_once, init_opts);
    return g_opts[id].load(std::memory_order_relaxed);
}
int set_option(const char* name, int value) {
    std::call_once(g_opts_once, init_opts);
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, kOptNames[i])) { g_opts[i].store(value); return 0; }
    return -1;
}
void prof_begin(int id, hipStream_t st) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = prof_get_event();
    (void)hipEventRecord(e, st);
    g_prof_open[id] = e;
}
void prof_end(int id, hipStream_t st) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t e = prof_get_event();
    (void)hipEventRecord(e, st);
    g_prof_recs.push_back(ProfRec{id, g_prof_open[id], e});
}
}  // namespace fa

namespace fa {
hipError_t ensure_dynamic_smem(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> granted;   // (device, kernel) -> bytes: the attribute is per device
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t key = reinterpret_cast<uint64_t>(kernel) ^ (static_cast<uint64_t>(dev + 1) << 56);
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) granted[key] = bytes;
    return e;
}
}  // namespace fa


extern "C" {

int fa_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) { g_prof_free.push_back(r.a); g_prof_free.push_back(r.b); }
    g_prof_recs.clear();
    return g_prof_on.exchange(on ? 1 : 0);
}

// Waits for the recorded events and writes "name count total_ms\n" lines for every kernel launched since
// fa_profile_enable(1). Returns the number of bytes written (excluding the NUL), or a negative error code.
int fa_profile_report(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double total[fa::K_COUNT] = {0};
    long count[fa::K_COUNT] = {0};
    for (auto& r : g_prof_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_profile_report: event sync failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_profile_report: elapsed failed");
        total[r.id] += ms;
        count[r.id] += 1;
    }
    std::string out;
    char line[128];
    for (int i = 0; i < fa::K_COUNT; ++i) {
        if (!count[i]) continue;
        snprintf(line, sizeof(line), "%s %ld %.6f\n", kKernelNames[i], count[i], total[i]);
        out += line;
    }
    if (!buf || cap == 0) return (int)out.size();
    const size_t nw = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), nw);
    buf[nw] = 0;
    return (int)nw;
}

int fa1_forward(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int64_t d,
                int dtype, int causal, double softmax_scale, int64_t br, int64_t bc, void* stream) {
    (void)br; (void)bc;  // tile hints: results are tile independent (SURVEY §8b "Tile params")
    return forward_impl("fa1_forward", q, k, v, o, lse, bh, n, d, dtype, causal, softmax_scale, stream);
}

int fa1_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                 void* dq, void* dk, void* dv, int64_t bh, int64_t n, int64_t d, int dtype, int causal,
                 double softmax_scale, int64_t br, int64_t bc, void* workspace, size_t workspace_bytes, void* stream) {
    (void)br; (void)bc;
    return backward_impl("fa1_backward", q, k, v, o, do_, lse, dq, dk, dv, bh, n, d, dtype, causal, softmax_scale,
                         workspace, workspace_bytes, stream);
}

int fa2_forward(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int64_t d,
                int dtype, int causal, double softmax_scale, int64_t br, int64_t bc, void* stream) {
    (void)br; (void)bc;
    return forward_impl("fa2_forward", q, k, v, o, lse, bh, n, d, dtype, causal, softmax_scale, stream);
}

int fa2_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                 void* dq, void* dk, void* dv, int64_t bh, int64_t n, int64_t d, int dtype, int causal,
                 double softmax_scale, int64_t br, int64_t bc, void* workspace, size_t workspace_bytes, void* stream) {
    (void)br; (void)bc;
    return backward_impl("fa2_backward", q, k, v, o, do_, lse, dq, dk, dv, bh, n, d, dtype, causal, softmax_scale,
                         workspace, workspace_bytes, stream);
}

int fa3_forward(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int64_t d,
                int dtype, int causal, double softmax_scale, int64_t br, int64_t bc, int64_t stages, int fp8,
                void* workspace, size_t workspace_bytes, void* stream) {
    (void)br; (void)bc; (void)stages;
    // fp8: Q, K and V through e4m3 (see fp8_path).  fp32 tensors and head dims the 16-bit kernels do not take run the regular,
    // more accurate path (as the reference quietly skips its rotation for non-power-of-two d, src/fa3/torch/impl.py:60-61).
    if (fp8 && fp8_path(dtype, n, d, softmax_scale, q, k, v, o)) {
        int rc = check_common("fa3_forward", bh, n, d, dtype, softmax_scale);
        if (rc != FA_OK) return rc;
        if (bh == 0 || n == 0) return FA_OK;
        if (!q || !k || !v || !o || !lse) return fail(FA_ERR_INVALID_ARGUMENT, "fa3_forward: null tensor pointer");
        const size_t need = fa3_forward_workspace_bytes(bh, n, d, dtype, 1);
        if (!workspace || workspace_bytes < need || !aligned16({workspace}))
            return fail(FA_ERR_WORKSPACE, "fa3_forward: a 16-byte aligned workspace of %zu bytes is needed, %zu given", need, workspace_bytes);
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const size_t slab = slab_bytes(bh, n, d);
        char* ws = reinterpret_cast<char*>(workspace);
        hipError_t e;
        if (fa::fwd_fp8_supported(dtype, d)) {
            // d = 128: [V~][the e4m3 kernels' own workspace: Q, K, V^T bytes and scales].  Default: S and P.V on the e4m3 MFMA;
            // option fp8_pv = 1: S on the e4m3 MFMA, P.V 16-bit on the round-tripped V.
            fa::FwdArgs a{q, k, v, o, lse, bh, n, d, dtype, causal ? 1 : 0, (float)softmax_scale};
            e = fa::launch_fwd_fp8(a, ws + slab, ws, st);
        } else {
            // [Q~][K~][V~] (original basis), then the 16-bit kernels as they are
            e = fa::launch_fp8_roundtrip(q, k, v, ws, ws + slab, ws + 2 * slab, bh, n, d, dtype, st);
            if (e == hipSuccess)
                return forward_impl("fa3_forward", ws, ws + slab, ws + 2 * slab, o, lse, bh, n, d, dtype, causal, softmax_scale, stream);
        }
        if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa3_forward: HIP error %d (%s)", (int)e, hipGetErrorString(e));
        return FA_OK;
    }
    return forward_impl("fa3_forward", q, k, v, o, lse, bh, n, d, dtype, causal, softmax_scale, stream);
}

int fa3_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse,
                 void* dq, void* dk, void* dv, int64_t bh, int64_t n, int64_t d, int dtype, int causal,
                 double softmax_scale, int64_t br, int64_t bc, int64_t stages, int fp8, void* workspace,
                 size_t workspace_bytes, void* stream) {
    (void)br; (void)bc; (void)stages;
    // fp8: differentiate the function the forward evaluated, i.e. attention of the e4m3-round-tripped Q, K and V
    // (the reference's fa3_backward does the same, csrc/fa3/fa3_bwd.cu:134-146); the gradients are returned for
    // q, k, v themselves (straight-through over the rounding).  o and lse then match the recomputed probabilities.
    if (fp8 && fp8_path(dtype, n, d, softmax_scale, q, k, v, o) && bh > 0 && n > 0) {
        int rc = check_common("fa3_backward", bh, n, d, dtype, softmax_scale);
        if (rc != FA_OK) return rc;
        if (!q || !k || !v) return fail(FA_ERR_INVALID_ARGUMENT, "fa3_backward: null tensor pointer");
        const size_t need = fa3_backward_workspace_bytes(bh, n, d, dtype, 1);
        if (!workspace || workspace_bytes < need || !aligned16({workspace}))
            return fail(FA_ERR_WORKSPACE, "fa3_backward: a 16-byte aligned workspace of %zu bytes is needed, %zu given", need, workspace_bytes);
        // layout: [Q~][K~][V~][the plain backward's workspace: everything that is left, so a caller who sized it with
        // fa_backward_workspace_bytes_fast's surplus gets the dS hand-over here too]
        const size_t slab = slab_bytes(bh, n, d);
        char* qt = reinterpret_cast<char*>(workspace);
        char *kt = qt + slab, *vt = qt + 2 * slab;
        hipError_t e = fa::launch_fp8_roundtrip(q, k, v, qt, kt, vt, bh, n, d, dtype, reinterpret_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa3_backward: HIP error %d (%s)", (int)e, hipGetErrorString(e));
        return backward_impl("fa3_backward", qt, kt, vt, o, do_, lse, dq, dk, dv, bh, n, d, dtype, causal, softmax_scale,
                             qt + 3 * slab, workspace_bytes - 3 * slab, stream);
    }
    return backward_impl("fa3_backward", q, k, v, o, do_, lse, dq, dk, dv, bh, n, d, dtype, causal, softmax_scale,
                         workspace, workspace_bytes, stream);
}

// ---- extended attention: see include/fa_mi355x.h
static int ex_check(const char* who, int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype, double scale, const uint8_t* block_mask,
                    int64_t br, int64_t bc, double p) {
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_F16 && dtype != FA_DTYPE_BF16)
        return fail(FA_ERR_INVALID_ARGUMENT, "%s: unknown dtype code %d", who, dtype);
    if (bh < 0 || nq < 0 || nk < 0 || d <= 0)
        return fail(FA_ERR_INVALID_ARGUMENT, "%s: bad shape (BH=%lld, Nq=%lld, Nk=%lld, d=%lld)", who, (long long)bh, (long long)nq,
                    (long long)nk, (long long)d);
    if (d > 256) return fail(FA_ERR_UNSUPPORTED, "%s: head_dim %lld > 256 is not supported", who, (long long)d);
    if (nq > (int64_t)1 << 24 || nk > (int64_t)1 << 24 || nq * d >= ((int64_t)1 << 31) || nk * d >= ((int64_t)1 << 31) ||
        bh * ((nq + 15) / 16) >= ((int64_t)1 << 31) || bh * ((nk + 15) / 16) >= ((int64_t)1 << 31))
        return fail(FA_ERR_UNSUPPORTED, "%s: problem too large for one launch", who);
    if (!(scale == scale)) return fail(FA_ERR_INVALID_ARGUMENT, "%s: softmax_scale is NaN", who);
    if (block_mask && (br <= 0 || bc <= 0)) return fail(FA_ERR_INVALID_ARGUMENT, "%s: block-sparse mask needs br, bc > 0", who);
    if (!(p >= 0.0 && p < 1.0)) return fail(FA_ERR_INVALID_ARGUMENT, "%s: dropout_p must lie in [0, 1)", who);
    return FA_OK;
}

int fa_ex_forward(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t nq, int64_t nk, int64_t d,
                  int dtype, int causal, double softmax_scale, const uint8_t* mask, int64_t mask_bh_stride,
                  const uint8_t* block_mask, int64_t br, int64_t bc, double dropout_p, uint64_t dropout_seed, void* stream) {
    int rc = ex_check("fa_ex_forward", bh, nq, nk, d, dtype, softmax_scale, block_mask, br, bc, dropout_p);
    if (rc != FA_OK) return rc;
    if (bh == 0 || nq == 0) return FA_OK;
    if (!q || !o || !lse || (nk > 0 && (!k || !v))) return fail(FA_ERR_INVALID_ARGUMENT, "fa_ex_forward: null tensor pointer");
    if (nk == 0) {   // no key at all: every row is a row without a visible key, o = 0 and lse = -inf
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        hipError_t e = hipMemsetAsync(o, 0, (size_t)bh * nq * d * (dtype == FA_DTYPE_F32 ? 4 : 2), st);
        if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(lse), (int)0xFF800000u, (size_t)bh * nq, st);
        if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_ex_forward: HIP error %d (%s)", (int)e, hipGetErrorString(e));
        return FA_OK;
    }
    fa::ExArgs a{q, k, v, o, lse, nullptr, nullptr, nullptr, nullptr, bh, nq, nk, d, dtype, causal ? 1 : 0, (float)softmax_scale,
                 mask, mask_bh_stride, block_mask, br, bc, dropout_p, dropout_seed, nullptr};
    hipError_t e = fa::launch_ex(a, false, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_ex_forward: HIP error %d (%s)", (int)e, hipGetErrorString(e));
    return FA_OK;
}

int fa_ex_backward(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse, void* dq, void* dk,
                   void* dv, int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype, int causal, double softmax_scale,
                   const uint8_t* mask, int64_t mask_bh_stride, const uint8_t* block_mask, int64_t br, int64_t bc, double dropout_p,
                   uint64_t dropout_seed, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = ex_check("fa_ex_backward", bh, nq, nk, d, dtype, softmax_scale, block_mask, br, bc, dropout_p);
    if (rc != FA_OK) return rc;
    if (bh == 0 || (nq == 0 && nk == 0)) return FA_OK;
    if (nq == 0 || nk == 0) {   // one side empty: the gradients of the other side are sums over nothing
        const size_t es = dtype == FA_DTYPE_F32 ? 4 : 2;
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        hipError_t e = hipSuccess;
        if (nq == 0) {
            if (!dk || !dv) return fail(FA_ERR_INVALID_ARGUMENT, "fa_ex_backward: null tensor pointer");
            e = hipMemsetAsync(dk, 0, (size_t)bh * nk * d * es, st);
            if (e == hipSuccess) e = hipMemsetAsync(dv, 0, (size_t)bh * nk * d * es, st);
        } else {
            if (!dq) return fail(FA_ERR_INVALID_ARGUMENT, "fa_ex_backward: null tensor pointer");
            e = hipMemsetAsync(dq, 0, (size_t)bh * nq * d * es, st);
        }
        if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_ex_backward: HIP error %d (%s)", (int)e, hipGetErrorString(e));
        return FA_OK;
    }
    if (!q || !k || !v || !o || !do_ || !lse || !dq || !dk || !dv) return fail(FA_ERR_INVALID_ARGUMENT, "fa_ex_backward: null tensor pointer");
    const size_t need = fa_ex_backward_workspace_bytes(bh, nq, nk, d, dtype);
    if (!workspace || workspace_bytes < need)
        return fail(FA_ERR_WORKSPACE, "fa_ex_backward: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    fa::ExArgs a{q, k, v, const_cast<void*>(o), const_cast<float*>(lse), do_, dq, dk, dv, bh, nq, nk, d, dtype, causal ? 1 : 0,
                 (float)softmax_scale, mask, mask_bh_stride, block_mask, br, bc, dropout_p, dropout_seed, workspace, workspace_bytes};
    hipError_t e = fa::launch_ex(a, true, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_ex_backward: HIP error %d (%s)", (int)e, hipGetErrorString(e));
    return FA_OK;
}

size_t fa_ex_backward_workspace_bytes(int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype) {
    (void)nk; (void)d; (void)dtype;
    if (bh <= 0 || nq <= 0) return 256;
    return (fa::ex_backward_workspace_bytes(bh, nq) + 255) & ~(size_t)255;
}

size_t fa_ex_backward_workspace_bytes_fast(int64_t bh, int64_t nq, int64_t nk, int64_t d, int dtype, int causal, int extras) {
    size_t need = fa_ex_backward_workspace_bytes(bh, nq, nk, d, dtype);
    if (extras || bh <= 0 || nq <= 0 || nk <= 0 || g_mode.load() == FA_MODE_F32_GENERIC || fa::option(fa::OPT_EX_PATH) == 1 || fa::option(fa::OPT_EX_PATH) == 3) return need;
    if (!fa::bwd_mfma_supported(dtype, d)) return need;
    if (nq == nk) return need + fa::bwd_ds_extra_bytes(bh, nq, d, dtype, causal != 0, bwd_atomic_variant());   // the plain backward's own rule
    if (!fa::nqnk_mfma_supported(dtype, d, bh, nq, nk, causal)) return need;   // (under the mask: Nk >= Nq)
    return need + fa::bwd_ds_extra_bytes(bh, nq, d, dtype, causal != 0, false, nk);
}

size_t fa3_backward_workspace_bytes(int64_t bh, int64_t n, int64_t d, int dtype, int fp8) {
    size_t need = fa_backward_workspace_bytes(bh, n, d, dtype);
    if (fp8 && bh > 0 && n > 0 && d > 0 && fa::fwd_mfma_supported(dtype, d) && fa::bwd_mfma_supported(dtype, d)) need += 3 * slab_bytes(bh, n, d);
    return need;
}

size_t fa_backward_workspace_bytes(int64_t bh, int64_t n, int64_t d, int dtype) {
    if (bh <= 0 || n <= 0 || d <= 0) return 256;
    size_t g = fa::bwd_generic_workspace_bytes(bh, n);
    size_t m = fa::bwd_mfma_supported(dtype, d) ? fa::bwd_mfma_workspace_bytes(bh, n, d, bwd_atomic_variant()) : 0;
    size_t need = g > m ? g : m;
    return (need + 255) & ~(size_t)255;
}

size_t fa_backward_workspace_bytes_fast(int64_t bh, int64_t n, int64_t d, int dtype, int causal) {
    size_t need = fa_backward_workspace_bytes(bh, n, d, dtype);
    if (bh > 0 && n > 0 && d > 0 && fa::bwd_mfma_supported(dtype, d) && g_mode.load() != FA_MODE_F32_GENERIC)
        need += fa::bwd_ds_extra_bytes(bh, n, d, dtype, causal != 0, bwd_atomic_variant());
    return need;
}

size_t fa3_forward_workspace_bytes(int64_t bh, int64_t n, int64_t d, int dtype, int fp8) {
    if (!fp8 || bh <= 0 || n <= 0 || d <= 0 || !fa::fwd_mfma_supported(dtype, d) || !fa::bwd_mfma_supported(dtype, d)) return 0;
    if (fa::fwd_fp8_supported(dtype, d)) return slab_bytes(bh, n, d) + fa::fwd_fp8_workspace_bytes(bh, n, d);
    return 3 * slab_bytes(bh, n, d);
}

int fa_debug_trace_buffer(void* device_ptr) {
    return fa::set_trace_buffer(device_ptr) == hipSuccess ? FA_OK : fail(FA_ERR_LAUNCH, "fa_debug_trace_buffer: hipMemcpyToSymbol failed");
}

int fa_set_option(const char* name, int value) {
    if (!name || fa::set_option(name, value) != 0) return fail(FA_ERR_INVALID_ARGUMENT, "fa_set_option: unknown option '%s'", name ? name : "(null)");
    return FA_OK;
}

const char* fa_last_error(void) { return g_err; }
const char* fa_version(void) { return "fa_mi355x 0.1.0 (gfx950)"; }

int fa_set_kernel_mode(int mode) {
    if (mode != FA_MODE_AUTO && mode != FA_MODE_F32_GENERIC && mode != FA_MODE_BWD_ATOMIC) return fail(FA_ERR_INVALID_ARGUMENT, "fa_set_kernel_mode: bad mode %d", mode);
    return g_mode.exchange(mode);
}

int fa_device_is_gfx950(int device) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

}  // extern "C"
