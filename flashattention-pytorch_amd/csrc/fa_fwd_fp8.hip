// FA3-style fp8 forward for gfx950: Q and K quantised to OCP e4m3 with one scale per (b,h, 64-row block),
// S = Q K^T on v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands, unit hardware scales; 2x the bf16 rate), softmax / P / V / O exactly as the 16-bit forward (P and V stay 16 bit,
// fp32 accumulation).  head_dim 128, f16 / bf16 tensors.
//
// What it replaces: the fp8 branch of csrc/fa3/fa3_fwd.cu:196-208 (= src/fa3/torch/impl.py:118-133): per-block absmax
// scales (block_absmax_scale, :70-85, eps 1e-6) and a quantise step.  The reference's quantise is an fp16 round trip
// that models no 8-bit rounding, and its "Hadamard" rotation is not orthogonal; this implements the
// intent — an orthogonal sign + Hadamard rotation, then real e4m3 with block scales — and is held to the reference's fp8 bar (1e-1,
// tests/test_correctness_fa3.py:31-32) against the oracle's e4m3 model (oracle.fp8_attention) and the exact result.
//
// Two launches: fp8_quant_kernel (q and k -> e4m3 bytes + scales in the caller's workspace), fwd_fp8_kernel.
// The e4m3 K tile is 64 keys x 128 BYTES: byte-for-byte the geometry of a 16-bit d=64 tile, so it reuses that LDS
// image and LDS-DMA path (TileSwz<64>, dma_stage_tile<64,...>).  Each lane reads two 16-byte chunks of a K row per
// MFMA; the Q fragment in registers uses the same byte order, so the k index matches.
#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

constexpr float kE4M3Max = 448.f;

// one workgroup = one 64-row block of one (b,h) of one tensor (blockIdx.z: 0 = q, 1 = k)
// OUT16 = false: write e4m3 bytes + scales (forward).  OUT16 = true: write the DEQUANTISED values back as 16-bit
// tensors (backward: the gradient is taken of the function the forward actually evaluated, i.e. with the
// quantised Q and K, which is also what the reference's fa3_backward does, csrc/fa3/fa3_bwd.cu:134-146).
// ROT: incoherent processing first — every row x of Q and K becomes x . diag(s) . H / sqrt(D) (s a fixed +-1 vector,
// H the Sylvester Hadamard matrix), the rotation the reference's FA3 path intends (src/fa3/torch/impl.py:20-45, SURVEY
// D6).  It is orthogonal, so S = Q K^T is unchanged, and it spreads an outlier channel over all D channels so that one
// e4m3 scale per 64-row block wastes no range on it.  A row lives in 16 consecutive lanes (8 elements each): three
// butterfly stages in registers, four across the lanes.  OUT16 writes the round-tripped values back in the ORIGINAL
// basis (H, then the signs: the rotation's transpose).
__device__ __forceinline__ float rot_sign(int e) {
    const unsigned h = ((unsigned)e * 0x9E3779B1u) >> 27;
    return ((0x5A3C96E1u >> h) & 1u) ? -1.f : 1.f;
}
template <int D>
__device__ __forceinline__ void hadamard_row(float (&x)[8], int ch) {   // ch = this lane's 8-element chunk of the row
#pragma unroll
    for (int st = 1; st < 8; st <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!(j & st)) { const float a = x[j], b = x[j | st]; x[j] = a + b; x[j | st] = a - b; }
#pragma unroll
    for (int bit = 1; bit < D / 8; bit <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float other = __shfl_xor(x[j], bit, 64);
            x[j] = (ch & bit) ? other - x[j] : x[j] + other;
        }
}

template <typename Tag, int D, bool OUT16, bool ROT>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                        uint8_t* __restrict__ q8, uint8_t* __restrict__ k8,
                                                        float* __restrict__ sq, float* __restrict__ sk, int n, int nb) {
    constexpr int CPR = D / 8, PER = (64 * CPR) / 256;  // 16-byte chunks per thread
    __shared__ float red[4];
    const uint16_t* src = blockIdx.z == 0 ? q : k;
    uint8_t* dst = blockIdx.z == 0 ? q8 : k8;
    float* sc = blockIdx.z == 0 ? sq : sk;
    const int bh = blockIdx.y, blk = blockIdx.x, row0 = blk * 64;
    const size_t base = (size_t)bh * n * D;
    float xf[PER][8];
    float amax = 0.f;
    const float rs = 0.08838834764831845f * (D == 128 ? 1.f : 0.f);   // 1 / sqrt(128); the fp8 path is built for D = 128
    static_assert(D == 128, "fp8 path: head_dim 128");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = threadIdx.x + 256 * i, row = row0 + c / CPR, ch = c % CPR;
        u32x4 x = u32x4{0u, 0u, 0u, 0u};
        if (row < n) x = *reinterpret_cast<const u32x4*>(src + base + (size_t)row * D + 8 * ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xf[i][2 * j] = unpack_lo<Tag>(x[j]); xf[i][2 * j + 1] = unpack_hi<Tag>(x[j]); }
        if (ROT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[i][j] *= rot_sign(8 * ch + j);
            hadamard_row<D>(xf[i], ch);
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[i][j] *= rs;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(xf[i][j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    amax = fmaxf(amax, 1e-6f);                       // eps of block_absmax_scale
    const float inv = kE4M3Max / amax;
    const float scl = amax / kE4M3Max;
    if (!OUT16 && threadIdx.x == 0) sc[(size_t)bh * nb + blk] = scl;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = threadIdx.x + 256 * i, row = row0 + c / CPR, ch = c % CPR;
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][0] * inv, xf[i][1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][2] * inv, xf[i][3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][4] * inv, xf[i][5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][6] * inv, xf[i][7] * inv, w1, true);
        if (!OUT16) {
            if (row >= n) continue;
            u32x2 o2 = {(unsigned)w0, (unsigned)w1};
            *reinterpret_cast<u32x2*>(dst + base + (size_t)row * D + 8 * ch) = o2;
        } else {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t a0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
            const f32x2_t a2 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
            float y[8] = {a0[0] * scl, a0[1] * scl, a1[0] * scl, a1[1] * scl, a2[0] * scl, a2[1] * scl, a3[0] * scl, a3[1] * scl};
            if (ROT) {   // back to the original basis (every lane takes part in the shuffles, rows past n included)
                hadamard_row<D>(y, ch);
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] *= rs * rot_sign(8 * ch + j);
            }
            if (row >= n) continue;
            u32x4 o4;
            o4[0] = pack2_rn<Tag>(y[0], y[1]);
            o4[1] = pack2_rn<Tag>(y[2], y[3]);
            o4[2] = pack2_rn<Tag>(y[4], y[5]);
            o4[3] = pack2_rn<Tag>(y[6], y[7]);
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(dst) + base + (size_t)row * D + 8 * ch) = o4;
        }
    }
}

// The e4m3 round trip of whole tensors, any head dim that is a multiple of 8 (up to 256): x -> dequantise(quantise(x)) with one
// absmax scale per (b,h, 64-row block), written back as 16-bit tensors.  This is the reference's own wiring of its fp8 mode —
// quantise-dequantise Q, K AND V ahead of the unchanged inner loops (csrc/fa3/fa3_fwd.cu:196-208, fa3_bwd.cu:134-146,
// src/fa3/torch/impl.py:123-131) — with a real 8-bit rounding in place of its fp16 no-op.  ROT (Q and K, power-of-two
// head dims: src/fa3/torch/impl.py:60-61 skips the others too): rotate, quantise, and rotate back, so that the round-tripped
// tensors live in the original basis and any kernel can consume them: (Q~ R^T)(K~ R^T)^T = Q~ K~^T.
// Serves: fa3_forward(fp8) for head dims without an e4m3 MFMA kernel (then the 16-bit kernels run on Q~, K~, V~), the V of the
// d = 128 kernel's bf16 P.V variant, and fa3_backward(fp8) at every head dim (it differentiates the function the forward evaluated).
struct Fp8RtArgs {
    const uint16_t* src[3];
    uint16_t* dst[3];
    int rot[3];
};
template <typename Tag, int MAXI>               // MAXI: 16-byte chunks per thread = ceil(64 rows x d / 8 chunks / 256 threads): 1, 2, 4, 8
__global__ __launch_bounds__(256) void fp8_roundtrip_kernel(Fp8RtArgs a, int n, int d) {
    __shared__ float red[4];
    const int z = blockIdx.z, bh = blockIdx.y, row0 = blockIdx.x * 64;
    const uint16_t* src = a.src[z];
    uint16_t* dst = a.dst[z];
    const bool rot = a.rot[z] != 0;             // the host sets it only for power-of-two d: a row is then cpr <= 32 consecutive lanes
    const int cpr = d >> 3, total = 64 * cpr;   // 16-byte chunks per row / per block
    const size_t base = (size_t)bh * n * d;
    const float rs = rsqrtf((float)d);
    float xf[MAXI][8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = threadIdx.x + 256 * i, row = row0 + c / cpr, ch = c % cpr;
        u32x4 x = u32x4{0u, 0u, 0u, 0u};
        if (c < total && row < n) x = *reinterpret_cast<const u32x4*>(src + base + (size_t)row * d + 8 * ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xf[i][2 * j] = unpack_lo<Tag>(x[j]); xf[i][2 * j + 1] = unpack_hi<Tag>(x[j]); }
        if (rot) {   // block-uniform branch; every lane of the wave takes part in the shuffles
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[i][j] *= rot_sign(8 * ch + j);
#pragma unroll
            for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (!(j & st)) { const float p = xf[i][j], m = xf[i][j | st]; xf[i][j] = p + m; xf[i][j | st] = p - m; }
            for (int bit = 1; bit < cpr; bit <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float other = __shfl_xor(xf[i][j], bit, 64);
                    xf[i][j] = (ch & bit) ? other - xf[i][j] : xf[i][j] + other;
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[i][j] *= rs;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(xf[i][j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-6f);   // eps of block_absmax_scale
    const float inv = kE4M3Max / amax, scl = amax / kE4M3Max;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = threadIdx.x + 256 * i, row = row0 + c / cpr, ch = c % cpr;
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][0] * inv, xf[i][1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][2] * inv, xf[i][3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][4] * inv, xf[i][5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][6] * inv, xf[i][7] * inv, w1, true);
        const f32x2_t a0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
        const f32x2_t a2 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
        float y[8] = {a0[0] * scl, a0[1] * scl, a1[0] * scl, a1[1] * scl, a2[0] * scl, a2[1] * scl, a3[0] * scl, a3[1] * scl};
        if (rot) {   // back to the original basis: H, then the signs (the rotation's transpose)
#pragma unroll
            for (int st = 1; st < 8; st <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (!(j & st)) { const float p = y[j], m = y[j | st]; y[j] = p + m; y[j | st] = p - m; }
            for (int bit = 1; bit < cpr; bit <<= 1)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float other = __shfl_xor(y[j], bit, 64);
                    y[j] = (ch & bit) ? other - y[j] : y[j] + other;
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] *= rs * rot_sign(8 * ch + j);
        }
        if (c >= total || row >= n) continue;
        u32x4 o4;
        o4[0] = pack2_rn<Tag>(y[0], y[1]);
        o4[1] = pack2_rn<Tag>(y[2], y[3]);
        o4[2] = pack2_rn<Tag>(y[4], y[5]);
        o4[3] = pack2_rn<Tag>(y[6], y[7]);
        *reinterpret_cast<u32x4*>(dst + base + (size_t)row * d + 8 * ch) = o4;
    }
}

template <typename Tag, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void fwd_fp8_kernel(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8,
                                                         const float* __restrict__ sq, const float* __restrict__ sk,
                                                         const uint16_t* __restrict__ v, uint16_t* __restrict__ o,
                                                         float* __restrict__ lse, int n, int nqt, int nb, float c_log2) {
    constexpr int D = 128, BM = 256, BN = 64, NM = D / 32, NDV = D / 32;
    constexpr int K_BYTES = BN * D, V_BYTES = BN * D * 2, BUF = K_BYTES + V_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K8 tile | V tile]

    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nqt;
    int qt = L - bh * nqt;
    if (CAUSAL) qt = nqt - 1 - qt;
    const int q0 = qt * BM;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int qrow = q0 + 32 * w + r;
    const size_t base = (size_t)bh * n * D;

    // ---- Q fragments: 16 e4m3 bytes per 32-wide d block: Q8[qrow][32 m + 16 h .. +15]
    const buf_rsrc_t q_rs = make_rsrc(q8 + base, (unsigned)n * D);
    u32x4 qf[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) qf[m] = __builtin_amdgcn_raw_buffer_load_b128(q_rs, qrow * D + 32 * m + 16 * h, 0, 0);
    const int qblk = min((q0 + 32 * w) / 64, nb - 1);
    const float fq = sq[(size_t)bh * nb + qblk] * c_log2;   // q-block scale folded with softmax_scale * log2(e)

    const int kend = CAUSAL ? min(n, q0 + BM) : n;
    const int ntiles = (kend + BN - 1) / BN;

    const rsrc_s_t k_rs = make_rsrc_s(k8 + base, (unsigned)n * D);
    const rsrc_s_t v_rs = make_rsrc_s(v + base, (unsigned)n * D * 2);
    const int voff_k = dma_lane_voff<64>(lane, w);    // 128-byte rows: the geometry of a 16-bit d = 64 tile
    const int voff_v = dma_lane_voff<D>(lane, w);
    auto stage = [&](int buf, int k0) {
        char* b_ = smem + buf * BUF;
        dma_stage_tile<64, BN, 8>(k_rs, b_, k0, voff_k, w);
        dma_stage_tile<D, BN, 8>(v_rs, b_ + K_BYTES, k0, voff_v, w);
    };

    f32x16 oacc[NDV];
#pragma unroll
    for (int t = 0; t < NDV; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;   // running max in log2 units of the scaled score

    stage(0, 0);
    dma_wait_all();
    __syncthreads();
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    const int ntiles_w = CAUSAL ? min(ntiles, (q0 + 32 * w + 31) / BN + 1) : ntiles;

    for (int t = 0; t < ntiles_w; ++t) {
        const int k0 = t * BN, cur = t & 1;
        if (t + 1 < ntiles) stage(cur ^ 1, k0 + BN);
        const char* Kt = smem + cur * BUF;
        const char* Vt = Kt + K_BYTES;
        const float f = fq * sk[(size_t)bh * nb + t];   // one K scale per 64-key tile
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
            // v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit block scales (E8M0 127): 64 of the 128
            // head-dim bytes per instruction at twice the bf16 rate per clock.  Lane (r, h) supplies 32 bytes of row r
            // as k = 32 h .. 32 h + 31 (layout probed with tools/ubench/probe_f8f6f4.hip); which 32 bytes is free as
            // long as K and Q agree: here the two 16-byte chunks 4M + h and 4M + 2 + h of the row.
            typedef int i32x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int M = 0; M < NM / 2; ++M) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(Kt + TileSwz<64>::off(32 * kb + r, 4 * M + h));
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(Kt + TileSwz<64>::off(32 * kb + r, 4 * M + 2 + h));
                const i32x8_t a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
                const u32x4 b0 = qf[2 * M], b1 = qf[2 * M + 1];
                const i32x8_t b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
                sacc[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, sacc[kb], 0, 0, 0, 127, 0, 127);
            }
        }
        const bool need_mask = (CAUSAL && (k0 + BN - 1 > q0 + 32 * w)) || (k0 + BN > n);
        const int lim = CAUSAL ? min(qrow, n - 1) : n - 1;
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int thr = need_mask ? lim - (k0 + 32 * kb + 4 * h) : 64;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float s = ((i & 3) + 8 * (i >> 2) > thr) ? -INFINITY : sacc[kb][i] * f;
                sacc[kb][i] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = fmaxf(mx, wave_half_swap(mx));
        const float m_new = fmaxf(m_run, mx);
        // lazy rescale as in fa_fwd_mfma.hip: O and l move to the new max only when some row of the wave grew by > 2^8
        float m_use;
        if (__any(m_new - m_run > 8.0f) != 0) {   // m_run = -inf on the first tile -> true
            m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int t2 = 0; t2 < NDV; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[t2][i] *= alpha;
        } else {
            m_use = m_run;
        }
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(sacc[kb][i] - m_use);
                sacc[kb][i] = p;
                rs += p;
            }
        l_run += rs;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = pack2<Tag>(sacc[kb][8 * s + 2 * j], sacc[kb][8 * s + 2 * j + 1]);
                const s16x8 pb = *reinterpret_cast<s16x8*>(&pk);
                const int key_a = 32 * kb + 16 * s + 4 * h + tq;
#pragma unroll
                for (int dvb = 0; dvb < NDV; ++dvb) {
                    const int ch = 4 * dvb + 2 * g16 + (tp >> 1);
                    const s16x8 a = cat8(lds_tr16(Vt + TileSwz<D>::off(key_a, ch) + 8 * (tp & 1)),
                                         lds_tr16(Vt + TileSwz<D>::off(key_a + 8, ch) + 8 * (tp & 1)));
                    oacc[dvb] = mfma32<Tag>(a, pb, oacc[dvb]);
                }
            }
        dma_wait_all();
        __syncthreads();
    }
    for (int t = ntiles_w; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage((t & 1) ^ 1, (t + 1) * BN);
        dma_wait_all();
        __syncthreads();
    }

    const float l_tot = l_run + wave_half_swap(l_run);
    if (qrow < n) {
        const float inv = 1.f / l_tot;
        uint16_t* orow = o + base + (size_t)qrow * D;
#pragma unroll
        for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk;
                pk[0] = pack2_rn<Tag>(oacc[dvb][4 * g + 0] * inv, oacc[dvb][4 * g + 1] * inv);
                pk[1] = pack2_rn<Tag>(oacc[dvb][4 * g + 2] * inv, oacc[dvb][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(orow + 32 * dvb + 8 * g + 4 * h) = pk;
            }
        if (h == 0) lse[(size_t)bh * n + qrow] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    }
}

// ---- the all-e4m3 forward (default at d = 128; option fp8_pv = 1 selects the kernel above with its 16-bit P.V) -------------
// V goes through e4m3 as well and P.V runs on the block-scaled fp8 MFMA too: per 128-key tile and wave 8 + 8 MFMAs of 64 cycles
// where the 16-bit kernel issues 32 + 32 of 32 cycles — half the matrix time, which leaves the kernel bound by the softmax's
// vector work.  What that needs:
//   * V^T as the A operand of O^T += V^T P^T, 32 contiguous bytes (= 32 keys) per lane: fp8_quant_v_kernel writes V TRANSPOSED,
//     tile by tile — [b,h][128-key tile][d row 0..127][128 key bytes], one tile = 16 KiB contiguous = byte for byte the geometry of
//     the K tile — with the keys of every 64-key group permuted into the order in which the S^T accumulators hold them:
//     byte 32 hi + 16 h + jj of a group is key 32 hi + 4 h + (jj & 3) + 8 (jj >> 2).  The e4m3 P of two 32-key blocks (registers
//     0..15 of block 2g, then of block 2g+1) is then the B operand as it stands, and lane (d row, h) reads chunks 4g + h and
//     4g + 2 + h of its row exactly as the S product reads its K rows.
//   * one V scale per 64-key block that the MFMA can apply itself: a power of two (2^e >= absmax / 448), handed to the
//     instruction as the E8M0 block scale of its A operand.  Keys past N are written as zeros (their P is 0: 0 x 0).
//   * P in e4m3: p = exp2(s - m) lies in (0, 256] under the lazy rescale, inside e4m3's range (448); the row sum and lse come
//     from the f32 p, so only O sees the 3-bit mantissa — 2^-4 relative per probability, which averages out over the keys of a
//     row: harmless from a few dozen keys on, up to 6 % of |v| on a row that sees two.  Rows that short exist only in the FIRST
//     query tile under the causal mask (every later tile's rows see >= 256 keys) and on problems of N <= 256: those run on the
//     kernel above with its 16-bit P (launch_fp8_t), so that o stays what the backward — which recomputes P exactly from lse —
//     differentiates (the delta = rowsum(dO o) it takes from the forward's o otherwise puts the same 6 % into dQ).
template <typename Tag>
__global__ __launch_bounds__(256) void fp8_quant_v_kernel(const uint16_t* __restrict__ v, uint8_t* __restrict__ v8t,
                                                          int* __restrict__ svexp, int n, int nb, int ntile) {
    constexpr int D = 128, CPR = D / 8, PER = (64 * CPR) / 256;
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) uint8_t tr[D * 64];   // [d row][permuted key position]
    const int bh = blockIdx.y, blk = blockIdx.x, row0 = blk * 64;
    const size_t base = (size_t)bh * n * D;
    float xf[PER][8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = threadIdx.x + 256 * i, row = row0 + c / CPR, ch = c % CPR;
        u32x4 x = u32x4{0u, 0u, 0u, 0u};
        if (row < n) x = *reinterpret_cast<const u32x4*>(v + base + (size_t)row * D + 8 * ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xf[i][2 * j] = unpack_lo<Tag>(x[j]); xf[i][2 * j + 1] = unpack_hi<Tag>(x[j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(xf[i][j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-6f);
    // smallest power of two 2^e with amax / 2^e <= 448 (exponent arithmetic on the f32 bits: exact)
    int e = (int)((__float_as_uint(amax / kE4M3Max) >> 23) & 0xff) - 127;
    if (__uint_as_float((unsigned)(e + 127) << 23) * kE4M3Max < amax) ++e;
    e = max(-126, min(126, e));
    const float inv = __uint_as_float((unsigned)(127 - e) << 23);   // 2^-e
    if (threadIdx.x == 0) svexp[(size_t)bh * nb + blk] = e + 127;   // E8M0
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = threadIdx.x + 256 * i, key = c / CPR, ch = c % CPR;   // key within the 64-key block
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][0] * inv, xf[i][1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][2] * inv, xf[i][3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][4] * inv, xf[i][5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(xf[i][6] * inv, xf[i][7] * inv, w1, true);
        const int wlo = key & 31, pos = 32 * (key >> 5) + 16 * ((wlo >> 2) & 1) + (wlo & 3) + 4 * (wlo >> 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) tr[(8 * ch + j) * 64 + pos] = (uint8_t)(((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xff);
    }
    __syncthreads();
    // d row r of the block: 64 bytes at [tile blk / 2][r][64 (blk & 1) ...]; a thread stores half a row
    const int r = threadIdx.x >> 1, half = threadIdx.x & 1;
    uint8_t* dst = v8t + (((size_t)bh * ntile + (blk >> 1)) * D + r) * 128 + 64 * (blk & 1) + 32 * half;
    const u32x4* src = reinterpret_cast<const u32x4*>(tr + r * 64 + 32 * half);
    reinterpret_cast<u32x4*>(dst)[0] = src[0];
    reinterpret_cast<u32x4*>(dst)[1] = src[1];
    // an odd block count leaves the second half of the last tile unwritten by any block: zero it (those keys lie past N)
    if ((nb & 1) && blk == nb - 1) {
        reinterpret_cast<u32x4*>(dst + 64)[0] = u32x4{0u, 0u, 0u, 0u};
        reinterpret_cast<u32x4*>(dst + 64)[1] = u32x4{0u, 0u, 0u, 0u};
    }
}

template <typename Tag, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void fwd_fp8v_kernel(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8,
                                                          const uint8_t* __restrict__ v8t, const float* __restrict__ sq,
                                                          const float* __restrict__ sk, const int* __restrict__ svexp,
                                                          uint16_t* __restrict__ o, float* __restrict__ lse, int n, int nqt,
                                                          int nb, int ntile, float c_log2, int nqt_run /* query tiles this launch covers: under the causal mask tiles nqt - 1 .. nqt - nqt_run */) {
    constexpr int D = 128, BM = 256, BN = 128, KB = 4, NM = D / 32, NDV = D / 32;
    constexpr int TILE = BN * D;                         // bytes of a K tile and of a V^T tile alike
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K8 tile | V8^T tile]
    typedef int i32x8_t __attribute__((ext_vector_type(8)));

    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nqt_run;
    int qt = L - bh * nqt_run;
    if (CAUSAL) qt = nqt - 1 - qt;
    const int q0 = qt * BM;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int qrow = q0 + 32 * w + r;
    const size_t base = (size_t)bh * n * D;

    const buf_rsrc_t q_rs = make_rsrc(q8 + base, (unsigned)n * D);
    u32x4 qf[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) qf[m] = __builtin_amdgcn_raw_buffer_load_b128(q_rs, qrow * D + 32 * m + 16 * h, 0, 0);
    const int qblk = min((q0 + 32 * w) / 64, nb - 1);
    const float fq = sq[(size_t)bh * nb + qblk] * c_log2;   // q-block scale folded with softmax_scale * log2(e)

    const int kend = CAUSAL ? min(n, q0 + BM) : n;
    const int ntiles = (kend + BN - 1) / BN;
    const rsrc_s_t k_rs = make_rsrc_s(k8 + base, (unsigned)n * D);
    const rsrc_s_t v_rs = make_rsrc_s(v8t + (size_t)bh * ntile * TILE, (unsigned)ntile * TILE);
    const int voff = dma_lane_voff<64>(lane, w);         // 128-byte rows: the geometry of a 16-bit d = 64 tile
    auto stage = [&](int buf, int t) {
        char* b_ = smem + buf * 2 * TILE;
        dma_stage_tile<64, BN, 8>(k_rs, b_, t * BN, voff, w);
        dma_stage_tile<64, BN, 8>(v_rs, b_ + TILE, t * BN, voff, w);   // V^T tile t = rows 128 t .. of the [tile][d row] matrix
    };

    f32x16 oacc[NDV];
#pragma unroll
    for (int t = 0; t < NDV; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                // running max in log2 units of the scaled score

    stage(0, 0);
    dma_wait_all();
    __syncthreads();
    const int ntiles_w = CAUSAL ? min(ntiles, (q0 + 32 * w + 31) / BN + 1) : ntiles;

    for (int t = 0; t < ntiles_w; ++t) {
        const int k0 = t * BN, cur = t & 1;
        if (t + 1 < ntiles) stage(cur ^ 1, t + 1);
        const char* Kt = smem + cur * 2 * TILE;
        const char* Vt = Kt + TILE;
        const size_t sb = (size_t)bh * nb;
        const int b0 = min(2 * t, nb - 1), b1 = min(2 * t + 1, nb - 1);   // the tile's two 64-key blocks (the second may lie past N)
        const float f0 = fq * sk[sb + b0], f1 = fq * sk[sb + b1];
        const int e0 = svexp[sb + b0], e1 = svexp[sb + b1];
        f32x16 sacc[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
            for (int M = 0; M < NM / 2; ++M) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(Kt + TileSwz<64>::off(32 * kb + r, 4 * M + h));
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(Kt + TileSwz<64>::off(32 * kb + r, 4 * M + 2 + h));
                const i32x8_t a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
                const u32x4 b0_ = qf[2 * M], b1_ = qf[2 * M + 1];
                const i32x8_t b = {(int)b0_[0], (int)b0_[1], (int)b0_[2], (int)b0_[3], (int)b1_[0], (int)b1_[1], (int)b1_[2], (int)b1_[3]};
                sacc[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, sacc[kb], 0, 0, 0, 127, 0, 127);
            }
        }
        // mask and row maximum on the RAW products (the block scales f0, f1 > 0 are applied to the two maxima, and to every
        // element inside the exp2 argument's fma: one vector instruction per element less than scaling first)
        const bool need_mask = (CAUSAL && (k0 + BN - 1 > q0 + 32 * w)) || (k0 + BN > n);
        const int lim = CAUSAL ? min(qrow, n - 1) : n - 1;
        float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (need_mask) {
                const int thr = lim - (k0 + 32 * kb + 4 * h);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) > thr) sacc[kb][i] = -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (kb < 2) mx0 = fmaxf(mx0, sacc[kb][i]);
                else mx1 = fmaxf(mx1, sacc[kb][i]);
            }
        }
        float mx = fmaxf(mx0 * f0, mx1 * f1);
        mx = fmaxf(mx, wave_half_swap(mx));
        const float m_new = fmaxf(m_run, mx);
        float m_use;
        if (__any(m_new - m_run > 8.0f) != 0) {   // lazy rescale (fa_fwd_mfma.hip); m_run = -inf on the first tile -> true
            m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int t2 = 0; t2 < NDV; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[t2][i] *= alpha;
        } else {
            m_use = m_run;
        }
        float rs = 0.f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            i32x8_t pb;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int kb = 2 * g + kk;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], g == 0 ? f0 : f1, -m_use));   // (-inf stays -inf: f > 0)
                    sacc[kb][i] = p;
                    rs += p;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int wd = 0;
                    wd = __builtin_amdgcn_cvt_pk_fp8_f32(sacc[kb][4 * j + 0], sacc[kb][4 * j + 1], wd, false);
                    wd = __builtin_amdgcn_cvt_pk_fp8_f32(sacc[kb][4 * j + 2], sacc[kb][4 * j + 3], wd, true);
                    pb[4 * kk + j] = wd;
                }
            }
            const int se = g == 0 ? e0 : e1;
#pragma unroll
            for (int dvb = 0; dvb < NDV; ++dvb) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(Vt + TileSwz<64>::off(32 * dvb + r, 4 * g + h));
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(Vt + TileSwz<64>::off(32 * dvb + r, 4 * g + 2 + h));
                const i32x8_t a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
                oacc[dvb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, pb, oacc[dvb], 0, 0, 0, se, 0, 127);
            }
        }
        l_run += rs;
        dma_wait_all();
        __syncthreads();
    }
    for (int t = ntiles_w; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage((t & 1) ^ 1, t + 1);
        dma_wait_all();
        __syncthreads();
    }

    // every wave is past the last barrier (the tile loops end with one): the K / V buffers are dead and each wave turns its
    // row-on-the-lane tile into whole-row stores through 8 KiB of them (fa_fwd_mfma.hip's epilogue)
    const float l_tot = l_run + wave_half_swap(l_run);
    {
        const float inv = 1.f / l_tot;
        u32x2 vals[NDV * 4];
#pragma unroll
        for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * dvb + g][0] = pack2_rn<Tag>(oacc[dvb][4 * g + 0] * inv, oacc[dvb][4 * g + 1] * inv);
                vals[4 * dvb + g][1] = pack2_rn<Tag>(oacc[dvb][4 * g + 2] * inv, oacc[dvb][4 * g + 3] * inv);
            }
        store_rows_via_lds<D>(smem + w * 32 * D * 2, vals, o + base, q0 + 32 * w, n, lane, D);
        if (qrow < n && h == 0) lse[(size_t)bh * n + qrow] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
    }
}

bool fwd_fp8_supported(int dtype, int64_t d) { return (dtype == 1 || dtype == 2) && d == 128; }

// workspace: [q8: BH*N*D bytes][k8: BH*N*D bytes][pad][sq: BH*nb floats][sk: BH*nb floats][svexp: BH*nb ints][pad]
//            [v8t: BH * ntile * 128 * 128 bytes], nb = ceil(N/64), ntile = ceil(N/128)
struct Fp8Ws { uint8_t *q8, *k8, *v8t; float *sq, *sk; int* svexp; size_t bytes; };
static Fp8Ws fp8_ws_layout(void* ws, int64_t bh, int64_t n, int64_t d) {
    const size_t nb = (size_t)((n + 63) / 64), ntile = (size_t)((n + 127) / 128), nbytes = (size_t)bh * n * d;
    char* p = reinterpret_cast<char*>(ws);
    Fp8Ws L;
    L.q8 = reinterpret_cast<uint8_t*>(p);
    L.k8 = L.q8 + nbytes;
    size_t off = (2 * nbytes + 255) & ~(size_t)255;
    L.sq = reinterpret_cast<float*>(p + off);
    L.sk = L.sq + (size_t)bh * nb;
    L.svexp = reinterpret_cast<int*>(L.sk + (size_t)bh * nb);
    off = (off + 3 * sizeof(float) * (size_t)bh * nb + 255) & ~(size_t)255;
    L.v8t = reinterpret_cast<uint8_t*>(p + off);
    L.bytes = off + (size_t)bh * ntile * 128 * 128 + 256;
    return L;
}
size_t fwd_fp8_workspace_bytes(int64_t bh, int64_t n, int64_t d) { return fp8_ws_layout(nullptr, bh, n, d).bytes; }

// Q and K -> e4m3 bytes + scales; then the all-e4m3 kernel (V quantised transposed) or, with option fp8_pv = 1, the kernel with
// the 16-bit P.V on the round-tripped V (`vslab`: room for one 16-bit tensor).  Under the causal mask the first query tile, and
// problems of N <= 256 altogether, take the 16-bit P.V in either case (see the comment above fp8_quant_v_kernel).
template <typename Tag>
static hipError_t launch_fp8_t(const FwdArgs& a, void* ws, void* vslab, hipStream_t st) {
    constexpr int D = 128;
    const int nb = (int)((a.n + 63) / 64), ntile = (int)((a.n + 127) / 128);
    const int nqt = (int)((a.n + 255) / 256);
    const Fp8Ws L = fp8_ws_layout(ws, a.bh, a.n, D);
    const bool pv16 = option(OPT_FP8_PV) == 1 || a.n <= 256;
    const int nqt16 = pv16 ? nqt : (a.causal ? 1 : 0);          // query tiles on the 16-bit P.V kernel (the first ones)
    const int64_t vrows = pv16 ? a.n : (a.causal ? (a.n < 256 ? a.n : 256) : 0);   // rows of V they read
    {
        ProfScope ps(K_FP8_QUANT, st);
        if (option(OPT_FP8_ROT) == 2)   // option fp8_rot = 2: quantise without the incoherent rotation (A/B, tests)
            hipLaunchKernelGGL((fp8_quant_kernel<Tag, D, false, false>), dim3(nb, (unsigned)a.bh, 2), dim3(256), 0, st, (const uint16_t*)a.q,
                               (const uint16_t*)a.k, L.q8, L.k8, L.sq, L.sk, (int)a.n, nb);
        else
            hipLaunchKernelGGL((fp8_quant_kernel<Tag, D, false, true>), dim3(nb, (unsigned)a.bh, 2), dim3(256), 0, st, (const uint16_t*)a.q,
                               (const uint16_t*)a.k, L.q8, L.k8, L.sq, L.sk, (int)a.n, nb);
        if (!pv16)
            hipLaunchKernelGGL(fp8_quant_v_kernel<Tag>, dim3(nb, (unsigned)a.bh), dim3(256), 0, st, (const uint16_t*)a.v, L.v8t, L.svexp,
                               (int)a.n, nb, ntile);
        if (vrows > 0) {   // V~ (the first `vrows` rows of every (b,h)) for the 16-bit P.V kernel
            Fp8RtArgs r{};
            r.src[0] = (const uint16_t*)a.v; r.dst[0] = (uint16_t*)vslab; r.rot[0] = 0;
            hipLaunchKernelGGL((fp8_roundtrip_kernel<Tag, 4>), dim3((unsigned)((vrows + 63) / 64), (unsigned)a.bh, 1), dim3(256), 0, st, r, (int)a.n, D);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const float c = a.scale * 1.4426950408889634f;
    ProfScope ps(K_FWD_FP8, st);
    if (nqt16 > 0) {
        const size_t smem = 2 * (64 * D + 64 * D * 2);
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e2 = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
            if (e2 != hipSuccess) return e2;
            // (nqt16 < nqt only under the causal mask: the kernel then maps its blocks to tiles nqt16 - 1 .. 0)
            hipLaunchKernelGGL(kern, dim3((unsigned)(nqt16 * a.bh)), dim3(512), smem, st, (const uint8_t*)L.q8, (const uint8_t*)L.k8,
                               (const float*)L.sq, (const float*)L.sk, (const uint16_t*)vslab, (uint16_t*)a.o, a.lse, (int)a.n, nqt16, nb, c);
            return hipGetLastError();
        };
        e = a.causal ? launch(fwd_fp8_kernel<Tag, true>) : launch(fwd_fp8_kernel<Tag, false>);
        if (e != hipSuccess) return e;
    }
    if (nqt16 < nqt) {
        const size_t smem = 2 * 2 * 128 * D;
        const int nrun = nqt - nqt16;
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e2 = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(kern, dim3((unsigned)(nrun * a.bh)), dim3(512), smem, st, (const uint8_t*)L.q8, (const uint8_t*)L.k8,
                               (const uint8_t*)L.v8t, (const float*)L.sq, (const float*)L.sk, (const int*)L.svexp, (uint16_t*)a.o, a.lse,
                               (int)a.n, nqt, nb, ntile, c, nrun);
            return hipGetLastError();
        };
        e = a.causal ? launch(fwd_fp8v_kernel<Tag, true>) : launch(fwd_fp8v_kernel<Tag, false>);
    }
    return e;
}

// q, k, v -> their e4m3 round trips as 16-bit tensors qt, kt, vt (a null source is skipped).  Q and K are rotated around the
// quantisation when the head dim is a power of two (option fp8_rot = 2: never), V never is.
hipError_t launch_fp8_roundtrip(const void* q, const void* k, const void* v, void* qt, void* kt, void* vt, int64_t bh, int64_t n,
                                int64_t d, int dtype, hipStream_t st) {
    const int nb = (int)((n + 63) / 64);
    const bool rot = option(OPT_FP8_ROT) != 2 && (d & (d - 1)) == 0;
    Fp8RtArgs a{};
    int cnt = 0;
    const void* srcs[3] = {q, k, v};
    void* dsts[3] = {qt, kt, vt};
    for (int i = 0; i < 3; ++i)
        if (srcs[i]) { a.src[cnt] = (const uint16_t*)srcs[i]; a.dst[cnt] = (uint16_t*)dsts[i]; a.rot[cnt] = (i < 2 && rot) ? 1 : 0; ++cnt; }
    if (!cnt || bh <= 0 || n <= 0) return hipSuccess;
    ProfScope ps(K_FP8_QUANT, st);
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(nb, (unsigned)bh, cnt), dim3(256), 0, st, a, (int)n, (int)d); };
    if (dtype == 2) { if (d <= 32) go(fp8_roundtrip_kernel<bf16_tag, 1>); else if (d <= 64) go(fp8_roundtrip_kernel<bf16_tag, 2>); else if (d <= 128) go(fp8_roundtrip_kernel<bf16_tag, 4>); else go(fp8_roundtrip_kernel<bf16_tag, 8>); }
    else { if (d <= 32) go(fp8_roundtrip_kernel<f16_tag, 1>); else if (d <= 64) go(fp8_roundtrip_kernel<f16_tag, 2>); else if (d <= 128) go(fp8_roundtrip_kernel<f16_tag, 4>); else go(fp8_roundtrip_kernel<f16_tag, 8>); }
    return hipGetLastError();
}

hipError_t launch_fwd_fp8(const FwdArgs& a, void* workspace, void* vslab, hipStream_t st) {
    return a.dtype == 2 ? launch_fp8_t<bf16_tag>(a, workspace, vslab, st) : launch_fp8_t<f16_tag>(a, workspace, vslab, st);
}

}  // namespace fa
