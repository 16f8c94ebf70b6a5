// Extended attention forward / backward for gfx950: the extras the reference's notebook model wires around its tiled
// attention (src/fa3/torch/flashattention_pytorch.py), as kernel features behind one entry point:
//   * Nq != Nk with the causal mask aligned bottom-right: key j is visible to query i iff j <= i + (Nk - Nq)
//     (look_ahead_mask_, flashattention_pytorch.py:176-190);
//   * a dense mask [Nq][Nk] of bytes, 0 = masked (scores.masked_fill(mask == 0, -inf), :139-141), shared by all (b,h) or
//     one per (b,h);
//   * a block-sparse mask [ceil(Nq/br)][ceil(Nk/bc)], 0 = the tile is skipped (Algorithm 5, line 8: :123-125);
//   * dropout on the attention probabilities, in-kernel: keep where rnd > p, scale 1/(1-p) (src/common/dropout.py:9-15,
//     flashattention_pytorch.py:85-87), with a counter-based generator so that the backward regenerates the mask from
//     (seed, b*h, i, j) instead of storing it; `tau` (:134) folds into softmax_scale.
// Exact-f32 math on the f32-input MFMA (any dtype in, head_dim <= 256), the structure of fa_generic.hip: forward by
// query tile with online softmax over the visible keys; backward = delta pre-pass + dK/dV kernel + dQ kernel, no
// atomics.  Rows without any visible key get o = 0, lse = -inf (the reference's softmax of an all -inf row is NaN).
// Dropout is the standard one — O = dropout(softmax(S)) V, the denominator counts every visible key — as in the model's
// dense branch (:85-87); its tiled branch renormalises by the sum of the KEPT probabilities (:155-163), which is a
// different function and is not reproduced.
#include "fa_common.h"
#include "fa_ex_common.h"
#include "fa_kernels.h"

namespace fa {

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// rows r0 .. r0 + rows - 1 of a (n, d) matrix into an f32 LDS tile, zero-filled past n and d; `vec` (wave-uniform: d % 4 == 0 and
// the tensor aligned for it): four elements per load (16 bytes of f32, 8 bytes of 16-bit elements) and one 16-byte LDS store
template <typename T, int DP, int LD, int NTHREADS>
__device__ __forceinline__ void ex_load_tile(float* __restrict__ dst, const T* __restrict__ src, int r0, int rows, int n,
                                             int d, bool vec) {
    if (vec) {
        for (int idx = threadIdx.x; idx < rows * (DP / 4); idx += NTHREADS) {
            const int r = idx / (DP / 4), c = 4 * (idx - r * (DP / 4));
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (r0 + r < n && c < d) {
                if constexpr (sizeof(T) == 4) {
                    x = *reinterpret_cast<const f32x4*>(src + (size_t)(r0 + r) * d + c);
                } else {
                    T t[4];
                    *reinterpret_cast<u32x2*>(t) = *reinterpret_cast<const u32x2*>(src + (size_t)(r0 + r) * d + c);
                    x = f32x4{to_f32<T>(t[0]), to_f32<T>(t[1]), to_f32<T>(t[2]), to_f32<T>(t[3])};
                }
            }
            *reinterpret_cast<f32x4*>(dst + r * LD + c) = x;
        }
        return;
    }
    for (int idx = threadIdx.x; idx < rows * DP; idx += NTHREADS) {
        const int r = idx / DP, c = idx - r * DP;
        float x = 0.f;
        if (r0 + r < n && c < d) x = to_f32<T>(src[(size_t)(r0 + r) * d + c]);
        dst[r * LD + c] = x;
    }
}
template <typename T> __device__ __forceinline__ bool ex_quad_ok(int d, const void* a, const void* b, const void* c, const void* e) {
    return d % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                           reinterpret_cast<uintptr_t>(e)) % (4 * sizeof(T))) == 0;
}
// Operand reads (as fa_generic.hip, round 2): the contraction index of the 16x16x4 MFMA is permuted so that a lane's operands for
// four consecutive MFMAs are one 16-byte LDS read — lane (lr, lq), step j of super-step S: k = 16 S + 4 lq + j (A and B agree; the
// sum stays an f32 fma chain).  The P / dS staging areas are wave-private: no workgroup barrier between writing and reading them.

// ---- forward: one workgroup = NW waves = 16 NW query rows of one (b,h); key tiles of 32
template <typename T, int DP, int NW>
__global__ __launch_bounds__(NW * 64) void ex_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, T* __restrict__ o,
                                                         float* __restrict__ lse, ExParams p) {
    constexpr int LD = DP + 4, BM = 16 * NW, BN = 32, NT = DP / 16, PLD = BN + 4, NTH = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Ks = Qs + BM * LD;
    float* Vs = Ks + BN * LD;
    float* Ps = Vs + BN * LD;
    const int ntile = (p.nq + BM - 1) / BM;
    const int bh = blockIdx.x / ntile;
    const int q0 = (blockIdx.x - bh * ntile) * BM;
    const size_t qbase = (size_t)bh * p.nq * p.d, kbase = (size_t)bh * p.nk * p.d;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;

    const bool vec = ex_quad_ok<T>(p.d, q, k, v, q);
    ex_load_tile<T, DP, LD, NTH>(Qs, q + qbase, q0, BM, p.nq, p.d, vec);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m[i] = -INFINITY; l[i] = 0.f; }
    // keys past the last row's diagonal are masked for every row of the tile
    const int kend = p.causal ? max(0, min(p.nk, q0 + BM + p.coff)) : p.nk;
    float* Pw = Ps + w * 16 * PLD;

    for (int k0 = 0; k0 < kend; k0 += BN) {
        if (!ex_tile_live(p, q0, min(q0 + BM, p.nq), k0, min(k0 + BN, p.nk))) continue;   // block-sparse skip (uniform)
        __syncthreads();
        ex_load_tile<T, DP, LD, NTH>(Ks, k + kbase, k0, BN, p.nk, p.d, vec);
        ex_load_tile<T, DP, LD, NTH>(Vs, v + kbase, k0, BN, p.nk, p.d, vec);
        __syncthreads();
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < DP / 16; ++S) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Qs + (w * 16 + lr) * LD + 16 * S + 4 * lq);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Ks + lr * LD + 16 * S + 4 * lq);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Ks + (16 + lr) * LD + 16 * S + 4 * lq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0 = MFMA_F32(a[j], b0[j], s0);
                s1 = MFMA_F32(a[j], b1[j], s1);
            }
        }
        const int key0 = k0 + lr, key1 = k0 + 16 + lr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = q0 + w * 16 + lq * 4 + i;
            float x0 = ex_visible(p, bh, row, key0) ? s0[i] * p.scale : -INFINITY;
            float x1 = ex_visible(p, bh, row, key1) ? s1[i] * p.scale : -INFINITY;
            float mx = fmaxf(x0, x1);
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
            const float mn = fmaxf(m[i], mx);
            const float msafe = (mn == -INFINITY) ? 0.f : mn;
            const float alpha = expf(m[i] - msafe);
            float p0 = expf(x0 - msafe), p1 = expf(x1 - msafe);
            float rs = p0 + p1;     // the softmax denominator counts every visible key, dropped or not
            rs += __shfl_xor(rs, 1, 64);
            rs += __shfl_xor(rs, 2, 64);
            rs += __shfl_xor(rs, 4, 64);
            rs += __shfl_xor(rs, 8, 64);
            l[i] = l[i] * alpha + rs;
            m[i] = mn;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][i] *= alpha;
            if (p.p_drop > 0.f) {
                p0 = ex_keep(p, bh, row, key0) ? p0 * p.keep_scale : 0.f;
                p1 = ex_keep(p, bh, row, key1) ? p1 * p.keep_scale : 0.f;
            }
            Pw[(lq * 4 + i) * PLD + lr] = p0;
            Pw[(lq * 4 + i) * PLD + 16 + lr] = p1;
        }
#pragma unroll
        for (int S = 0; S < BN / 16; ++S) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Pw + lr * PLD + 16 * S + 4 * lq);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = MFMA_F32(a[j], Vs[(16 * S + 4 * lq + j) * LD + 16 * t + lr], acc[t]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = q0 + w * 16 + lq * 4 + i;
        if (row < p.nq) {
            const float inv = l[i] > 0.f ? 1.f / l[i] : 0.f;   // a row without a visible key: o = 0, lse = -inf
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 16 * t + lr;
                if (c < p.d) o[qbase + (size_t)row * p.d + c] = from_f32<T>(acc[t][i] * inv);
            }
            if (lr == 0) lse[(size_t)bh * p.nq + row] = l[i] > 0.f ? m[i] + logf(l[i]) : -INFINITY;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ex_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                       float* __restrict__ delta, long long rows, int d) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    float s = 0.f;
    if (row < rows)
        for (int c = sub; c < d; c += 16) s += to_f32<T>(o[row * d + c]) * to_f32<T>(dout[row * d + c]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (row < rows && sub == 0) delta[row] = s;
}

// ---- backward dK/dV: one workgroup = 16 NW keys resident in LDS; loops over 32-row query tiles
//      dV = P_drop^T dO,  dP = keep/(1-p) * (dO V^T),  dS = P (dP - delta),  dK = scale dS^T Q
template <typename T, int DP, int NW>
__global__ __launch_bounds__(NW * 64) void ex_dkdv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, const T* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          T* __restrict__ dk, T* __restrict__ dv, ExParams p) {
    constexpr int LD = DP + 4, BK = 16 * NW, BQ = 32, NT = DP / 16, PLD = BQ + 4, NTH = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = Ks + BK * LD;
    float* Qs = Vs + BK * LD;
    float* Os = Qs + BQ * LD;
    float* Pt = Os + BQ * LD;
    float* St = Pt + NW * 16 * PLD;
    float* Ls = St + NW * 16 * PLD;
    const int ntile = (p.nk + BK - 1) / BK;
    const int bh = blockIdx.x / ntile;
    const int k0 = (blockIdx.x - bh * ntile) * BK;
    const size_t qbase = (size_t)bh * p.nq * p.d, kbase = (size_t)bh * p.nk * p.d;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;

    const bool vec = ex_quad_ok<T>(p.d, q, k, v, dout);
    ex_load_tile<T, DP, LD, NTH>(Ks, k + kbase, k0, BK, p.nk, p.d, vec);
    ex_load_tile<T, DP, LD, NTH>(Vs, v + kbase, k0, BK, p.nk, p.d, vec);
    f32x4 dka[NT], dva[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { dka[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dva[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float* Pw = Pt + w * 16 * PLD;
    float* Sw = St + w * 16 * PLD;
    // rows before the tile's first key's diagonal see none of it: key k0 is visible from row k0 - coff on
    const int qstart = p.causal ? (max(0, k0 - p.coff) / BQ) * BQ : 0;
    for (int r0 = qstart; r0 < p.nq; r0 += BQ) {
        if (!ex_tile_live(p, r0, min(r0 + BQ, p.nq), k0, min(k0 + BK, p.nk))) continue;
        __syncthreads();
        ex_load_tile<T, DP, LD, NTH>(Qs, q + qbase, r0, BQ, p.nq, p.d, vec);
        ex_load_tile<T, DP, LD, NTH>(Os, dout + qbase, r0, BQ, p.nq, p.d, vec);
        if (threadIdx.x < BQ) {
            const int r = r0 + threadIdx.x;
            Ls[threadIdx.x] = r < p.nq ? lse[(size_t)bh * p.nq + r] : 0.f;
            Ls[BQ + threadIdx.x] = r < p.nq ? delta[(size_t)bh * p.nq + r] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < DP / 16; ++S) {
                const f32x4 ak = *reinterpret_cast<const f32x4*>(Ks + (w * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 av = *reinterpret_cast<const f32x4*>(Vs + (w * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(Qs + (qb * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 bo = *reinterpret_cast<const f32x4*>(Os + (qb * 16 + lr) * LD + 16 * S + 4 * lq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    st = MFMA_F32(ak[j], bq[j], st);
                    dpt = MFMA_F32(av[j], bo[j], dpt);
                }
            }
            const int row = r0 + qb * 16 + lr;   // query index (MFMA column)
            const float lq_ = Ls[qb * 16 + lr], dl_ = Ls[BQ + qb * 16 + lr];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = k0 + w * 16 + lq * 4 + i;
                const float pr = ex_visible(p, bh, row, key) ? expf(st[i] * p.scale - lq_) : 0.f;
                const float ks_ = (p.p_drop > 0.f) ? (ex_keep(p, bh, row, key) ? p.keep_scale : 0.f) : 1.f;
                Pw[(lq * 4 + i) * PLD + qb * 16 + lr] = pr * ks_;                       // P_drop^T (feeds dV)
                Sw[(lq * 4 + i) * PLD + qb * 16 + lr] = pr * (dpt[i] * ks_ - dl_);      // dS^T
            }
        }
#pragma unroll
        for (int S = 0; S < BQ / 16; ++S) {
            const f32x4 ap = *reinterpret_cast<const f32x4*>(Pw + lr * PLD + 16 * S + 4 * lq);
            const f32x4 as = *reinterpret_cast<const f32x4*>(Sw + lr * PLD + 16 * S + 4 * lq);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dva[t] = MFMA_F32(ap[j], Os[(16 * S + 4 * lq + j) * LD + 16 * t + lr], dva[t]);
                    dka[t] = MFMA_F32(as[j], Qs[(16 * S + 4 * lq + j) * LD + 16 * t + lr], dka[t]);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = k0 + w * 16 + lq * 4 + i;
        if (key < p.nk) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 16 * t + lr;
                if (c < p.d) {
                    dk[kbase + (size_t)key * p.d + c] = from_f32<T>(dka[t][i] * p.scale);
                    dv[kbase + (size_t)key * p.d + c] = from_f32<T>(dva[t][i]);
                }
            }
        }
    }
}

// ---- backward dQ: one workgroup = 16 NW query rows; loops over 32-key tiles (S and dP recomputed: deterministic)
template <typename T, int DP, int NW>
__global__ __launch_bounds__(NW * 64) void ex_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, const T* __restrict__ dout,
                                                        const float* __restrict__ lse, const float* __restrict__ delta,
                                                        T* __restrict__ dq, ExParams p) {
    constexpr int LD = DP + 4, BM = 16 * NW, BN = 32, NT = DP / 16, PLD = BN + 4, NTH = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Os = Qs + BM * LD;
    float* Ks = Os + BM * LD;
    float* Vs = Ks + BN * LD;
    float* Ss = Vs + BN * LD;
    const int ntile = (p.nq + BM - 1) / BM;
    const int bh = blockIdx.x / ntile;
    const int q0 = (blockIdx.x - bh * ntile) * BM;
    const size_t qbase = (size_t)bh * p.nq * p.d, kbase = (size_t)bh * p.nk * p.d;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;

    const bool vec = ex_quad_ok<T>(p.d, q, k, v, dout);
    ex_load_tile<T, DP, LD, NTH>(Qs, q + qbase, q0, BM, p.nq, p.d, vec);
    ex_load_tile<T, DP, LD, NTH>(Os, dout + qbase, q0, BM, p.nq, p.d, vec);
    float lrow[4], drow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = q0 + w * 16 + lq * 4 + i;
        lrow[i] = row < p.nq ? lse[(size_t)bh * p.nq + row] : 0.f;
        drow[i] = row < p.nq ? delta[(size_t)bh * p.nq + row] : 0.f;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* Sw = Ss + w * 16 * PLD;
    const int kend = p.causal ? max(0, min(p.nk, q0 + BM + p.coff)) : p.nk;
    for (int k0 = 0; k0 < kend; k0 += BN) {
        if (!ex_tile_live(p, q0, min(q0 + BM, p.nq), k0, min(k0 + BN, p.nk))) continue;
        __syncthreads();
        ex_load_tile<T, DP, LD, NTH>(Ks, k + kbase, k0, BN, p.nk, p.d, vec);
        ex_load_tile<T, DP, LD, NTH>(Vs, v + kbase, k0, BN, p.nk, p.d, vec);
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < DP / 16; ++S) {
                const f32x4 aq = *reinterpret_cast<const f32x4*>(Qs + (w * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 ao = *reinterpret_cast<const f32x4*>(Os + (w * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 bk = *reinterpret_cast<const f32x4*>(Ks + (nb * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(Vs + (nb * 16 + lr) * LD + 16 * S + 4 * lq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s = MFMA_F32(aq[j], bk[j], s);
                    dp = MFMA_F32(ao[j], bv[j], dp);
                }
            }
            const int key = k0 + nb * 16 + lr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = q0 + w * 16 + lq * 4 + i;
                const float pr = ex_visible(p, bh, row, key) ? expf(s[i] * p.scale - lrow[i]) : 0.f;
                const float ks_ = (p.p_drop > 0.f) ? (ex_keep(p, bh, row, key) ? p.keep_scale : 0.f) : 1.f;
                Sw[(lq * 4 + i) * PLD + nb * 16 + lr] = pr * (dp[i] * ks_ - drow[i]);
            }
        }
#pragma unroll
        for (int S = 0; S < BN / 16; ++S) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Sw + lr * PLD + 16 * S + 4 * lq);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = MFMA_F32(a[j], Ks[(16 * S + 4 * lq + j) * LD + 16 * t + lr], acc[t]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = q0 + w * 16 + lq * 4 + i;
        if (row < p.nq) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 16 * t + lr;
                if (c < p.d) dq[qbase + (size_t)row * p.d + c] = from_f32<T>(acc[t][i] * p.scale);
            }
        }
    }
}

// ---- host launchers
template <typename T, int DP, int NW>
static hipError_t ex_fwd_t(const ExArgs& a, hipStream_t st) {
    constexpr int LD = DP + 4;
    const size_t smem = sizeof(float) * ((16 * NW + 64) * LD + NW * 16 * 36);
    auto kern = ex_fwd_kernel<T, DP, NW>;
    hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(((a.nq + 16 * NW - 1) / (16 * NW)) * a.bh));
    ProfScope ps(K_EX_FWD, st);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, (const T*)a.q, (const T*)a.k, (const T*)a.v, (T*)a.o, a.lse,
                       make_ex_params(a));
    return hipGetLastError();
}

template <typename T, int DP, int NW>
static hipError_t ex_bwd_t(const ExArgs& a, hipStream_t st) {
    constexpr int LD = DP + 4;
    float* delta = reinterpret_cast<float*>(a.workspace);
    const long long rows = (long long)a.bh * a.nq;
    const ExParams p = make_ex_params(a);
    ProfScope ps(K_EX_BWD, st);
    hipLaunchKernelGGL(ex_delta_kernel<T>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, (const T*)a.o,
                       (const T*)a.dout, delta, rows, (int)a.d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    {
        const size_t smem = sizeof(float) * ((2 * 16 * NW + 64) * LD + 2 * NW * 16 * 36 + 64);
        auto kern = ex_dkdv_kernel<T, DP, NW>;
        e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        dim3 grid((unsigned)(((a.nk + 16 * NW - 1) / (16 * NW)) * a.bh));
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, (const T*)a.q, (const T*)a.k, (const T*)a.v, (const T*)a.dout,
                           (const float*)a.lse, (const float*)delta, (T*)a.dk, (T*)a.dv, p);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    {
        const size_t smem = sizeof(float) * ((2 * 16 * NW + 64) * LD + NW * 16 * 36);
        auto kern = ex_dq_kernel<T, DP, NW>;
        e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        dim3 grid((unsigned)(((a.nq + 16 * NW - 1) / (16 * NW)) * a.bh));
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, (const T*)a.q, (const T*)a.k, (const T*)a.v, (const T*)a.dout,
                           (const float*)a.lse, (const float*)delta, (T*)a.dq, p);
        e = hipGetLastError();
    }
    return e;
}

template <typename T>
static hipError_t ex_by_d(const ExArgs& a, bool backward, hipStream_t st) {
    if (a.d <= 64) return backward ? ex_bwd_t<T, 64, 4>(a, st) : ex_fwd_t<T, 64, 4>(a, st);
    if (a.d <= 128) return backward ? ex_bwd_t<T, 128, 4>(a, st) : ex_fwd_t<T, 128, 4>(a, st);
    return backward ? ex_bwd_t<T, 256, 2>(a, st) : ex_fwd_t<T, 256, 4>(a, st);
}

hipError_t launch_ex(const ExArgs& a, bool backward, hipStream_t st) {
    const int path = option(OPT_EX_PATH);   // 0: MFMA kernels where they apply, 1: always these, 2: MFMA or fail, 3: MFMA, never the plain kernels
    // no extras at all on a square problem: this IS the plain path — hand it to the tuned kernels (same results contract;
    // the workspace of fa_ex_backward_workspace_bytes covers their row constants)
    const bool plain = (path == 0 || path == 2) && !a.mask && !a.block_mask && a.dropout_p <= 0.0 && a.scale > 0.f;
    if (plain && a.nq == a.nk && (backward ? bwd_mfma_supported(a.dtype, a.d) : fwd_mfma_supported(a.dtype, a.d))) {
        if (!backward) return launch_fwd_mfma(FwdArgs{a.q, a.k, a.v, a.o, a.lse, a.bh, a.nq, a.d, a.dtype, a.causal, a.scale}, st);
        // (with a workspace of fa_ex_backward_workspace_bytes_fast the plain backward hands dS over as it does behind fa2_backward)
        return launch_bwd_mfma(BwdArgs{a.q, a.k, a.v, a.o, a.dout, a.lse, a.dq, a.dk, a.dv, a.bh, a.nq, a.d, a.dtype, a.causal, a.scale,
                                       a.workspace, a.workspace_bytes > ex_backward_workspace_bytes(a.bh, a.nq) ? a.workspace_bytes : ex_backward_workspace_bytes(a.bh, a.nq), 0}, st);
    }
    // the same for what the 16-bit kernels do not take (fp32 tensors, head dims that are not a multiple of 8): the plain path's
    // exact-f32 kernels (fa_generic.hip: register fragments, 16-byte operand reads — 2.5 x the rate of the kernels below)
    if ((path == 0) && !a.mask && !a.block_mask && a.dropout_p <= 0.0 && a.nq == a.nk && a.d <= 256 &&
        !(backward ? bwd_mfma_supported(a.dtype, a.d) : fwd_mfma_supported(a.dtype, a.d))) {
        if (!backward) return launch_fwd_generic(FwdArgs{a.q, a.k, a.v, a.o, a.lse, a.bh, a.nq, a.d, a.dtype, a.causal, a.scale}, st);
        return launch_bwd_generic(BwdArgs{a.q, a.k, a.v, a.o, a.dout, a.lse, a.dq, a.dk, a.dv, a.bh, a.nq, a.d, a.dtype, a.causal, a.scale,
                                          a.workspace, ex_backward_workspace_bytes(a.bh, a.nq), 0}, st);
    }
    // Nq != Nk without masks or dropout (cross attention; a cached prefix under the causal mask): the d = 128 kernels of the plain
    // path take separate row counts.  Under the causal mask only with Nk >= Nq: they assume that every query row sees key 0.
    if (plain && a.nq != a.nk && nqnk_mfma_supported(a.dtype, a.d, a.bh, a.nq, a.nk, a.causal)) {
        if (!backward) {
            FwdArgs f{a.q, a.k, a.v, a.o, a.lse, a.bh, a.nq, a.d, a.dtype, a.causal, a.scale};
            f.nk = a.nk;
            return launch_fwd_nqnk(f, st);
        }
        BwdArgs b{a.q, a.k, a.v, a.o, a.dout, a.lse, a.dq, a.dk, a.dv, a.bh, a.nq, a.d, a.dtype, a.causal, a.scale, a.workspace,
                  ex_backward_workspace_bytes(a.bh, a.nq), 0};
        b.nk = a.nk;
        float* nlse = reinterpret_cast<float*>(a.workspace);
        float* ndelta = nlse + (size_t)a.bh * a.nq;
        // with room for the dS tiles behind the row constants the dK/dV kernel hands dS to the dQ product kernel,
        // as the square backward does and by its rule — since round 3 also under the (shifted) causal diagonal
        const size_t base = (ex_backward_workspace_bytes(a.bh, a.nq) + 255) & ~(size_t)255;
        const size_t extra = bwd_ds_extra_bytes(a.bh, a.nq, a.d, a.dtype, a.causal != 0, false, a.nk);
        if (extra && a.workspace_bytes >= base + extra)
            return launch_bwd_handover(b, nlse, ndelta, reinterpret_cast<char*>(a.workspace) + base, st);
        hipError_t e = launch_bwd_dq_w4(b, nlse, ndelta, st);   // makes the row constants on its way
        if (e != hipSuccess) return e;
        return launch_bwd_dkdv_w4(b, nlse, ndelta, st);
    }
    if (path != 1 && ex_mfma_supported(a)) return launch_ex_mfma(a, backward, st);
    if (path >= 2) return hipErrorInvalidConfiguration;
    switch (a.dtype) {
        case 0: return ex_by_d<float>(a, backward, st);
        case 1: return ex_by_d<__half>(a, backward, st);
        default: return ex_by_d<__hip_bfloat16>(a, backward, st);
    }
}
// [-lse/scale | -delta] for the MFMA kernels (the exact kernels keep delta alone in the first half)
size_t ex_backward_workspace_bytes(int64_t bh, int64_t nq) { return sizeof(float) * 2 * (((size_t)bh * (size_t)nq + 63) & ~(size_t)63) + 256; }

}  // namespace fa
