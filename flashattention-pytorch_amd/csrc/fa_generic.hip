// Exact-f32 FlashAttention forward / backward for gfx950 (any dtype in, f32 math, head_dim <= 256).
//
// This is the path taken for float32 tensors (the reference's parity bar there is 1e-4,
// tests/utils.py:36, which rules out 16-bit MFMA operands) and for head dims the 16-bit MFMA
// kernels do not cover (the reference tests d = 32, 40, 48: tests/test_correctness_fa2.py:40,92).
// All tile GEMMs run on the f32-input matrix instruction v_mfma_f32_16x16x4_f32, whose result is
// bit-for-bit an f32 fma chain, so the numerics are those of an f32 CPU implementation.
//
// Replaces the host tile loops of csrc/fa{1,2,3}/fa?_fwd.cu:56-103 and fa?_bwd.cu:59-110
// (same math: online softmax over K tiles; backward recomputes P from lse).
//
// Layouts (16x16x4 f32 MFMA, lane l): A[row = l&15][k = l>>4], B[k = l>>4][col = l&15],
// C/D: col = l&15, row = 4*(l>>4) + reg.
#include "fa_common.h"
#include "fa_kernels.h"
#include <initializer_list>

namespace fa {

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// copy `rows` x d elements (row-major, stride d) starting at row r0 of a (n, d) matrix into an
// f32 LDS tile with leading dimension LD, zero-filling rows >= n and columns in [d, DP)
template <typename T, int DP, int LD, int NTHREADS>
__device__ __forceinline__ void load_tile_f32(float* __restrict__ dst, const T* __restrict__ src, int r0, int rows,
                                              int n, int d) {
    for (int idx = threadIdx.x; idx < rows * DP; idx += NTHREADS) {
        const int r = idx / DP, c = idx - r * DP;
        float x = 0.f;
        if (r0 + r < n && c < d) x = to_f32<T>(src[(size_t)(r0 + r) * d + c]);
        dst[r * LD + c] = x;
    }
}

// Tile staging, round 2: a thread moves 4 consecutive elements of a row at a time — one 16-byte (f32) or 8-byte (16-bit) load
// when the row length is a multiple of 4 and the tensor is aligned for it (VEC), four scalar loads otherwise — first into
// registers (issued BEFORE the tile that is being computed, so the latency hides behind its MFMAs), then into LDS.
template <typename T> struct Quad { float x[4]; };
template <typename T, bool VEC>
__device__ __forceinline__ void load_quad(float (&x)[4], const T* __restrict__ src, int row, int c, int n, int d) {
    if (VEC) {
        if (row < n && c < d) {   // d % 4 == 0: the quad is whole or absent
            if constexpr (sizeof(T) == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(src + (size_t)row * d + c);
                x[0] = t[0]; x[1] = t[1]; x[2] = t[2]; x[3] = t[3];
            } else {
                T t[4];
                *reinterpret_cast<u32x2*>(t) = *reinterpret_cast<const u32x2*>(src + (size_t)row * d + c);
                x[0] = to_f32<T>(t[0]); x[1] = to_f32<T>(t[1]); x[2] = to_f32<T>(t[2]); x[3] = to_f32<T>(t[3]);
            }
        } else {
            x[0] = x[1] = x[2] = x[3] = 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = (row < n && c + i < d) ? to_f32<T>(src[(size_t)row * d + c + i]) : 0.f;
    }
}
// rows r0 .. r0 + ROWS - 1 of a (n, d) matrix as quads in registers: thread t holds quads t, t + NTH, ... (quad q = row q / (DP/4),
// columns 4 (q % (DP/4)) ..)
template <typename T, int DP, int ROWS, int NTH, bool VEC>
struct TileRegs {
    static constexpr int NQ = (ROWS * (DP / 4) + NTH - 1) / NTH;
    float x[NQ][4];
    __device__ __forceinline__ void load(const T* __restrict__ src, int r0, int n, int d) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int qd = threadIdx.x + i * NTH, r = qd / (DP / 4), c = 4 * (qd - r * (DP / 4));
            if (ROWS * (DP / 4) % NTH == 0 || r < ROWS) load_quad<T, VEC>(x[i], src, r0 + r, c, n, d);
        }
    }
    // row-major image [ROWS][LD]
    __device__ __forceinline__ void store_rows(float* __restrict__ dst, int LD) const {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int qd = threadIdx.x + i * NTH, r = qd / (DP / 4), c = 4 * (qd - r * (DP / 4));
            if (ROWS * (DP / 4) % NTH == 0 || r < ROWS) *reinterpret_cast<f32x4*>(dst + r * LD + c) = f32x4{x[i][0], x[i][1], x[i][2], x[i][3]};
        }
    }
};

// ------------------------------------------------------------------------------------------------
// forward: one workgroup = NW waves = 16*NW query rows of one (b,h); key tiles of 32.
// Round 2: the contraction index of the 16x16x4 MFMA is permuted so that a lane's operands for FOUR consecutive MFMAs are one
// 16-byte LDS read (lane (lr, lq), step j of super-step S: k = 16 S + 4 lq + j — any permutation will do as long as A and B
// agree, and the sum stays an f32 fma chain); the wave's Q fragments stay in registers for the whole sweep; in the P V product
// the P side is read that way (V is contracted over its rows: 4-byte reads, conflict free); the next key tile is fetched into
// registers while this one is multiplied.  82 LDS reads per key tile and wave instead of 168, none of them in front of a
// global-memory wait.
// ------------------------------------------------------------------------------------------------
template <typename T, int DP, int NW, bool VEC>
__global__ __launch_bounds__(NW * 64) void fwd_f32_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, T* __restrict__ o,
                                                          float* __restrict__ lse, int n, int d, int causal,
                                                          float scale) {
    constexpr int LD = DP + 4, BM = 16 * NW, BN = 32, NT = DP / 16, PLD = BN + 4, NTH = NW * 64, NS = DP / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                 // [BN][LD]
    float* Vs = Ks + BN * LD;         // [BN][LD]
    float* Ps = Vs + BN * LD;         // [NW][16][PLD]

    const int ntile = (n + BM - 1) / BM;
    // Under the causal mask a workgroup takes a heavy and a light tile (TILE and ntile - 1 - TILE): every workgroup then
    // carries the same number of inner tiles (launches of a few rounds of workgroups ended with a tail as long as a quarter
    // of the kernel: 45 % occupancy at 64 x 2048).
    const int npair = causal ? (ntile + 1) / 2 : ntile;
    const int bh = blockIdx.x / npair;
    const int jp = blockIdx.x - bh * npair;
    for (int half = 0; half < (causal ? 2 : 1); ++half) {
    const int tq_ = (causal && half == 0) ? ntile - 1 - jp : jp;
    if (half == 1 && jp >= ntile - 1 - jp) break;   // odd tile count: the middle tile is its own pair
    if (half == 1) __syncthreads();                 // the LDS images are about to be refilled
    const int q0 = tq_ * BM;
    const size_t base = (size_t)bh * n * d;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;

    // this lane's share of its query row: elements 16 S + 4 lq .. + 3 for S = 0 .. NS - 1
    f32x4 qf[NS];
    {
        const int row = q0 + w * 16 + lr;
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            float x[4];
            load_quad<T, VEC>(x, q + base, row, 16 * S + 4 * lq, n, d);
            qf[S] = f32x4{x[0], x[1], x[2], x[3]};
        }
    }

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m[i] = -INFINITY; l[i] = 0.f; }

    const int kend = causal ? min(n, q0 + BM) : n;  // keys >= kend are masked for every row of the tile
    float* Pw = Ps + w * 16 * PLD;
    TileRegs<T, DP, BN, NTH, VEC> kr, vr;
    kr.load(k + base, 0, n, d);
    vr.load(v + base, 0, n, d);

    for (int k0 = 0; k0 < kend; k0 += BN) {
        __syncthreads();              // everybody is done with the previous tile's images
        kr.store_rows(Ks, LD);
        vr.store_rows(Vs, LD);
        __syncthreads();
        if (k0 + BN < kend) {         // the next tile: in flight during this one's products
            kr.load(k + base, k0 + BN, n, d);
            vr.load(v + base, k0 + BN, n, d);
        }

        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Ks + lr * LD + 16 * S + 4 * lq);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Ks + (16 + lr) * LD + 16 * S + 4 * lq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0 = MFMA_F32(qf[S][j], b0[j], s0);
                s1 = MFMA_F32(qf[S][j], b1[j], s1);
            }
        }
        const int key0 = k0 + lr, key1 = k0 + 16 + lr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = q0 + w * 16 + lq * 4 + i;
            float x0 = s0[i] * scale, x1 = s1[i] * scale;
            if (key0 >= n || (causal && key0 > row)) x0 = -INFINITY;
            if (key1 >= n || (causal && key1 > row)) x1 = -INFINITY;
            float mx = fmaxf(x0, x1);
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
            const float mn = fmaxf(m[i], mx);
            const float msafe = (mn == -INFINITY) ? 0.f : mn;
            const float alpha = expf(m[i] - msafe);
            const float p0 = expf(x0 - msafe), p1 = expf(x1 - msafe);
            float rs = p0 + p1;
            rs += __shfl_xor(rs, 1, 64);
            rs += __shfl_xor(rs, 2, 64);
            rs += __shfl_xor(rs, 4, 64);
            rs += __shfl_xor(rs, 8, 64);
            l[i] = l[i] * alpha + rs;
            m[i] = mn;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][i] *= alpha;
            Pw[(lq * 4 + i) * PLD + lr] = p0;
            Pw[(lq * 4 + i) * PLD + 16 + lr] = p1;
        }
        // (Pw is wave-private: no workgroup barrier, the LDS operations of a wave execute in order)
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
        if (row < n) {
            const float inv = 1.f / l[i];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 16 * t + lr;
                if (c < d) o[base + (size_t)row * d + c] = from_f32<T>(acc[t][i] * inv);
            }
            if (lr == 0) lse[(size_t)bh * n + row] = m[i] + logf(l[i]);
        }
    }
    }   // tiles of this workgroup
}

// ------------------------------------------------------------------------------------------------
// backward pre-pass: delta[bh][row] = sum_d dO*O   (csrc/fa2/fa2_bwd.cu:57)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                    float* __restrict__ delta, long long rows, int d) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    float s = 0.f;
    if (row < rows) {
        const T* po = o + row * d;
        const T* pd = dout + row * d;
        for (int c = sub; c < d; c += 16) s += to_f32<T>(po[c]) * to_f32<T>(pd[c]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (row < rows && sub == 0) delta[row] = s;
}

// ------------------------------------------------------------------------------------------------
// backward dK/dV: one workgroup = NW waves = 16*NW keys; loops over 32-row Q tiles.
// S^T[key][q] and dP^T[key][q] are computed with the key on the MFMA row so that dK/dV accumulate
// per wave without any cross-workgroup sum.   (csrc/fa2/fa2_bwd.cu:70-109)
// Round 2, as the forward: the wave's K and V rows are register fragments for the whole sweep (KREG; the 256-wide tiles keep
// them in LDS and read them 16 bytes at a time), Q / dO rows are read 16 bytes at a time (permuted contraction index), the
// next Q / dO tile is fetched into registers during the products, P^T and dS^T are wave-private in LDS (no workgroup barrier
// between writing and reading them).
// ------------------------------------------------------------------------------------------------
template <typename T, int DP, int NW, bool VEC>
__global__ __launch_bounds__(NW * 64) void bwd_dkdv_f32_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ v, const T* __restrict__ dout,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta, T* __restrict__ dk,
                                                               T* __restrict__ dv, int n, int d, int causal,
                                                               float scale) {
    constexpr int LD = DP + 4, BK = 16 * NW, BQ = 32, NT = DP / 16, PLD = BQ + 4, NTH = NW * 64, NS = DP / 16;
    constexpr bool KREG = DP <= 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                   // [BQ][LD]
    float* Os = Qs + BQ * LD;           // [BQ][LD]  (dO)
    float* Pt = Os + BQ * LD;           // [NW][16][PLD]   P^T
    float* St = Pt + NW * 16 * PLD;     // [NW][16][PLD]   dS^T
    float* Ls = St + NW * 16 * PLD;     // [BQ] lse, then [BQ] delta
    float* Ks = Ls + 2 * BQ;            // [BK][LD], [BK][LD] (V): only when !KREG

    const int ntile = (n + BK - 1) / BK;
    // Under the causal mask a workgroup takes a heavy and a light tile (TILE and ntile - 1 - TILE): every workgroup then
    // carries the same number of inner tiles (launches of a few rounds of workgroups ended with a tail as long as a quarter
    // of the kernel: 45 % occupancy at 64 x 2048).
    const int npair = causal ? (ntile + 1) / 2 : ntile;
    const int bh = blockIdx.x / npair;
    const int jp = blockIdx.x - bh * npair;
    for (int half = 0; half < (causal ? 2 : 1); ++half) {
    const int tq_ = (causal && half == 1) ? ntile - 1 - jp : jp   /* key tiles: the first ones see the most rows */;
    if (half == 1 && jp >= ntile - 1 - jp) break;   // odd tile count: the middle tile is its own pair
    if (half == 1) __syncthreads();                 // the LDS images are about to be refilled
    const int k0 = tq_ * BK;
    const size_t base = (size_t)bh * n * d;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;

    f32x4 kf[KREG ? NS : 1], vf[KREG ? NS : 1];
    if constexpr (KREG) {
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            float x[4];
            load_quad<T, VEC>(x, k + base, k0 + w * 16 + lr, 16 * S + 4 * lq, n, d);
            kf[S] = f32x4{x[0], x[1], x[2], x[3]};
            load_quad<T, VEC>(x, v + base, k0 + w * 16 + lr, 16 * S + 4 * lq, n, d);
            vf[S] = f32x4{x[0], x[1], x[2], x[3]};
        }
    } else {
        TileRegs<T, DP, BK, NTH, VEC> t;
        t.load(k + base, k0, n, d);
        t.store_rows(Ks, LD);
        t.load(v + base, k0, n, d);
        t.store_rows(Ks + BK * LD, LD);
    }
    const float* Kw = Ks + (w * 16 + lr) * LD + 4 * lq;
    const float* Vw = Kw + BK * LD;

    f32x4 dka[NT], dva[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { dka[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dva[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float* Pw = Pt + w * 16 * PLD;
    float* Sw = St + w * 16 * PLD;

    const int qstart = causal ? (k0 / BQ) * BQ : 0;  // rows < k0 see none of this tile's keys
    TileRegs<T, DP, BQ, NTH, VEC> qr, orr;
    float lreg = 0.f, dreg = 0.f;
    auto fetch = [&](int r0) {
        qr.load(q + base, r0, n, d);
        orr.load(dout + base, r0, n, d);
        if (threadIdx.x < BQ) {
            const int r = r0 + threadIdx.x;
            lreg = r < n ? lse[(size_t)bh * n + r] : 0.f;
            dreg = r < n ? delta[(size_t)bh * n + r] : 0.f;
        }
    };
    if (qstart < n) fetch(qstart);
    for (int r0 = qstart; r0 < n; r0 += BQ) {
        __syncthreads();
        qr.store_rows(Qs, LD);
        orr.store_rows(Os, LD);
        if (threadIdx.x < BQ) { Ls[threadIdx.x] = lreg; Ls[BQ + threadIdx.x] = dreg; }
        __syncthreads();
        if (r0 + BQ < n) fetch(r0 + BQ);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < NS; ++S) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(Qs + (qb * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 bo = *reinterpret_cast<const f32x4*>(Os + (qb * 16 + lr) * LD + 16 * S + 4 * lq);
                f32x4 ak, av;
                if constexpr (KREG) { ak = kf[S]; av = vf[S]; }
                else { ak = *reinterpret_cast<const f32x4*>(Kw + 16 * S); av = *reinterpret_cast<const f32x4*>(Vw + 16 * S); }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    st = MFMA_F32(ak[j], bq[j], st);
                    dpt = MFMA_F32(av[j], bo[j], dpt);
                }
            }
            const int row = r0 + qb * 16 + lr;  // query index (MFMA column)
            const float lq_ = Ls[qb * 16 + lr], dl_ = Ls[BQ + qb * 16 + lr];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = k0 + w * 16 + lq * 4 + i;
                const bool ok = row < n && key < n && !(causal && key > row);
                const float p = ok ? expf(st[i] * scale - lq_) : 0.f;
                Pw[(lq * 4 + i) * PLD + qb * 16 + lr] = p;
                Sw[(lq * 4 + i) * PLD + qb * 16 + lr] = p * (dpt[i] - dl_);
            }
        }
        // (P^T, dS^T are wave-private: the LDS operations of a wave execute in order)
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
        if (key < n) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 16 * t + lr;
                if (c < d) {
                    dk[base + (size_t)key * d + c] = from_f32<T>(dka[t][i] * scale);
                    dv[base + (size_t)key * d + c] = from_f32<T>(dva[t][i]);
                }
            }
        }
    }
    }   // tiles of this workgroup
}

// ------------------------------------------------------------------------------------------------
// backward dQ: one workgroup = NW waves = 16*NW query rows; loops over 32-key tiles (deterministic,
// no atomics: S and dP are recomputed a second time here).  Round 2: the wave's Q and dO rows are register fragments (QREG; the
// 256-wide tiles keep them in LDS), K / V rows are read 16 bytes at a time, the next K / V tile is fetched during the products.
// ------------------------------------------------------------------------------------------------
template <typename T, int DP, int NW, bool VEC>
__global__ __launch_bounds__(NW * 64) void bwd_dq_f32_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const T* __restrict__ v, const T* __restrict__ dout,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dq,
                                                             int n, int d, int causal, float scale) {
    constexpr int LD = DP + 4, BM = 16 * NW, BN = 32, NT = DP / 16, PLD = BN + 4, NTH = NW * 64, NS = DP / 16;
    constexpr bool QREG = DP <= 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                // [BN][LD]
    float* Vs = Ks + BN * LD;        // [BN][LD]
    float* Ss = Vs + BN * LD;        // [NW][16][PLD]  dS
    float* Qs = Ss + NW * 16 * PLD;  // [BM][LD], [BM][LD] (dO): only when !QREG

    const int ntile = (n + BM - 1) / BM;
    // Under the causal mask a workgroup takes a heavy and a light tile (TILE and ntile - 1 - TILE): every workgroup then
    // carries the same number of inner tiles (launches of a few rounds of workgroups ended with a tail as long as a quarter
    // of the kernel: 45 % occupancy at 64 x 2048).
    const int npair = causal ? (ntile + 1) / 2 : ntile;
    const int bh = blockIdx.x / npair;
    const int jp = blockIdx.x - bh * npair;
    for (int half = 0; half < (causal ? 2 : 1); ++half) {
    const int tq_ = (causal && half == 0) ? ntile - 1 - jp : jp;
    if (half == 1 && jp >= ntile - 1 - jp) break;   // odd tile count: the middle tile is its own pair
    if (half == 1) __syncthreads();                 // the LDS images are about to be refilled
    const int q0 = tq_ * BM;
    const size_t base = (size_t)bh * n * d;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;

    f32x4 qf[QREG ? NS : 1], of[QREG ? NS : 1];
    if constexpr (QREG) {
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            float x[4];
            load_quad<T, VEC>(x, q + base, q0 + w * 16 + lr, 16 * S + 4 * lq, n, d);
            qf[S] = f32x4{x[0], x[1], x[2], x[3]};
            load_quad<T, VEC>(x, dout + base, q0 + w * 16 + lr, 16 * S + 4 * lq, n, d);
            of[S] = f32x4{x[0], x[1], x[2], x[3]};
        }
    } else {
        TileRegs<T, DP, BM, NTH, VEC> t;
        t.load(q + base, q0, n, d);
        t.store_rows(Qs, LD);
        t.load(dout + base, q0, n, d);
        t.store_rows(Qs + BM * LD, LD);
    }
    const float* Qw = Qs + (w * 16 + lr) * LD + 4 * lq;
    const float* Ow = Qw + BM * LD;

    float lrow[4], drow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = q0 + w * 16 + lq * 4 + i;
        lrow[i] = row < n ? lse[(size_t)bh * n + row] : 0.f;
        drow[i] = row < n ? delta[(size_t)bh * n + row] : 0.f;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* Sw = Ss + w * 16 * PLD;

    const int kend = causal ? min(n, q0 + BM) : n;
    TileRegs<T, DP, BN, NTH, VEC> kr, vr;
    kr.load(k + base, 0, n, d);
    vr.load(v + base, 0, n, d);
    for (int k0 = 0; k0 < kend; k0 += BN) {
        __syncthreads();
        kr.store_rows(Ks, LD);
        vr.store_rows(Vs, LD);
        __syncthreads();
        if (k0 + BN < kend) {
            kr.load(k + base, k0 + BN, n, d);
            vr.load(v + base, k0 + BN, n, d);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < NS; ++S) {
                const f32x4 bk = *reinterpret_cast<const f32x4*>(Ks + (nb * 16 + lr) * LD + 16 * S + 4 * lq);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(Vs + (nb * 16 + lr) * LD + 16 * S + 4 * lq);
                f32x4 aq, ao;
                if constexpr (QREG) { aq = qf[S]; ao = of[S]; }
                else { aq = *reinterpret_cast<const f32x4*>(Qw + 16 * S); ao = *reinterpret_cast<const f32x4*>(Ow + 16 * S); }
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
                const bool ok = row < n && key < n && !(causal && key > row);
                const float p = ok ? expf(s[i] * scale - lrow[i]) : 0.f;
                Sw[(lq * 4 + i) * PLD + nb * 16 + lr] = p * (dp[i] - drow[i]);
            }
        }
        // (dS is wave-private: no workgroup barrier)
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
        if (row < n) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 16 * t + lr;
                if (c < d) dq[base + (size_t)row * d + c] = from_f32<T>(acc[t][i] * scale);
            }
        }
    }
    }   // tiles of this workgroup
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
template <typename K>
static hipError_t set_smem(K kern, size_t bytes) {
    return ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)bytes);
}

// VEC: rows of a multiple of 4 elements in tensors aligned for 4-element loads (16 bytes for f32, 8 for the 16-bit types)
template <typename T>
static bool quad_loads_ok(int64_t d, std::initializer_list<const void*> ps) {
    if (d % 4) return false;
    for (const void* p_ : ps)
        if (reinterpret_cast<uintptr_t>(p_) % (4 * sizeof(T))) return false;
    return true;
}

template <typename T, int DP, int NW>
static hipError_t launch_fwd_f32_t(const FwdArgs& a, hipStream_t st) {
    constexpr int LD = DP + 4;
    const size_t smem = sizeof(float) * (64 * LD + NW * 16 * 36);
    const bool vec = quad_loads_ok<T>(a.d, {a.q, a.k, a.v});
    const int64_t nt = (a.n + 16 * NW - 1) / (16 * NW);
    dim3 grid((unsigned)((a.causal ? (nt + 1) / 2 : nt) * a.bh));   // causal: a heavy and a light tile per workgroup
    ProfScope ps(K_FWD_F32, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = set_smem(kern, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, (const T*)a.q, (const T*)a.k, (const T*)a.v, (T*)a.o, a.lse,
                           (int)a.n, (int)a.d, a.causal, a.scale);
        return hipGetLastError();
    };
    return vec ? launch(fwd_f32_kernel<T, DP, NW, true>) : launch(fwd_f32_kernel<T, DP, NW, false>);
}

template <typename T, int DP, int NW>
static hipError_t launch_bwd_f32_t(const BwdArgs& a, hipStream_t st) {
    constexpr int LD = DP + 4;
    constexpr bool REG = DP <= 128;   // the wave's own rows as register fragments (else a second pair of LDS tiles)
    float* delta = reinterpret_cast<float*>(a.workspace);
    const long long rows = (long long)a.bh * a.n;
    const bool vec = quad_loads_ok<T>(a.d, {a.q, a.k, a.v, a.dout});
    {
        ProfScope ps(K_BWD_DELTA, st);
        hipLaunchKernelGGL(delta_kernel<T>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, (const T*)a.o,
                           (const T*)a.dout, delta, rows, (int)a.d);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t nt = (a.n + 16 * NW - 1) / (16 * NW);
    dim3 grid((unsigned)((a.causal ? (nt + 1) / 2 : nt) * a.bh));   // causal: a heavy and a light tile per workgroup
    {
        const size_t smem = sizeof(float) * ((64 + (REG ? 0 : 2 * 16 * NW)) * LD + 2 * NW * 16 * 36 + 64);
        ProfScope ps(K_BWD_DKDV_F32, st);
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e2 = set_smem(kern, smem);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, (const T*)a.q, (const T*)a.k, (const T*)a.v,
                               (const T*)a.dout, a.lse, (const float*)delta, (T*)a.dk, (T*)a.dv, (int)a.n, (int)a.d,
                               a.causal, a.scale);
            return hipGetLastError();
        };
        e = vec ? launch(bwd_dkdv_f32_kernel<T, DP, NW, true>) : launch(bwd_dkdv_f32_kernel<T, DP, NW, false>);
        if (e != hipSuccess) return e;
    }
    {
        const size_t smem = sizeof(float) * ((64 + (REG ? 0 : 2 * 16 * NW)) * LD + NW * 16 * 36);
        ProfScope ps(K_BWD_DQ_F32, st);
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e2 = set_smem(kern, smem);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, (const T*)a.q, (const T*)a.k, (const T*)a.v,
                               (const T*)a.dout, a.lse, (const float*)delta, (T*)a.dq, (int)a.n, (int)a.d, a.causal,
                               a.scale);
            return hipGetLastError();
        };
        e = vec ? launch(bwd_dq_f32_kernel<T, DP, NW, true>) : launch(bwd_dq_f32_kernel<T, DP, NW, false>);
    }
    return e;
}

template <typename T>
static hipError_t fwd_by_d(const FwdArgs& a, hipStream_t st) {
    if (a.d <= 64) return launch_fwd_f32_t<T, 64, 4>(a, st);
    if (a.d <= 128) return launch_fwd_f32_t<T, 128, 4>(a, st);
    return launch_fwd_f32_t<T, 256, 4>(a, st);
}
template <typename T>
static hipError_t bwd_by_d(const BwdArgs& a, hipStream_t st) {
    if (a.d <= 64) return launch_bwd_f32_t<T, 64, 4>(a, st);
    if (a.d <= 128) return launch_bwd_f32_t<T, 128, 4>(a, st);
    return launch_bwd_f32_t<T, 256, 2>(a, st);
}

hipError_t launch_fwd_generic(const FwdArgs& a, hipStream_t st) {
    switch (a.dtype) {
        case 0: return fwd_by_d<float>(a, st);
        case 1: return fwd_by_d<__half>(a, st);
        default: return fwd_by_d<__hip_bfloat16>(a, st);
    }
}
hipError_t launch_bwd_generic(const BwdArgs& a, hipStream_t st) {
    switch (a.dtype) {
        case 0: return bwd_by_d<float>(a, st);
        case 1: return bwd_by_d<__half>(a, st);
        default: return bwd_by_d<__hip_bfloat16>(a, st);
    }
}
size_t bwd_generic_workspace_bytes(int64_t bh, int64_t n) { return sizeof(float) * (size_t)bh * (size_t)n; }

}  // namespace fa
