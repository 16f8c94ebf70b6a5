// FlashAttention backward for gfx950, 16-bit inputs (bf16 / f16), head_dim 64 or 128.
//
// Replaces the host tile loop of csrc/fa2/fa2_bwd.cu:57-110 (and fa1/fa3 siblings):
//   delta = rowsum(dO * O);  per (K tile, Q tile): P = exp(S - lse), dV += P^T dO, dP = dO V^T,
//   dS = P * (dP - delta), dQ += dS K * scale, dK += dS^T Q * scale        (fp32 accumulation).
//
// This file: the single-kernel backward kept for the sweep (the default split backward is fa_bwd_dq_mfma.hip +
// fa_bwd_dkdv_mfma.hip, dispatched from launch_bwd_t below).  Three launches:
//   1. bwd_prep_kernel   : nlse = -lse / scale, ndelta = -rowsum(dO*O)   (row constants, fp32 workspace)
//   2. bwd_mfma_kernel   : one workgroup = 4 waves = 256 keys of one (b,h); every wave keeps dK^T and dV^T of
//                          its 64 keys in 256 accumulator registers (one wave per SIMD, 512-register budget)
//                          while the workgroup sweeps 32-row query slices.  dQ partial tiles are summed into an
//                          fp32 scratch with global float atomics (one 32x32 tile per wave per slice).
//   3. dq_convert_kernel : dq = (dtype)(scale * dq_scratch)
//
// MFMA orientation ("key on the lane"): S[q][key] and dP[q][key] are computed with the key as the accumulator
// column (lane & 31), so the P and dS accumulators are directly the B operands of dV^T += dO^T P and
// dK^T += Q^T dS (an accumulator tile as the next MFMA's operand); only dS crosses LDS, once, transposed,
// for dQ += dS K.  The row constants -lse/scale and -delta are loaded as the INITIAL accumulators of the S and
// dP chains, so P = exp2(c * S') and dS = P * dP' need no subtraction.
//
// LDS images (all 16-bit, XOR-swizzled 16-byte chunks, see TileSwz in fa_fwd_mfma.hip / below):
//   Ks  [256 keys][D]   read by rows (B operand of S) and by columns (ds_read_b64_tr_b16, B operand of dQ)
//   Qs  [2][32][D]      read by rows (A operand of S) and by columns (A operand of dK^T)
//   dOs [2][32][D]      read by rows (A operand of dP) and by columns (A operand of dV^T)
//   dSs [256 keys][32 q]  written 8 B per lane from the accumulators, read by columns (A operand of dQ)
#include "fa_common.h"
#include "fa_kernels.h"
#include <cstdlib>
#include <cstring>

namespace fa {

// dS^T image: 64-byte rows (32 queries), 8-byte slots
__device__ __forceinline__ int ds_off(int key, int slot) { return 64 * key + 8 * (slot ^ ((key >> 1) & 7)); }

// ------------------------------------------------------------------------------------------------
template <typename Tag>
__global__ __launch_bounds__(256) void bwd_prep_kernel(const uint16_t* __restrict__ o, const uint16_t* __restrict__ dout,
                                                       const float* __restrict__ lse, float* __restrict__ nlse,
                                                       float* __restrict__ ndelta, long long rows, int d,
                                                       float inv_scale) {
    // 16 lanes per row, 8 elements (16 B) per lane per step
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    float s = 0.f;
    if (row < rows) {
        for (int c = sub * 8; c < d; c += 128) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(o + row * d + c);
            const u32x4 b = *reinterpret_cast<const u32x4*>(dout + row * d + c);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s += unpack_lo<Tag>(a[j]) * unpack_lo<Tag>(b[j]) + unpack_hi<Tag>(a[j]) * unpack_hi<Tag>(b[j]);
        }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (row < rows && sub == 0) {
        ndelta[row] = -s;
        nlse[row] = -lse[row] * inv_scale;
    }
}

template <typename Tag>
__global__ __launch_bounds__(256) void dq_convert_kernel(const float* __restrict__ acc, uint16_t* __restrict__ dq,
                                                         long long n8, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(acc + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(acc + i * 8 + 4);
    u32x4 r;
    r[0] = pack2_rn<Tag>(a[0] * scale, a[1] * scale);
    r[1] = pack2_rn<Tag>(a[2] * scale, a[3] * scale);
    r[2] = pack2_rn<Tag>(b[0] * scale, b[1] * scale);
    r[3] = pack2_rn<Tag>(b[2] * scale, b[3] * scale);
    *reinterpret_cast<u32x4*>(dq + i * 8) = r;
}

// ------------------------------------------------------------------------------------------------
template <typename Tag, int D, bool CAUSAL, bool FUSED_DQ>
__global__ __launch_bounds__(256, 1) void bwd_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                          const uint16_t* __restrict__ v,
                                                          const uint16_t* __restrict__ dout,
                                                          const float* __restrict__ nlse,
                                                          const float* __restrict__ ndelta, float* __restrict__ dq_acc,
                                                          uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, int n,
                                                          int nkt, float c_log2, float scale) {
    constexpr int BK = 256, BQ = 32, NKS = D / 16, NDB = D / 32, CPR = D / 8;
    constexpr int K_BYTES = BK * D * 2, Q_BYTES = BQ * D * 2, DS_BYTES = BK * BQ * 2;
    constexpr int LPT = (BQ * CPR) / 256;          // 16-B chunks per thread per tensor per slice
    constexpr int KSPLIT = 4 / NDB;                // waves sharing one dQ tile (1 for D=128, 2 for D=64)
    constexpr int KSTEPS = (BK / 16) / KSPLIT;     // 16-key steps per wave in the dQ product
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Qs = Ks + K_BYTES;              // [2][Q_BYTES]
    char* Os = Qs + 2 * Q_BYTES;          // [2][Q_BYTES]   (dO)
    char* Ss = Os + 2 * Q_BYTES;          // dS^T
    float* Ls = reinterpret_cast<float*>(Ss + DS_BYTES);  // [2][2][32]: (-lse/scale, -delta) per buffer

    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nkt;
    const int kt = L - bh * nkt;          // key tile; under the causal mask tile 0 is the heaviest and goes first
    const int key0 = kt * BK;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t base = (size_t)bh * n * D;
    const size_t rbase = (size_t)bh * n;

    // ---- K tile -> LDS (zero rows past n)
    for (int c = tid; c < BK * CPR; c += 256) {
        const int row = c / CPR, ch = c - row * CPR;
        u32x4 t = {0u, 0u, 0u, 0u};
        if (key0 + row < n) t = *reinterpret_cast<const u32x4*>(k + base + (size_t)(key0 + row) * D + 8 * ch);
        *reinterpret_cast<u32x4*>(Ks + TileSwz<D>::off(row, ch)) = t;
    }
    // ---- V fragments (B operand of dP = dO V^T): lane holds V[key][16 ks + 8 h .. +7] for its two key blocks
    s16x8 vf[2][NKS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = key0 + 64 * w + 32 * kb + r;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 t = {0u, 0u, 0u, 0u};
            if (key < n) t = *reinterpret_cast<const u32x4*>(v + base + (size_t)key * D + 16 * ks + 8 * h);
            vf[kb][ks] = *reinterpret_cast<s16x8*>(&t);
        }
    }

    // ---- Q / dO slice staging
    int st_row[LPT], st_ch[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int c = tid + 256 * i;
        st_row[i] = c / CPR;
        st_ch[i] = c - st_row[i] * CPR;
    }
    u32x4 qreg[LPT], oreg[LPT];
    float lreg = 0.f;
    auto stage_load = [&](int qs) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = qs + st_row[i];
            qreg[i] = u32x4{0u, 0u, 0u, 0u};
            oreg[i] = u32x4{0u, 0u, 0u, 0u};
            if (row < n) {
                const size_t g = base + (size_t)row * D + 8 * st_ch[i];
                qreg[i] = *reinterpret_cast<const u32x4*>(q + g);
                oreg[i] = *reinterpret_cast<const u32x4*>(dout + g);
            }
        }
        if (tid < 64) {
            const int row = qs + (tid & 31);
            if (tid < 32) lreg = row < n ? nlse[rbase + row] : -1e30f;
            else lreg = row < n ? ndelta[rbase + row] : 0.f;
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int off = TileSwz<D>::off(st_row[i], st_ch[i]);
            *reinterpret_cast<u32x4*>(Qs + buf * Q_BYTES + off) = qreg[i];
            *reinterpret_cast<u32x4*>(Os + buf * Q_BYTES + off) = oreg[i];
        }
        if (tid < 64) Ls[buf * 64 + tid] = lreg;
    };

    f32x16 dka[2][NDB], dva[2][NDB];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < NDB; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dka[kb][t][i] = 0.f; dva[kb][t][i] = 0.f; }

    const int qs_first = CAUSAL ? (key0 / BQ) * BQ : 0;   // query slices before the tile's first key see none of it
    const int nslice = (n - qs_first + BQ - 1) / BQ;
    stage_load(qs_first);
    stage_write(0);
    __syncthreads();

    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    const int db_q = w % NDB, kpart = w / NDB;

    for (int it = 0; it < nslice; ++it) {
        const int qs = qs_first + it * BQ;
        const int cur = it & 1;
        if (it + 1 < nslice) stage_load(qs + BQ);
        const char* Qt = Qs + cur * Q_BYTES;
        const char* Ot = Os + cur * Q_BYTES;
        const float* Lt = Ls + cur * 64;

#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int kw0 = key0 + 64 * w + 32 * kb;          // first key of this wave's block
            const int kl = 64 * w + 32 * kb + r;              // this lane's key, local to the tile
            const bool active = (kw0 < n) && (!CAUSAL || kw0 <= qs + BQ - 1);
            u32x4 pp[2], sp[2];                               // packed P and dS: k-steps s = 0, 1
            if (active) {
                // ---- S' = Q K^T - lse/scale,  dP' = dO V^T - delta   (row constants as initial accumulators)
                f32x16 sacc, pacc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(Lt + 8 * g + 4 * h);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(Lt + 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sacc[4 * g + j] = a[j]; pacc[4 * g + j] = b[j]; }
                }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const s16x8 qa = *reinterpret_cast<const s16x8*>(Qt + TileSwz<D>::off(r, 2 * ks + h));
                    const s16x8 kbf = *reinterpret_cast<const s16x8*>(Ks + TileSwz<D>::off(64 * w + 32 * kb + r, 2 * ks + h));
                    mfma32_v<Tag>(qa, kbf, sacc);
                    const s16x8 oa = *reinterpret_cast<const s16x8*>(Ot + TileSwz<D>::off(r, 2 * ks + h));
                    mfma32_v<Tag>(oa, vf[kb][ks], pacc);
                    if (FUSED_DQ && D > 64 && (ks & 1)) __builtin_amdgcn_sched_barrier(0);   // bound the operand prefetch depth
                }
                mfma_result_fence(sacc, pacc);
                if (FUSED_DQ) __builtin_amdgcn_sched_barrier(0);
                // ---- P = exp2(c S'), dS = P dP'; mask on diagonal / ragged blocks
                const bool need_mask = (CAUSAL && (kw0 + 31 > qs)) || (kw0 + 32 > n);
                if (need_mask) {
                    // register i holds query qs + 4 h + rc(i); masked when that query precedes this lane's key
                    // (causal) or the key lies past n: rc(i) < thr with one per-lane threshold
                    const int key = key0 + kl;
                    const int thr = key >= n ? 64 : (CAUSAL ? key - qs - 4 * h : -1);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = ((i & 3) + 8 * (i >> 2) < thr) ? 0.f : __builtin_amdgcn_exp2f(sacc[i] * c_log2);
                        sacc[i] = p;
                        pacc[i] = p * pacc[i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = __builtin_amdgcn_exp2f(sacc[i] * c_log2);
                        sacc[i] = p;
                        pacc[i] = p * pacc[i];
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pp[s][j] = pack2<Tag>(sacc[8 * s + 2 * j], sacc[8 * s + 2 * j + 1]);
                        sp[s][j] = pack2<Tag>(pacc[8 * s + 2 * j], pacc[8 * s + 2 * j + 1]);
                    }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) { pp[s] = u32x4{0u, 0u, 0u, 0u}; sp[s] = u32x4{0u, 0u, 0u, 0u}; }
            }
            if (FUSED_DQ) __builtin_amdgcn_sched_barrier(0);
            // ---- dS^T -> LDS: lane (key kl) writes the 4 queries 8g + 4h .. +3 of register group g as 8 bytes
            if (FUSED_DQ) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 t;
                    t[0] = sp[g >> 1][2 * (g & 1)];
                    t[1] = sp[g >> 1][2 * (g & 1) + 1];
                    *reinterpret_cast<u32x2*>(Ss + ds_off(kl, 2 * g + h)) = t;
                }
            }
            if (FUSED_DQ) __builtin_amdgcn_sched_barrier(0);
            {
                // ---- dV^T += dO^T P ,  dK^T += Q^T dS   (A operands: transposed reads of the dO / Q tiles).
                // Unconditional on purpose: an inactive block (causal, before its diagonal) adds zeros.  Guarding
                // the accumulate makes hipcc merge the 256 resident accumulators through VGPR copies every slice.
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const s16x8 pb = *reinterpret_cast<s16x8*>(&pp[s]);
                    const s16x8 sb = *reinterpret_cast<s16x8*>(&sp[s]);
                    const int qa = 16 * s + 4 * h + tq;   // rows (queries) of the first 4-row block; second is +8
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const int ch = 4 * db + 2 * g16 + (tp >> 1);
                        const int o1 = TileSwz<D>::off(qa, ch) + 8 * (tp & 1);
                        const int o2 = TileSwz<D>::off(qa + 8, ch) + 8 * (tp & 1);
                        const s16x8 doT = cat8(lds_tr16(Ot + o1), lds_tr16(Ot + o2));
                        dva[kb][db] = mfma32<Tag>(doT, pb, dva[kb][db]);
                        const s16x8 qT = cat8(lds_tr16(Qt + o1), lds_tr16(Qt + o2));
                        dka[kb][db] = mfma32<Tag>(qT, sb, dka[kb][db]);
                    }
                    if (FUSED_DQ) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (FUSED_DQ) __builtin_amdgcn_sched_barrier(0);
        }
        if (FUSED_DQ) __syncthreads();  // dS^T of all 256 keys is in LDS

        // ---- dQ tile (32 queries x 32 columns db_q) = sum over this wave's share of the 256 keys of dS K
        if (FUSED_DQ) {
            f32x16 dqa;
#pragma unroll
            for (int i = 0; i < 16; ++i) dqa[i] = 0.f;
            // keys past the slice's last query are masked (dS = 0): stop there under the causal mask
            int ksteps = KSTEPS;
            if (CAUSAL) {
                const int lim = (qs + BQ - key0 + 15) / 16 - kpart * KSTEPS;   // 16-key steps that can be non-zero
                ksteps = lim < KSTEPS ? (lim < 0 ? 0 : lim) : KSTEPS;
            }
            for (int st = 0; st < ksteps; ++st) {
                const int kb16 = 16 * (kpart * KSTEPS + st);
                const int ka = kb16 + 4 * h + tq;
                const s16x8 a = cat8(lds_tr16(Ss + ds_off(ka, 4 * g16 + tp)), lds_tr16(Ss + ds_off(ka + 8, 4 * g16 + tp)));
                const int ch = 4 * db_q + 2 * g16 + (tp >> 1);
                const s16x8 b = cat8(lds_tr16(Ks + TileSwz<D>::off(ka, ch) + 8 * (tp & 1)),
                                     lds_tr16(Ks + TileSwz<D>::off(ka + 8, ch) + 8 * (tp & 1)));
                mfma32_v<Tag>(a, b, dqa);
            }
            mfma_result_fence(dqa);
            if (ksteps > 0) {
                float* dst = dq_acc + (rbase + qs) * D + 32 * db_q + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (qs + row < n) atomicAdd(dst + (size_t)row * D, dqa[i]);
                }
            }
        }
        if (it + 1 < nslice) stage_write(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: dK = scale * dK^T (transposed back on the store), dV
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = key0 + 64 * w + 32 * kb + r;
        if (key < n) {
            uint16_t* dkrow = dk + base + (size_t)key * D;
            uint16_t* dvrow = dv + base + (size_t)key * D;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 a, b;
                    a[0] = pack2_rn<Tag>(dka[kb][db][4 * g + 0] * scale, dka[kb][db][4 * g + 1] * scale);
                    a[1] = pack2_rn<Tag>(dka[kb][db][4 * g + 2] * scale, dka[kb][db][4 * g + 3] * scale);
                    b[0] = pack2_rn<Tag>(dva[kb][db][4 * g + 0], dva[kb][db][4 * g + 1]);
                    b[1] = pack2_rn<Tag>(dva[kb][db][4 * g + 2], dva[kb][db][4 * g + 3]);
                    *reinterpret_cast<u32x2*>(dkrow + 32 * db + 8 * g + 4 * h) = a;
                    *reinterpret_cast<u32x2*>(dvrow + 32 * db + 8 * g + 4 * h) = b;
                }
        }
    }
}

bool bwd_mfma_supported(int dtype, int64_t d) { return (dtype == 1 || dtype == 2) && d >= 8 && d <= 256 && d % 8 == 0; }

// workspace: [nlse (bh*n)] [ndelta (bh*n)] [pad to 256 B] [dq scratch fp32 (bh*n*d): single-kernel (atomic) variant only]
static size_t row_constants_bytes(int64_t bh, int64_t n) { return (sizeof(float) * 2 * (size_t)bh * n + 255) & ~(size_t)255; }
// The dS hand-over (d = 128, no mask, launches big enough for the stream kernels): the dK/dV kernel stores dS, the dQ kernel is
// one product over it (fa_bwd_dq_ds.hip) — 5 products instead of 7.  The (b,h) units are worked through in equal chunks whose dS
// fits the chunk size, so the workspace is bounded whatever BH and N are: 4 GiB by default (option ds_chunk_mb).  Whole step at
// 256 x 4096 x 128 against the recomputing backward (profiles/r03_ds_chunk_sweep.md, one process, interleaved): chunk 8.6 GiB (all
// units at once) -4.0 %, 4 GiB -3.3 %, 2 GiB -2.3 %, 1 GiB -2.0 %, 512 MiB +3.8 %: every chunk is a launch pair with its own
// ramp and tail.  Under the causal mask the hand-over is within 1 % of the recomputing stream kernels either way even with the
// whole launch in one 16 GiB chunk (config 3: 8.60 vs 8.66 ms; N = 4096: 4.91 vs 4.93) and loses from 8 GiB down, for a tile
// grid half of which is never written: there the recomputing backward — O(N) memory, the property the algorithm is named
// for (csrc/fa2/fa2_bwd.cu:53-57 keeps O(BH N d) scratch) — is the default and the hand-over is by option only.
// Option dq: 6 = always (causal and small launches too), 5 / 8 = never.
static size_t ds_chunk_bytes() {
    const int mb = option(OPT_DS_CHUNK_MB);
    return mb > 0 ? (size_t)mb << 20 : (size_t)4 << 30;
}
static int64_t ds_chunk_units(int64_t bh, int64_t n, int64_t nk = 0) {
    const int64_t fit = (int64_t)(ds_chunk_bytes() / ds_workspace_bytes(1, n, nk > 0 ? nk : n));
    if (fit >= bh) return bh;
    if (fit < 1) return 0;
    const int64_t nch = (bh + fit - 1) / fit;
    return (bh + nch - 1) / nch;   // equal chunks: a short last one would leave CUs idle
}
static bool bwd_ds_path(int dtype, int64_t d, int64_t bh, int64_t n, bool causal, bool atomic_variant, int64_t nk = 0) {
    const int dq_opt = option(OPT_DQ), dkdv_opt = option(OPT_DKDV);
    if (atomic_variant || !bwd_dkdv_w4_supported(dtype, d)) return false;
    // a kernel pinned by option (A/B runs of one pass against another) keeps the other pass as it was: only dq = 6 asks for this path
    if (dq_opt == 6 ? (dkdv_opt != 0 && dkdv_opt != 5) : (dq_opt != 0 || dkdv_opt != 0)) return false;
    if (option(OPT_DQ_KT) || option(OPT_DQ_TPW) || option(OPT_DQ_NLF) || (option(OPT_DQ_W4) && option(OPT_DQ_W4) != 3) || option(OPT_DKDV_TPW) || (option(OPT_DKDV_ABL) && option(OPT_DKDV_ABL) < 32) || option(OPT_DQ_ABL)) return false;
    if (ds_workspace_bytes(1, n, nk > 0 ? nk : n) > ds_chunk_bytes()) return false;   // one (b,h) alone is over the chunk size
    if (dq_opt == 6) return true;
    if (option(OPT_SMALL_GRID) == 2) return false;   // (sweeps: the small-launch kernels forced)
    // Without the mask the hand-over is ahead at every launch size (round 3, profiles/r03_bwd_variants.md: 12 - 17 % on launches of
    // 16 - 96 row tiles — a recomputing dQ workgroup walks all the keys, the product kernel's stages are short —, 5 - 9 % at the
    // headline size; round 2 had the recomputing kernels 2 - 3 % ahead up to 256 tiles: the stream kernel's start of a tile was
    // 2 500 cycles longer then).  Under the causal mask: rows of 4096 and more always (4 - 9 %), shorter rows from 160 row tiles
    // on (below, the 8-wave / 4-wave dK/dV kernels with 128-key tiles serve the launch better: 64 x 512: 0.063 against 0.079 ms),
    // and while a chunk holds 16 (b,h) units or the whole launch (N = 16384 x 32 units: 8 per chunk, -1 %).
    if (causal) return (n >= 4096 || bh * ((n + 255) / 256) >= 160) && ds_chunk_units(bh, n, nk) >= (bh < 16 ? bh : 16);
    return true;
}
size_t bwd_mfma_workspace_bytes(int64_t bh, int64_t n, int64_t d, bool atomic_variant) {
    return row_constants_bytes(bh, n) + (atomic_variant ? sizeof(float) * (size_t)bh * n * d : 0) + 256;
}
// what the dS hand-over wants on top of that (0 where it does not serve the call): a call whose workspace is smaller runs
// the recomputing dQ pass instead
size_t bwd_ds_extra_bytes(int64_t bh, int64_t n, int64_t d, int dtype, bool causal, bool atomic_variant, int64_t nk) {
    const int64_t nkk = nk > 0 ? nk : n;
    return bwd_ds_path(dtype, d, bh, n, causal, atomic_variant, nkk) ? ds_workspace_bytes(ds_chunk_units(bh, n, nkk), n, nkk) : 0;
}

// The hand-over itself: row constants, then per chunk of (b,h) units dK/dV (stores dS) and dQ = scale * dS K.  `a.nk` keys
// (0: = a.n query rows); `ds`: room for bwd_ds_extra_bytes.  Shared by the plain backward below and the extended path's Nq != Nk
// calls without masks or dropout (fa_ex.hip).
template <typename Tag>
static hipError_t run_handover_t(const BwdArgs& a, float* nlse, float* ndelta, void* ds, hipStream_t st) {
    const long long rows = (long long)a.bh * a.n;
    const int64_t nk = a.nk > 0 ? a.nk : a.n;
    {
        ProfScope ps(K_BWD_DELTA, st);
        hipLaunchKernelGGL(bwd_prep_kernel<Tag>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st,
                           (const uint16_t*)a.o, (const uint16_t*)a.dout, a.lse, nlse, ndelta, rows, (int)a.d, 1.0f / a.scale);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t step = ds_chunk_units(a.bh, a.n, nk);
    for (int64_t b0 = 0; b0 < a.bh; b0 += step) {
        BwdArgs c = a;
        c.bh = a.bh - b0 < step ? a.bh - b0 : step;
        const size_t qo = (size_t)b0 * a.n * a.d * 2, ko = (size_t)b0 * nk * a.d * 2;   // query-side / key-side tensors
        c.q = (const char*)a.q + qo; c.dout = (const char*)a.dout + qo; c.dq = (char*)a.dq + qo;
        c.k = (const char*)a.k + ko; c.v = (const char*)a.v + ko; c.dk = (char*)a.dk + ko; c.dv = (char*)a.dv + ko;
        if ((e = launch_bwd_dkdv_w4(c, nlse + b0 * a.n, ndelta + b0 * a.n, st, ds)) != hipSuccess) return e;
        if ((e = launch_bwd_dq_ds(c, ds, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_bwd_handover(const BwdArgs& a, float* nlse, float* ndelta, void* ds, hipStream_t st) {
    return a.dtype == 2 ? run_handover_t<bf16_tag>(a, nlse, ndelta, ds, st) : run_handover_t<f16_tag>(a, nlse, ndelta, ds, st);
}

template <typename Tag, int D>
static hipError_t launch_bwd_t(const BwdArgs& a, hipStream_t st) {
    constexpr int BK = 256;
    const size_t nel = (size_t)a.bh * a.n * a.d;
    const bool pad = a.d != D || D == 256;   // padded head dims and the 256-wide tiles run the split backward only
    const long long rows = (long long)a.bh * a.n;
    float* nlse = reinterpret_cast<float*>(a.workspace);
    float* ndelta = nlse + rows;
    float* dq_acc = reinterpret_cast<float*>(reinterpret_cast<char*>(a.workspace) + row_constants_bytes(a.bh, a.n));
    const bool fused = a.fused_dq != 0 && !pad;
    if (fused && a.workspace_bytes < bwd_mfma_workspace_bytes(a.bh, a.n, a.d, true)) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    if (fused) {
        e = hipMemsetAsync(dq_acc, 0, nel * sizeof(float), st);
        if (e != hipSuccess) return e;
    }
    // Split backward (default).  d > 64: the dQ kernel runs first and makes the row constants nlse = -lse / scale and
    // ndelta = -rowsum(dO * O) on its way (it holds every query row's dO in registers), the dK/dV kernel reads them.
    // d <= 64: a preparation launch makes them (fa_bwd_dq_mfma.hip, PREP).  FA_DKDV=4 selects the 4-wave dK/dV kernel.
    const int dkdv_env = option(OPT_DKDV);
    const bool split = pad || (!fused && dkdv_env != 4);
    const size_t ds_extra = pad ? 0 : bwd_ds_extra_bytes(a.bh, a.n, a.d, a.dtype, a.causal != 0, a.fused_dq != 0);
    if (ds_extra && option(OPT_DQ) == 6 && a.workspace_bytes < bwd_mfma_workspace_bytes(a.bh, a.n, a.d, false) + ds_extra)
        return hipErrorInvalidValue;   // asked for by option: fail rather than fall back
    if (ds_extra && a.workspace_bytes >= bwd_mfma_workspace_bytes(a.bh, a.n, a.d, false) + ds_extra) {
        return run_handover_t<Tag>(a, nlse, ndelta, dq_acc /* behind the row constants, 256-byte aligned */, st);
    }
    if (split && dq_makes_row_constants(a.d)) {
        e = launch_bwd_dq_mfma(a, nlse, ndelta, st);
        if (e != hipSuccess) return e;
        return launch_bwd_dkdv_mfma(a, nlse, ndelta, st);
    }
    {   // the single-kernel / 4-wave variants take their row constants from a preparation launch
        ProfScope ps(K_BWD_DELTA, st);
        hipLaunchKernelGGL(bwd_prep_kernel<Tag>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st,
                           (const uint16_t*)a.o, (const uint16_t*)a.dout, a.lse, nlse, ndelta, rows, (int)a.d, 1.0f / a.scale);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (split) {
        e = launch_bwd_dkdv_mfma(a, nlse, ndelta, st);
        if (e != hipSuccess) return e;
        return launch_bwd_dq_mfma(a, nlse, ndelta, st);
    }
    if constexpr (D != 256) {
    const int nkt = (int)((a.n + BK - 1) / BK);
    const size_t smem = (size_t)BK * D * 2 + 4 * 32 * D * 2 + BK * 32 * 2 + 2 * 64 * sizeof(float);
    const float c = a.scale * 1.4426950408889634f;
    dim3 grid((unsigned)(nkt * a.bh));
    {
        ProfScope ps(K_BWD_MFMA, st);
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e2 = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k,
                               (const uint16_t*)a.v, (const uint16_t*)a.dout, (const float*)nlse, (const float*)ndelta,
                               dq_acc, (uint16_t*)a.dk, (uint16_t*)a.dv, (int)a.n, nkt, c, a.scale);
            return hipGetLastError();
        };
        if (fused) e = a.causal ? launch(bwd_mfma_kernel<Tag, D, true, true>) : launch(bwd_mfma_kernel<Tag, D, false, true>);
        else e = a.causal ? launch(bwd_mfma_kernel<Tag, D, true, false>) : launch(bwd_mfma_kernel<Tag, D, false, false>);
    }
    if (e != hipSuccess) return e;
    if (!fused) return launch_bwd_dq_mfma(a, nlse, ndelta, st);
    {
        ProfScope ps(K_BWD_DQ_CVT, st);
        const long long n8 = (long long)(nel / 8);
        hipLaunchKernelGGL(dq_convert_kernel<Tag>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st,
                           (const float*)dq_acc, (uint16_t*)a.dq, n8, a.scale);
    }
    }
    return hipGetLastError();
}

hipError_t launch_bwd_mfma(const BwdArgs& a, hipStream_t st) {
    if (a.d > 128) return a.dtype == 2 ? launch_bwd_t<bf16_tag, 256>(a, st) : launch_bwd_t<f16_tag, 256>(a, st);
    if (a.dtype == 2) return a.d > 64 ? launch_bwd_t<bf16_tag, 128>(a, st) : launch_bwd_t<bf16_tag, 64>(a, st);
    return a.d > 64 ? launch_bwd_t<f16_tag, 128>(a, st) : launch_bwd_t<f16_tag, 64>(a, st);
}

}  // namespace fa
