// FlashAttention backward, dK/dV pass, one wave per SIMD (gfx950, bf16 / f16, head_dim 128).
//
//   dV[key] = sum_q P[q][key] dO[q],   dK[key] = scale * sum_q dS[q][key] Q[q]
//   P = exp(S - lse), dS = P * (dO V^T - delta)            (csrc/fa2/fa2_bwd.cu:91-104, the dK/dV half)
//
// Same products, same orientation ("key on the lane") and the same rounding points as fa_bwd_dkdv_mfma.hip — the
// results are bitwise the same — but a different machine mapping:
//   * workgroup = 4 waves = 256 keys; a wave owns 64 keys (two 32-key blocks) and the whole 512-register file of
//     its SIMD: dK^T and dV^T of both blocks live in the 256 accumulation registers, V rows in 64 VGPRs.  Every Q / dO
//     operand fragment read from LDS now feeds TWO MFMAs (one per key block): 0.75 KB of LDS reads per MFMA
//     instead of 1.25 KB (the 8-wave kernel's limiter is the LDS operand traffic, profiles/r01_tile_sweep.md).
//   * no partner wave hides latency, so the 64 MFMAs of a (32-query x 64-key) block are ONE hand-ordered stream:
//     LDS operands requested two operand groups (four MFMAs) ahead from inline asm with counted lgkmcnt waits
//     fused to the MFMA, and the vector work (exp2, dS, packs, address updates, the next tile's LDS-DMA) cut into
//     slices that sit in the MFMA gaps: P of a block is formed under its dP chains, dS under its dV products.
//   * Q / dO / row constants arrive in 32-row tiles by LDS-DMA, four buffers, issued three tiles ahead with counted
//     vmcnt waits: a tile has two block times to land and the first operands of block t+1 are requested before
//     the barrier that ends block t.
#include "fa_common.h"
#include "fa_kernels.h"
#include <type_traits>
#include <utility>

namespace fa {

// MFMA from inline asm with the accumulator pinned to the architectural VGPRs ("+v"): S' and dP' are consumed by VALU
// code, and in a 512-register kernel hipcc gives every builtin MFMA an AGPR accumulator (v_accvgpr copies around each
// use).  hipcc pads nothing around asm: every consumer of these accumulators sits at least two MFMAs downstream, and
// tools/mfma_hazard_audit.py checks the distances in the built code object (hipcc is free to move a tile).
#define FA_W4_MFMA_IMPL(TAG, OPC)                                                                                       \
    struct W4Mfma_##TAG {                                                                                               \
        template <int N> static __device__ __forceinline__ void v_wait(s16x8 a, s16x8 b, f32x16& c) {                   \
            asm volatile("s_waitcnt lgkmcnt(%3)\n\t" OPC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b), "n"(N));           \
        }                                                                                                               \
        static __device__ __forceinline__ void v(s16x8 a, s16x8 b, f32x16& c) {                                         \
            asm volatile(OPC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));                                             \
        }                                                                                                               \
    };
FA_W4_MFMA_IMPL(bf16, "v_mfma_f32_32x32x16_bf16")
FA_W4_MFMA_IMPL(f16, "v_mfma_f32_32x32x16_f16")
template <typename Tag> struct W4Mfma;
template <> struct W4Mfma<bf16_tag> : W4Mfma_bf16 {};
template <> struct W4Mfma<f16_tag> : W4Mfma_f16 {};

// Row constants as initial accumulators: four broadcast reads of 4 floats (registers 4g .. 4g+3 <- floats 8g .. 8g+3
// past addr).  Compiler-visible loads: hipcc waits for them itself before the chain's first MFMA.
template <int OFF> __device__ __forceinline__ void lds_acc_init(unsigned addr, f32x16& acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 a = *(const f32x4 __attribute__((address_space(3)))*)(uintptr_t)(addr + OFF + 32 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = a[j];
    }
}

template <typename Tag, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void bwd_dkdv_w4_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                             const uint16_t* __restrict__ v,
                                                             const uint16_t* __restrict__ dout,
                                                             const float* __restrict__ nlse,
                                                             const float* __restrict__ ndelta, uint16_t* __restrict__ dk,
                                                             uint16_t* __restrict__ dv, int n, int nkt, float c_log2,
                                                             float scale) {
    constexpr int D = 128, NKS = 8, NDB = 4, BK = 256, BQ = 32, NBUF = 4, RS = 12;
    constexpr int K_BYTES = BK * D * 2;          // 64 KiB: the workgroup's K rows (B operand of S)
    constexpr int QT = BQ * D * 2;               // 8 KiB: one 32-row tile of Q (or dO)
    constexpr int BUF = 2 * QT + 512;            // Q | dO | 64 x -lse/scale | 64 x -delta   (the constants use 32 of the 64)
    using M = W4Mfma<Tag>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Bs = smem + K_BYTES;                   // [NBUF][BUF]: tile t in buffer t % NBUF

    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nkt;
    const int kt = L - bh * nkt;                 // key tile; under the causal mask tile 0 is the heaviest and goes first
    const int key0 = kt * BK;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t base = (size_t)bh * n * D;
    const size_t rbase = (size_t)bh * n;
    const int kw0 = key0 + 64 * w;               // first key of this wave

    const rsrc_s_t k_rs = make_rsrc_s(k + base, (unsigned)n * D * 2);
    const rsrc_s_t q_rs = make_rsrc_s(q + base, (unsigned)n * D * 2);
    const rsrc_s_t o_rs = make_rsrc_s(dout + base, (unsigned)n * D * 2);
    const rsrc_s_t l_rs = make_rsrc_s(nlse + rbase, (unsigned)n * 4);
    const rsrc_s_t d_rs = make_rsrc_s(ndelta + rbase, (unsigned)n * 4);
    const buf_rsrc_t v_rs = make_rsrc(v + base, (unsigned)n * D * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, D);

    const int qs_first = CAUSAL ? key0 : 0;      // earlier queries see none of this workgroup's keys
    const int nblk = (n - qs_first + BQ - 1) / BQ;
    const unsigned bbase = lds_addr_of(Bs);
    // one LDS-DMA piece (1 KiB = 4 rows) of tile t: J = 0, 1 -> Q, J = 2, 3 -> dO, J = 4 -> the wave's row constants
    auto dma_piece = [&](auto jc, int t) {
        constexpr int J = decltype(jc)::value;
        const int qs = qs_first + BQ * t;
        const unsigned b = bbase + (t & (NBUF - 1)) * BUF;
        if constexpr (J < 4) {
            const int pc = w + 4 * (J & 1);
            dma16_issue(J < 2 ? q_rs : o_rs, b + (J < 2 ? 0 : QT) + pc * 1024, dma_voff,
                        __builtin_amdgcn_readfirstlane((qs + 4 * pc) * 2 * D));
        } else {
            // 64 floats each (rows qs .. qs+63; the second half is never read); rows >= n read as 0: harmless, their dO is 0
            if (w == 0) dma4_issue(l_rs, b + 2 * QT, lane * 4, __builtin_amdgcn_readfirstlane(qs * 4));
            if (w == 1) dma4_issue(d_rs, b + 2 * QT + 256, lane * 4, __builtin_amdgcn_readfirstlane(qs * 4));
        }
    };
    auto stage = [&](int t) { for_each_const([&](auto jc) { dma_piece(jc, t); }, std::make_integer_sequence<int, 5>{}); };
    // all but this wave's newest tile (4 pieces, + 1 row-constant piece on waves 0 and 1) have landed
    auto wait_tiles = [&]() {
        if (w < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };

    // ---- prologue: K tile, V fragments (B operand of dP = dO V^T), tiles 0 .. 2
    dma_stage_tile<D, BK, 4>(k_rs, Ks, key0, dma_voff, w);
    s16x8 vf[2][NKS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) vf[kb][ks] = buf_load_frag(v_rs, frag_off(kw0 + 32 * kb + r, 16 * ks + 8 * h, D, false));
    stage(0);
    stage(1);
    stage(2);

    f32x16 dka[2][NDB], dva[2][NDB];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < NDB; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dka[kb][t][i] = 0.f; dva[kb][t][i] = 0.f; }

    // lane-constant operand addresses (LDS bytes).  Q / dO rows and K rows: row r, chunk 2 ks + h; transposed reads:
    // 4-row blocks at rows 4 h + tq (+8), chunks 4 db + 2 g16 + (tp >> 1); everything else is an immediate offset
    // (the swizzle depends on the row modulo 16 only).  The tile addresses move from buffer to buffer during the stream.
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    unsigned kaddr[NKS], qaddr[NKS], tlo[NDB], thi[NDB], laddr;
    // first block this wave computes: earlier ones hold only queries before its first key (causal)
    const int fb = CAUSAL ? 2 * w : 0;
    {
        const unsigned kbase = lds_addr_of(Ks) + 64 * w * 2 * D, b0 = bbase + (fb & (NBUF - 1)) * BUF;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int off = TileSwz<D>::off(r, 2 * ks + h);
            kaddr[ks] = kbase + off;
            qaddr[ks] = b0 + off;
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int ch = 4 * db + 2 * g16 + (tp >> 1);
            tlo[db] = b0 + TileSwz<D>::off(4 * h + tq, ch) + 8 * (tp & 1);
            thi[db] = b0 + TileSwz<D>::off(4 * h + tq + 8, ch) + 8 * (tp & 1);
        }
        laddr = b0 + 2 * QT + 16 * h;
    }

    dma_wait_all();
    __syncthreads();
    // the V fragments are first used inside the stream: make hipcc wait for them here, not in the loop (its vmcnt wait
    // there would also drain the LDS-DMA of the tiles in flight)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ks += 4)
            asm volatile("" : "+v"(vf[kb][ks]), "+v"(vf[kb][ks + 1]), "+v"(vf[kb][ks + 2]), "+v"(vf[kb][ks + 3]));

    // ---- the stream.  Operand groups g = 0 .. 31 of a block, two MFMAs each (key block 0, then 1):
    //   g =  0 ..  7  S'[kb]  += Q[ks] K[kb][ks]          reads: Q rows, K rows of block 0, K rows of block 1
    //   g =  8 .. 15  dP'[kb] += dO[ks] V[kb][ks]         reads: dO rows                    (V in registers)
    //   g = 16 .. 23  dV^T[kb][db] += dO^T[s][db] P[kb][s]    reads: two transposed 4-row blocks of dO
    //   g = 24 .. 31  dK^T[kb][db] += Q^T[s][db] dS[kb][s]    reads: two transposed 4-row blocks of Q
    // Group g + 2 is requested right before group g's MFMAs; groups 32, 33 are groups 0, 1 of the next block.
    s16x8 ring[RS];
    f32x16 sacc[2], pacc[2];
    u32x4 pp[2][2], sp[2][2];
    struct G {
        static constexpr int reads(int g) { return (g % 32) < 8 ? 3 : ((g % 32) < 16 ? 1 : 2); }
        static constexpr int opidx(int g) { return (g % 32) < 8 ? 3 * (g % 32) : ((g % 32) < 16 ? 16 + (g % 32) : 16 + (g % 32)); }
        static constexpr int slot(int g) { return opidx(g) % RS; }
    };
    static_assert(G::opidx(8) == 24 && G::opidx(16) == 32 && G::opidx(31) == 47 && 48 % RS == 0, "ring bookkeeping");
    auto fetch = [&](auto gc) {
        constexpr int g = decltype(gc)::value % 32, ph = g / 8, i = g % 8, s0 = G::slot(g);
        if constexpr (ph == 0) {
            ring[s0] = lds_b128_asm<0>(qaddr[i]);
            ring[s0 + 1] = lds_b128_asm<0>(kaddr[i]);
            ring[s0 + 2] = lds_b128_asm<32 * 2 * D>(kaddr[i]);
        } else if constexpr (ph == 1) {
            ring[s0] = lds_b128_asm<QT>(qaddr[i]);
        } else {
            constexpr int off = (ph == 2 ? QT : 0) + (i / 4) * 16 * 2 * D;
            ring[s0] = cat8(lds_tr16_asm<off>(tlo[i % 4]), lds_tr16_asm<off>(thi[i % 4]));
        }
    };

    // Mask (diagonal blocks under the causal mask, keys past n): applied to the INITIAL accumulator of S' — a masked
    // element starts at -1e30, so P = exp2(c S') = 0 and dS = 0 — right after the row constants are loaded, in a short
    // wave-uniform branch.  The stream itself has one form: two copies of it under a branch cost hipcc's register
    // allocator 1.1 KB of spills.  Register i holds query qs + 4 h + rc(i), rc(i) = (i & 3) + 8 (i >> 2); it is masked
    // when it precedes this lane's key (causal) or the key lies past n: rc(i) < thr, one per-lane threshold per key block.
    auto mask_init = [&](int blk) {   // blk: the block whose initial accumulators were just requested
        const int qs = qs_first + BQ * blk;
        const bool need_mask = (CAUSAL && (kw0 + 63 > qs)) || (kw0 + 64 > n);   // wave-uniform
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int key = kw0 + 32 * kb + r;
                const int thr = key >= n ? 64 : (CAUSAL ? key - qs - 4 * h : -1);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) < thr) sacc[kb][i] = -1e30f;
            }
        }
    };
    auto block = [&](int blk) {
        const int dlt = ((blk + 1) & (NBUF - 1)) ? BUF : -(NBUF - 1) * BUF;   // to the next tile's buffer
        auto PU = [&](auto kbc, auto mc) {   // P pair: elements 2m, 2m+1 of S'[kb] -> exp2 -> one packed dword
            constexpr int kb = decltype(kbc)::value, m = decltype(mc)::value;
            const float p0 = __builtin_amdgcn_exp2f(sacc[kb][2 * m] * c_log2), p1 = __builtin_amdgcn_exp2f(sacc[kb][2 * m + 1] * c_log2);
            pp[kb][m >> 2][m & 3] = pack2<Tag>(p0, p1);
        };
        auto SU = [&](auto kbc, auto mc) {   // dS pair = P dP' with the 16-bit P that also feeds dV
            constexpr int kb = decltype(kbc)::value, m = decltype(mc)::value;
            sp[kb][m >> 2][m & 3] = mul_pack<Tag>(pp[kb][m >> 2][m & 3], pacc[kb][2 * m], pacc[kb][2 * m + 1]);
        };
        using std::integral_constant;
        auto slice = [&](auto sc) {          // the vector work that follows MFMA S of the block (S = 0 .. 63)
            constexpr int S = decltype(sc)::value;
            if constexpr (S >= 17 && S <= 24) PU(integral_constant<int, 0>{}, integral_constant<int, S - 17>{});
            else if constexpr (S >= 25 && S <= 32) PU(integral_constant<int, 1>{}, integral_constant<int, S - 25>{});
            else if constexpr (S >= 33 && S <= 36) {
                SU(integral_constant<int, 0>{}, integral_constant<int, 2 * (S - 33)>{});
                SU(integral_constant<int, 0>{}, integral_constant<int, 2 * (S - 33) + 1>{});
            } else if constexpr (S >= 37 && S <= 40) {
                SU(integral_constant<int, 1>{}, integral_constant<int, 2 * (S - 37)>{});
                SU(integral_constant<int, 1>{}, integral_constant<int, 2 * (S - 37) + 1>{});
            } else if constexpr (S == 41) laddr += dlt;
            // the next block's row constants become the initial accumulators (S' and dP' are free by now)
            else if constexpr (S == 42) lds_acc_init<0>(laddr, sacc[0]);
            else if constexpr (S == 43) { lds_acc_init<0>(laddr, sacc[1]); mask_init(blk + 1); }
            else if constexpr (S == 44) lds_acc_init<256>(laddr, pacc[0]);
            else if constexpr (S == 45) lds_acc_init<256>(laddr, pacc[1]);
            else if constexpr (S >= 46 && S <= 50) {
                dma_piece(integral_constant<int, S - 46>{}, blk + 3);
                // hipcc waits for its row-constant loads at their first use: give it one here, where few operand requests
                // are in flight, instead of the head of the next block's chains (its wait drains our requests too)
                if constexpr (S == 50) asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(pacc[0]), "+v"(pacc[1]));
            }
            else if constexpr (S >= 51 && S <= 58) qaddr[S - 51] += dlt;   // dO rows were last requested at MFMA 26
            else if constexpr (S >= 60) { tlo[S - 60] += dlt; thi[S - 60] += dlt; }   // last transposed request: MFMA 58
        };
        auto group = [&](auto gc) {
            constexpr int g = decltype(gc)::value, ph = g / 8, i = g % 8, s0 = G::slot(g);
            constexpr int NWAIT = G::reads(g + 1) + G::reads(g + 2);
            fetch(integral_constant<int, g + 2>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ph == 0) M::template v_wait<NWAIT>(ring[s0], ring[s0 + 1], sacc[0]);
            else if constexpr (ph == 1) M::template v_wait<NWAIT>(ring[s0], vf[0][i], pacc[0]);
            else {
                // resident accumulators: compiler-visible MFMAs (hipcc keeps them in the accumulation registers and pads
                // their hazards); the operand wait is ours, the requests came from asm
                lds_wait_for<NWAIT>(ring[s0]);
                if constexpr (ph == 2) dva[0][i % 4] = mfma32<Tag>(ring[s0], *reinterpret_cast<s16x8*>(&pp[0][i / 4]), dva[0][i % 4]);
                else dka[0][i % 4] = mfma32<Tag>(ring[s0], *reinterpret_cast<s16x8*>(&sp[0][i / 4]), dka[0][i % 4]);
            }
            __builtin_amdgcn_sched_barrier(0);
            slice(integral_constant<int, 2 * g>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ph == 0) M::v(ring[s0], ring[s0 + 2], sacc[1]);
            else if constexpr (ph == 1) M::v(ring[s0], vf[1][i], pacc[1]);
            else if constexpr (ph == 2) dva[1][i % 4] = mfma32<Tag>(ring[s0], *reinterpret_cast<s16x8*>(&pp[1][i / 4]), dva[1][i % 4]);
            else dka[1][i % 4] = mfma32<Tag>(ring[s0], *reinterpret_cast<s16x8*>(&sp[1][i / 4]), dka[1][i % 4]);
            __builtin_amdgcn_sched_barrier(0);
            slice(integral_constant<int, 2 * g + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        for_each_const(group, std::make_integer_sequence<int, 32>{});
    };

    // feed-only blocks (causal: queries before this wave's first key): the wave's share of the LDS-DMA and the barriers
    for (int blk = 0; blk < min(fb, nblk); ++blk) {
        stage(blk + 3);
        wait_tiles();
        __builtin_amdgcn_s_barrier();
    }
    if (fb < nblk) {
        // first block: row constants into the accumulators, operand groups 0 and 1 in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing older may sit in the LDS queue: the counts are exact
        lds_acc_init<0>(laddr, sacc[0]);
        lds_acc_init<0>(laddr, sacc[1]);
        mask_init(fb);
        lds_acc_init<256>(laddr, pacc[0]);
        lds_acc_init<256>(laddr, pacc[1]);
        fetch(std::integral_constant<int, 0>{});
        fetch(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int blk = fb; blk < nblk; ++blk) {
            block(blk);
            wait_tiles();
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue: dK = scale * dK^T (transposed back on the store), dV
    dma_wait_all();   // nothing of this workgroup may still be writing LDS when the next one takes the CU
    // a wave only ever read its own 64 K rows: that slice of the K tile is its staging area for whole-row stores
    char* stg = Ks + w * 64 * D * 2;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        u32x2 vals[NDB * 4];
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * db + g][0] = pack2_rn<Tag>(dka[kb][db][4 * g + 0] * scale, dka[kb][db][4 * g + 1] * scale);
                vals[4 * db + g][1] = pack2_rn<Tag>(dka[kb][db][4 * g + 2] * scale, dka[kb][db][4 * g + 3] * scale);
            }
        store_rows_via_lds<D>(stg, vals, dk + base, kw0 + 32 * kb, n, lane, D);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * db + g][0] = pack2_rn<Tag>(dva[kb][db][4 * g + 0], dva[kb][db][4 * g + 1]);
                vals[4 * db + g][1] = pack2_rn<Tag>(dva[kb][db][4 * g + 2], dva[kb][db][4 * g + 3]);
            }
        store_rows_via_lds<D>(stg + 32 * D * 2, vals, dv + base, kw0 + 32 * kb, n, lane, D);
    }
}

template <typename Tag>
static hipError_t launch_dkdv_w4_t(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st) {
    constexpr int D = 128, BK = 256;
    const int nkt = (int)((a.n + BK - 1) / BK);
    const size_t smem = (size_t)BK * D * 2 + 4 * (2 * 32 * D * 2 + 512);
    const float c = a.scale * 1.4426950408889634f;
    dim3 grid((unsigned)(nkt * a.bh));
    ProfScope ps(K_BWD_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k, (const uint16_t*)a.v,
                           (const uint16_t*)a.dout, nlse, ndelta, (uint16_t*)a.dk, (uint16_t*)a.dv, (int)a.n, nkt, c, a.scale);
        return hipGetLastError();
    };
    return a.causal ? launch(bwd_dkdv_w4_kernel<Tag, true>) : launch(bwd_dkdv_w4_kernel<Tag, false>);
}

bool bwd_dkdv_w4_supported(int dtype, int64_t d) { return (dtype == 1 || dtype == 2) && d == 128; }

hipError_t launch_bwd_dkdv_w4(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st) {
    return a.dtype == 2 ? launch_dkdv_w4_t<bf16_tag>(a, nlse, ndelta, st) : launch_dkdv_w4_t<f16_tag>(a, nlse, ndelta, st);
}

}  // namespace fa
