// FlashAttention backward, dQ pass, one wave per SIMD (gfx950, bf16 / f16, head_dim 128).
// Launched FIRST: it also makes the row constants -lse/scale and -rowsum(dO*O) for the dK/dV pass.
//
//   dQ[q] = scale * sum_key dS[q][key] K[key],   dS = P * (dP - delta),  P = exp(S - lse),  dP = dO V^T
//   (csrc/fa2/fa2_bwd.cu:91-103 restricted to the dQ accumulation; S and dP are recomputed here so that the
//   backward needs no cross-workgroup sum: no atomics, bitwise reproducible.)
//
// Same products and orientation ("query on the lane") as fa_bwd_dq_mfma.hip, mapped like fa_bwd_dkdv_w4.hip:
//   * workgroup = 4 waves = 256 query rows; a wave owns 64 rows (two 32-row blocks) and its SIMD's whole register
//     file: dQ^T of both blocks in 128 accumulation registers, the Q and dO fragments (B operands of S^T and dP^T) in
//     the other 128 — gfx950 MFMAs take A / B operands from the accumulation registers.  Every K / V fragment read from
//     LDS feeds TWO MFMAs (one per query block): 0.5 KB of LDS reads per MFMA instead of 1 KB.
//   * per 32-key block ONE hand-ordered stream of 48 MFMAs: S^T chains, dP^T chains, dQ^T products; operands requested
//     three groups (six MFMAs) ahead from inline asm, counted lgkmcnt waits fused to the MFMA, P = exp2(.) formed under
//     the dP^T chains; dS needs both chains finished, so the block has one seam (MFMA 31 -> 32) and the next tile's
//     LDS-DMA is issued there, where the matrix pipe would wait anyway.
//   * -delta enters as the C operand of the first dP^T MFMA (a constant tuple, D != C: no accumulator initialisation),
//     -lse through the exp2 argument, the causal / ragged mask as the C operand of the first S^T MFMA (zeros except on
//     diagonal blocks), so the stream has one form and no branch.
//   * K / V arrive in 64-key tiles by LDS-DMA, four buffers, three tiles ahead, counted vmcnt waits, one barrier per tile.
#include "fa_common.h"
#include "fa_kernels.h"
#include <type_traits>
#include <utility>

namespace fa {

// One asm statement per operand group = [request of group g + 3] [counted wait for group g] [group g's first MFMA]
// (see fa_bwd_dkdv_w4.hip).  Operand classes: "v" architectural VGPR, "a" accumulation register.
//   S^T / dP^T chains: A = K / V rows (v, from the ring), B = Q / dO fragments (a), accumulator v
//   dQ^T products:     A = K^T (v), B = dS^T (v), accumulator a
// `first` forms write D = A B + C with C a separate constant tuple.
#define FA_DQ_STREAM_IMPL(TAG, OPC)                                                                                     \
    struct DqStream_##TAG {                                                                                             \
        template <int N, int OFF> static __device__ __forceinline__ void r_first(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& d, const f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%6\n\ts_waitcnt lgkmcnt(%7)\n\t" OPC " %1, %3, %4, %5"             \
                         : "=&v"(r0), "=&v"(d) : "v"(ad), "v"(a), "a"(b), "v"(c), "n"(OFF), "n"(N));                     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void r_acc(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\ts_waitcnt lgkmcnt(%6)\n\t" OPC " %1, %3, %4, %1"             \
                         : "=&v"(r0), "+v"(c) : "v"(ad), "v"(a), "a"(b), "n"(OFF), "n"(N));                             \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void r_acca(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\ts_waitcnt lgkmcnt(%6)\n\t" OPC " %1, %3, %4, %1"             \
                         : "=&v"(r0), "+a"(c) : "v"(ad), "v"(a), "v"(b), "n"(OFF), "n"(N));                             \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void t_acc(unsigned lo_a, unsigned hi_a, s16x4& lo, s16x4& hi, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b64_tr_b16 %0, %3 offset:%7\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\ts_waitcnt lgkmcnt(%8)\n\t" OPC " %2, %5, %6, %2" \
                         : "=&v"(lo), "=&v"(hi), "+v"(c) : "v"(lo_a), "v"(hi_a), "v"(a), "a"(b), "n"(OFF), "n"(N));     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void t_acca(unsigned lo_a, unsigned hi_a, s16x4& lo, s16x4& hi, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b64_tr_b16 %0, %3 offset:%7\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\ts_waitcnt lgkmcnt(%8)\n\t" OPC " %2, %5, %6, %2" \
                         : "=&v"(lo), "=&v"(hi), "+a"(c) : "v"(lo_a), "v"(hi_a), "v"(a), "v"(b), "n"(OFF), "n"(N));     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void r_first0(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& d) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\ts_waitcnt lgkmcnt(%6)\n\t" OPC " %1, %3, %4, 0"              \
                         : "=&v"(r0), "=&v"(d) : "v"(ad), "v"(a), "a"(b), "n"(OFF), "n"(N));                            \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void r_first_nw(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& d, const f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%6\n\t" OPC " %1, %3, %4, %5"             \
                         : "=&v"(r0), "=&v"(d) : "v"(ad), "v"(a), "a"(b), "v"(c), "n"(OFF));                     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void r_acc_nw(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\t" OPC " %1, %3, %4, %1"             \
                         : "=&v"(r0), "+v"(c) : "v"(ad), "v"(a), "a"(b), "n"(OFF));                             \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void r_acca_nw(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\t" OPC " %1, %3, %4, %1"             \
                         : "=&v"(r0), "+a"(c) : "v"(ad), "v"(a), "v"(b), "n"(OFF));                             \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void t_acc_nw(unsigned lo_a, unsigned hi_a, s16x4& lo, s16x4& hi, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b64_tr_b16 %0, %3 offset:%7\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t" OPC " %2, %5, %6, %2" \
                         : "=&v"(lo), "=&v"(hi), "+v"(c) : "v"(lo_a), "v"(hi_a), "v"(a), "a"(b), "n"(OFF));     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void t_acca_nw(unsigned lo_a, unsigned hi_a, s16x4& lo, s16x4& hi, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b64_tr_b16 %0, %3 offset:%7\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t" OPC " %2, %5, %6, %2" \
                         : "=&v"(lo), "=&v"(hi), "+a"(c) : "v"(lo_a), "v"(hi_a), "v"(a), "v"(b), "n"(OFF));     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void r_first0_nw(unsigned ad, s16x8& r0, s16x8 a, s16x8 b, f32x16& d) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\t" OPC " %1, %3, %4, 0"              \
                         : "=&v"(r0), "=&v"(d) : "v"(ad), "v"(a), "a"(b), "n"(OFF));                            \
        }                                                                                                               \
        static __device__ __forceinline__ void first0(s16x8 a, s16x8 b, f32x16& d) {                                    \
            asm volatile(OPC " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));                                             \
        }                                                                                                               \
        static __device__ __forceinline__ void first(s16x8 a, s16x8 b, f32x16& d, const f32x16& c) {                    \
            asm volatile(OPC " %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));                                    \
        }                                                                                                               \
        static __device__ __forceinline__ void acc(s16x8 a, s16x8 b, f32x16& c) {                                       \
            asm volatile(OPC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));                                             \
        }                                                                                                               \
        static __device__ __forceinline__ void acca(s16x8 a, s16x8 b, f32x16& c) {                                      \
            asm volatile(OPC " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));                                             \
        }                                                                                                               \
    };
FA_DQ_STREAM_IMPL(bf16, "v_mfma_f32_32x32x16_bf16")
FA_DQ_STREAM_IMPL(f16, "v_mfma_f32_32x32x16_f16")
template <typename Tag> struct DqStream;
template <> struct DqStream<bf16_tag> : DqStream_bf16 {};
template <> struct DqStream<f16_tag> : DqStream_f16 {};

__device__ __forceinline__ void dq_dma16_issue_s(rsrc_s_t rsrc, unsigned lds_dst, int voff, int soff) {
    // LDS address and soffset come from scalar arithmetic (no v_readfirstlane feeds them): one wait state for M0
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                 :: "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
}

// Placement of an iteration's vector work in its 48 MFMA gaps (slice S = the gap behind MFMA S), made by
// tools/gen_dq_schedule.py from the gap budgets (an MFMA leaves the wave about 24 cycles of issue and 5 instructions; the
// operand requests in front of the even MFMAs take their share).  FMA / EXP(h, e): one element of P^T = exp2(c S^T - lse),
// h = 0: block b, rows 32 .. 63; h = 1: block b + 1, rows 0 .. 31;  EXPL(e): the exp2 the previous iteration left over
// (its tail is the fullest part of an iteration, the head the emptiest);  SM(qb, e): one element of dS^T = P^T (dP^T - delta),
// in place;  SC(qb, m): a packed dword of dS^T;  DMA(j): an LDS-DMA piece of the tile three ahead;  KADDR / TADDR: operand
// addresses to the next tile's buffer (second block of a tile only).
namespace dqsched {
enum : unsigned char { NONE = 0, OP_FMA, OP_EXP, OP_EXPL, OP_SM, OP_SC, OP_DMA, OP_KADDR, OP_TADDR };
struct Op { unsigned char op, a, b; };
constexpr Op FMA(int h, int e) { return {OP_FMA, (unsigned char)h, (unsigned char)e}; }
constexpr Op EXP(int h, int e) { return {OP_EXP, (unsigned char)h, (unsigned char)e}; }
constexpr Op EXPL(int e) { return {OP_EXPL, 0, (unsigned char)e}; }
constexpr Op SM(int qb, int e) { return {OP_SM, (unsigned char)qb, (unsigned char)e}; }
constexpr Op SC(int qb, int m) { return {OP_SC, (unsigned char)qb, (unsigned char)m}; }
constexpr Op DMA(int j) { return {OP_DMA, (unsigned char)j, 0}; }
constexpr Op KADDR(int i) { return {OP_KADDR, (unsigned char)i, 0}; }
constexpr Op TADDR(int j) { return {OP_TADDR, (unsigned char)j, 0}; }
constexpr int kWidth = 5, kLate = 4;   // kLate: elements 16 - kLate .. 15 of P^T(b + 1), rows 0 .. 31, get their exp2 in the next iteration
// generated by tools/gen_dq_schedule.py: 0 cycles over budget in 0 gaps, 0 gaps over 5 instructions
// slice S (after MFMA S): up to 5 operations
constexpr Op kSched[48][kWidth] = {
    /*  0 (24) */ {EXPL(12), EXPL(13), EXPL(14)},
    /*  1 (20) */ {EXPL(15), KADDR(0), KADDR(1), KADDR(2)},
    /*  2 (24) */ {KADDR(3), DMA(0)},
    /*  3 (16) */ {KADDR(4), FMA(0,0), FMA(0,1)},
    /*  4 (24) */ {DMA(1), KADDR(5)},
    /*  5 (20) */ {DMA(2)},
    /*  6 (24) */ {DMA(3), KADDR(6)},
    /*  7 (16) */ {FMA(0,2), FMA(0,3), FMA(0,4)},
    /*  8 (24) */ {KADDR(7), FMA(0,5), FMA(0,6), FMA(0,7), FMA(0,8)},
    /*  9 (20) */ {FMA(0,9), FMA(0,10), FMA(0,11), FMA(0,12)},
    /* 10 (24) */ {FMA(0,13), FMA(0,14), FMA(0,15), EXP(0,0)},
    /* 11 (16) */ {EXP(0,1), EXP(0,2)},
    /* 12 (24) */ {EXP(0,3), EXP(0,4), EXP(0,5)},
    /* 13 (20) */ {EXP(0,6), EXP(0,7)},
    /* 14 (24) */ {EXP(0,8), EXP(0,9), EXP(0,10)},
    /* 15 (16) */ {EXP(0,11), EXP(0,12)},
    /* 16 (24) */ {SM(0,0), SM(0,1), EXP(0,13)},
    /* 17 (20) */ {SM(0,2), SM(0,3)},
    /* 18 (24) */ {SM(0,4), SM(0,5), SC(0,0)},
    /* 19 (16) */ {SM(0,6), SC(0,1)},
    /* 20 (24) */ {SM(0,7), SC(0,2), SM(1,0)},
    /* 21 (20) */ {SC(0,3), SM(1,1), EXP(0,14)},
    /* 22 (24) */ {SM(1,2), SM(1,3), SC(1,0)},
    /* 23 (16) */ {SM(1,4), SC(1,1)},
    /* 24 (24) */ {SM(1,5), SM(1,6), EXP(0,15)},
    /* 25 (16) */ {SM(1,7), SC(1,2)},
    /* 26 (24) */ {SC(1,3), SM(0,8), SM(0,9)},
    /* 27 (12) */ {SM(0,10)},
    /* 28 (24) */ {SM(0,11), SM(0,12), SC(0,4)},
    /* 29 (16) */ {SM(0,13), SC(0,5)},
    /* 30 (24) */ {SM(0,14), SM(0,15), SC(0,6)},
    /* 31 (12) */ {SC(0,7)},
    /* 32 (24) */ {SM(1,8), SM(1,9), FMA(1,0)},
    /* 33 (16) */ {SM(1,10), SC(1,4)},
    /* 34 (24) */ {TADDR(0), SM(1,11), FMA(1,1)},
    /* 35 (12) */ {SM(1,12)},
    /* 36 (24) */ {SM(1,13), SM(1,14), SC(1,5)},
    /* 37 (16) */ {SM(1,15), SC(1,6)},
    /* 38 (24) */ {TADDR(1), SC(1,7), TADDR(2)},
    /* 39 (12) */ {FMA(1,2), FMA(1,3)},
    /* 40 (24) */ {TADDR(3), FMA(1,4), FMA(1,5), FMA(1,6)},
    /* 41 (20) */ {FMA(1,7), FMA(1,8), FMA(1,9), FMA(1,10)},
    /* 42 (24) */ {FMA(1,11), FMA(1,12), FMA(1,13), FMA(1,14), FMA(1,15)},
    /* 43 (16) */ {EXP(1,0), EXP(1,1)},
    /* 44 (24) */ {EXP(1,2), EXP(1,3), EXP(1,4)},
    /* 45 (20) */ {EXP(1,5), EXP(1,6)},
    /* 46 (24) */ {EXP(1,7), EXP(1,8), EXP(1,9)},
    /* 47 (16) */ {EXP(1,10), EXP(1,11)},
};
}  // namespace dqsched

// ABL != 0: profiling ablations (wrong results on purpose; option dq_abl):
//   bit 0: no vector slices   bit 1: no LDS-DMA in the stream, no tile wait, no barrier   bit 2: no operand requests
//   bit 5: shader-clock stamps around the tile loop; wave 0 of workgroup 0 overwrites dq[0..7] with (cycles, tiles)
template <typename Tag, bool CAUSAL, int ABL = 0, int TPW = (CAUSAL ? 2 : 1)>
__global__ __launch_bounds__(256, 1) void bwd_dq_w4_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                           const uint16_t* __restrict__ v, const uint16_t* __restrict__ dout,
                                                           const uint16_t* __restrict__ o, const float* __restrict__ lse,
                                                           float* __restrict__ nlse, float* __restrict__ ndelta,
                                                           uint16_t* __restrict__ dq, int n, int nqt, float c_log2,
                                                           float scale, int nk /* keys; n = query rows; causal: nk >= n, diagonal at key = row + nk - n */) {
    constexpr int D = 128, NKS = 8, NDB = 4, BM = 256, BN = 64, NBUF = 4, RS = 8, AHEAD = 3;
    constexpr int KT = BN * D * 2;               // 16 KiB: the K rows of a tile (the V rows follow)
    constexpr int BUF = 2 * KT;                  // K | V
    constexpr int PIECES = 8;                    // LDS-DMA pieces (1 KiB = 4 rows) per wave and tile: 4 of K, 4 of V
    using M = DqStream<Tag>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][BUF]: tile t in buffer t % NBUF

    // TPW query tiles per workgroup: causal the heavy + light pair (as fa_bwd_dq_mfma.hip), else one
    static_assert(!CAUSAL || TPW <= 2, "causal pairing is defined for two tiles");
    const int gpb = (nqt + TPW - 1) / TPW;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / gpb;
    const int grp = L - bh * gpb;
    auto tile_of = [&](int i) { return CAUSAL ? (i == 0 ? nqt - 1 - grp : grp) : grp * TPW + i; };
    int ntile_wg = 0;
#pragma unroll
    for (int i = 0; i < TPW; ++i)
        if (CAUSAL ? (i == 0 || grp < nqt - 1 - grp) : (tile_of(i) < nqt)) ntile_wg = i + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t base = (size_t)bh * n * D, kbase = (size_t)bh * nk * D;
    const int coff = nk - n;

    const buf_rsrc_t q_rs = make_rsrc(q + base, (unsigned)n * D * 2);
    const buf_rsrc_t o_rs = make_rsrc(dout + base, (unsigned)n * D * 2);
    const buf_rsrc_t y_rs = make_rsrc(o + base, (unsigned)n * D * 2);   // the forward's output
    const rsrc_s_t k_rs = make_rsrc_s(k + kbase, (unsigned)nk * D * 2);
    const rsrc_s_t v_rs = make_rsrc_s(v + kbase, (unsigned)nk * D * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, D);
    const unsigned bbase = lds_addr_of(smem);

    int gtile = 0;   // key tiles consumed so far by this workgroup: tile t of the current query tile sits in buffer (gtile + t) & 3
    // piece J of key tile t (of the current query tile): J < 4 -> K rows 4 (w + 4 J) .., J >= 4 -> the same rows of V.
    // Tiles past the end of the tensor cost nothing: the range check answers with zeros.
    auto dma_piece = [&](auto jc, int t) {
        constexpr int J = decltype(jc)::value;
        const int pc = w + 4 * (J & 3);
        const unsigned b = bbase + ((gtile + t) & (NBUF - 1)) * BUF + (J < 4 ? 0 : KT) + pc * 1024;
        dq_dma16_issue_s(J < 4 ? k_rs : v_rs, b, dma_voff, (BN * t + 4 * pc) * 2 * D);
    };
    auto stage = [&](int t) { for_each_const([&](auto jc) { dma_piece(jc, t); }, std::make_integer_sequence<int, PIECES>{}); };
    // all but this wave's newest tile have landed.  The s_nop pads the distance from the tile's last MFMA: past the
    // barrier hipcc may move an accumulator tile (it does, at the loop exit, to line up the epilogue's asm operands)
    auto wait_tiles = [&]() { asm volatile("s_waitcnt vmcnt(8)\n\ts_nop 7" ::: "memory"); };

    s16x8 qf[2][NKS], of[2][NKS];      // B operands of S^T = K Q^T and dP^T = V dO^T: lane holds row (32 qb + r), k = 16 ks + 8 h ..
    f32x16 dqa[2][NDB];
    f32x16 sacc[2][2], pacc[2], mt[2];   // sacc[pair][qb]: blocks alternate between the two pairs
    u32x4 dsb[2][2];
    s16x8 ring[RS];
    float nl2[2], nd[2];                // this lane's two rows: -lse * log2(e) (P = exp2(c S + nl2)) and -delta
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    unsigned kaddr[NKS], tlo[NDB], thi[NDB];
    unsigned stamp_cycles = 0, stamp_tiles = 0;

    // request of operand group g of a block (8 S^T groups: K rows; 8 dP^T groups: V rows; 8 dQ^T groups: K^T)
    struct G {
        static constexpr int reads(int g) { return (g % 24) < 16 ? 1 : 2; }
        static constexpr int slot(int g) { return (g % 24) % RS; }
    };
    // The mask (diagonal blocks under the causal mask, keys past n) is the C operand of the first S^T MFMA: register i of
    // query block qb holds key k0 + 4 h + rc(i), rc(i) = (i & 3) + 8 (i >> 2); it starts at -1e30 when that key lies past
    // the lane's row (or past n), so P = 0 and dS = 0.  Rebuilt between two blocks in a wave-uniform branch; all zeros
    // otherwise.
    bool mt_dirty = false;
    auto mask_setup = [&](int q0w, int k0) {   // q0w: first row of the wave, k0: first key of the block
        const bool need_mask = (CAUSAL && (k0 + 31 > q0w + coff)) || (k0 + 32 > nk);   // wave-uniform
        if (need_mask) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int qrow = q0w + 32 * qb + r;
                const int lim = CAUSAL ? min(qrow + coff, nk - 1) : nk - 1;   // last visible key of this lane's row
                const int thr = lim - (k0 + 4 * h);
#pragma unroll
                for (int i = 0; i < 16; ++i) mt[qb][i] = ((i & 3) + 8 * (i >> 2) > thr) ? -1e30f : 0.f;
            }
            mt_dirty = true;
        } else if (mt_dirty) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mt[qb][i] = 0.f;
            mt_dirty = false;
        }
    };

    for (int it = 0; it < ntile_wg; ++it) {
    const int q0 = tile_of(it) * BM;
    const int q0w = q0 + 64 * w;
    const int kend = CAUSAL ? min(nk, q0 + BM + coff) : nk;
    const int ntiles = (kend + BN - 1) / BN;
    // tiles this wave computes: under the causal mask a tile whose first key lies past the wave's last row is skipped
    const int ntiles_w = CAUSAL ? min(ntiles, (q0w + 63 + coff) / BN + 1) : ntiles;

    // ---- prologue of a query tile: Q, dO fragments; row constants (made here and stored for the dK/dV kernel:
    // -delta = -rowsum(dO * O), csrc/fa2/fa2_bwd.cu:57, from this lane's half of the row plus lane ^ 32's; -lse / scale)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int row = q0w + 32 * qb + r;
        const bool live = row < n;
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[qb][ks] = buf_load_frag(q_rs, frag_off(row, 16 * ks + 8 * h, D, false));
            of[qb][ks] = buf_load_frag(o_rs, frag_off(row, 16 * ks + 8 * h, D, false));
            const s16x8 yf = buf_load_frag(y_rs, frag_off(row, 16 * ks + 8 * h, D, false));
            const u32x4 a = *reinterpret_cast<const u32x4*>(&of[qb][ks]), b = *reinterpret_cast<const u32x4*>(&yf);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                part += unpack_lo<Tag>(a[j]) * unpack_lo<Tag>(b[j]) + unpack_hi<Tag>(a[j]) * unpack_hi<Tag>(b[j]);
        }
        part += wave_half_swap(part);
        const float nl = live ? -lse[(size_t)bh * n + row] / scale : -1e30f;   // a padded row gets P = 0
        nd[qb] = live ? -part : 0.f;
        nl2[qb] = nl * c_log2;
        if (live && h == 0) {
            nlse[(size_t)bh * n + row] = nl;
            ndelta[(size_t)bh * n + row] = nd[qb];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) mt[qb][i] = 0.f;
#pragma unroll
        for (int t = 0; t < NDB; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) dqa[qb][t][i] = 0.f;
    }
    // (pinned here, under the loads in flight: hipcc sinks the zeroing of the resident accumulators behind the barrier otherwise)
    asm volatile("" : "+a"(dqa[0][0]), "+a"(dqa[0][1]), "+a"(dqa[0][2]), "+a"(dqa[0][3]), "+a"(dqa[1][0]), "+a"(dqa[1][1]), "+a"(dqa[1][2]), "+a"(dqa[1][3]));
    mt_dirty = false;
    // key tiles 0 .. 2 of this query tile (the tile buffers are free: every wave is past the previous tile's last barrier)
    stage(0);
    stage(1);
    stage(2);
    {
        const unsigned b0 = bbase + (gtile & (NBUF - 1)) * BUF;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kaddr[ks] = b0 + TileSwz<D>::off(r, 2 * ks + h);
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int ch = 4 * db + 2 * g16 + (tp >> 1);
            tlo[db] = b0 + TileSwz<D>::off(4 * h + tq, ch) + 8 * (tp & 1);
            thi[db] = b0 + TileSwz<D>::off(4 * h + tq + 8, ch) + 8 * (tp & 1);
        }
    }
    dma_wait_all();
    __syncthreads();
    // the fragments are first used inside the stream: make hipcc wait for them here (its vmcnt wait in the loop would
    // also drain the LDS-DMA of the tiles in flight)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < NKS; ks += 4) {
            asm volatile("" : "+a"(qf[qb][ks]), "+a"(qf[qb][ks + 1]), "+a"(qf[qb][ks + 2]), "+a"(qf[qb][ks + 3]));
            asm volatile("" : "+a"(of[qb][ks]), "+a"(of[qb][ks + 1]), "+a"(of[qb][ks + 2]), "+a"(of[qb][ks + 3]));
        }

    // ---- the stream, software-pipelined over 32-key blocks so that no chain is read right after it ends: iteration b
    // runs 24 operand groups, two MFMAs each (query block 0, then 1):
    //   g =  0 ..  7  dP^T[qb]     += V(b)[ks] dO[qb][ks]^T            request: V rows of block b
    //   g =  8 .. 15  S^T(b+1)[qb] += K(b+1)[ks] Q[qb][ks]^T           request: K rows of block b + 1
    //   g = 16 .. 23  dQ^T[qb][db] += K(b)^T[s][db] dS^T(b)[qb][s]     request: two transposed 4-row blocks of K
    // and the vector work rides in their gaps: the second half of P^T(b) = exp2(c S^T(b) - lse) under the dP^T chains,
    // dS^T(b) = P^T (dP^T - delta) under the S^T(b+1) chains (which use the other pair of S accumulators), the first half
    // of P^T(b+1) under the dQ^T products — instruction by instruction as dqsched::kSched places them.
    auto iter = [&](auto kbc, int t) {   // t: key tile (relative to the query tile), KB: 32-key block b in it
        constexpr int KB = decltype(kbc)::value, CUR = KB, NXT = 1 - KB;   // S accumulator pairs of blocks b and b + 1
        const int dlt = ((gtile + t + 1) & (NBUF - 1)) ? BUF : -(NBUF - 1) * BUF;   // to the next tile's buffer
        using std::integral_constant;
        // the work that follows MFMA S of the iteration (S = 0 .. 47), from the table above.  dP^T(b)[qb] is complete after
        // MFMA 14 + qb, S^T(b+1)[qb] after MFMA 30 + qb; dS^T(b)[.][s] feeds MFMAs 32 + 8 s ...; a consumer sits at least
        // two MFMAs behind the chain it reads (hipcc pads nothing around the asm MFMAs; tools/mfma_hazard_audit.py checks).
        auto slice = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            for_each_const([&](auto jc) {
                constexpr dqsched::Op o = dqsched::kSched[S][decltype(jc)::value];
                constexpr int a = o.a, e = o.b;
                // h = 0: P^T(b), rows 32 .. 63 -> sacc[CUR][1];  h = 1: P^T(b + 1), rows 0 .. 31 -> sacc[NXT][0]
                if constexpr (o.op == dqsched::OP_FMA && !(ABL & 1)) {
                    if constexpr (a == 0) sacc[CUR][1][e] = fmaf(sacc[CUR][1][e], c_log2, nl2[1]);
                    else sacc[NXT][0][e] = fmaf(sacc[NXT][0][e], c_log2, nl2[0]);
                } else if constexpr (o.op == dqsched::OP_EXP && !(ABL & 1)) {
                    if constexpr (a == 0) sacc[CUR][1][e] = __builtin_amdgcn_exp2f(sacc[CUR][1][e]);
                    else sacc[NXT][0][e] = __builtin_amdgcn_exp2f(sacc[NXT][0][e]);
                } else if constexpr (o.op == dqsched::OP_EXPL && !(ABL & 1)) {
                    sacc[CUR][0][e] = __builtin_amdgcn_exp2f(sacc[CUR][0][e]);
                } else if constexpr (o.op == dqsched::OP_SM && !(ABL & 1)) {
                    sacc[CUR][a][e] *= pacc[a][e] + nd[a];          // P^T(b) is dead behind its dS^T: in place
                } else if constexpr (o.op == dqsched::OP_SC && !(ABL & 1)) {
                    dsb[a][e >> 2][e & 3] = pack2<Tag>(sacc[CUR][a][2 * e], sacc[CUR][a][2 * e + 1]);
                } else if constexpr (o.op == dqsched::OP_DMA && !(ABL & 2)) {
                    dma_piece(integral_constant<int, 4 * KB + a>{}, t + 3);
                } else if constexpr (o.op == dqsched::OP_KADDR && KB == 1 && !(ABL & 1)) {
                    kaddr[a] += dlt;     // V rows of this tile: last requested in front of MFMA 2 (a - 3); K rows of the next: MFMA 10 + 2 a
                } else if constexpr (o.op == dqsched::OP_TADDR && KB == 1 && !(ABL & 1)) {
                    tlo[a] += dlt;       // last transposed request of the tile: in front of MFMA 34 + 2 a
                    thi[a] += dlt;
                }
            }, std::make_integer_sequence<int, dqsched::kWidth>{});
        };
        auto group = [&](auto gc) {
            constexpr int g = decltype(gc)::value, ph = g / 8, i = g % 8, s0 = G::slot(g);
            // one counted wait per TWO groups: an even group also waits for the next group's operands (requested a group
            // later, so the lead is two groups for those), an odd group issues no wait at all — one instruction less per
            // four MFMAs in a stream whose pace is set by the wave's instruction issue
            constexpr bool WAITS = (g % 2) == 0;
            constexpr int NWAIT = G::reads(g + 2) + G::reads(g + 3);
            constexpr int g2 = (g + AHEAD) % 24, ph2 = g2 / 8, i2 = g2 % 8, t0 = G::slot(g2);   // the group requested here
            // the 32-key block the requested group reads, as a row offset into the tile kaddr / tlo / thi point at:
            // dP^T groups of this iteration and dQ^T groups: block b; S^T groups: block b + 1; dP^T groups of the
            // NEXT iteration (g + 3 >= 24): block b + 1.  After the mid-iteration advance (KB == 1) kaddr already
            // points at the next tile, whose block 0 is "b + 1".
            constexpr bool next_blk = ph2 == 1 || g + AHEAD >= 24;
            constexpr int kbr = next_blk ? (KB == 0 ? 1 : 0) : KB;
            const s16x8 opa = ring[s0];
            auto opb = [&](auto qbc) -> s16x8 {   // read where it is used: a slice may be what makes the second one
                constexpr int qb = decltype(qbc)::value;
                if constexpr (ph == 0) return of[qb][i];
                else if constexpr (ph == 1) return qf[qb][i];
                else return *reinterpret_cast<s16x8*>(&dsb[qb][i / 4]);
            };
            const s16x8 opb0 = opb(integral_constant<int, 0>{});
            if constexpr (ABL & 4) {
                if constexpr (ph == 0 && i == 0) M::first0(opa, opb0, pacc[0]);
                else if constexpr (ph == 0) M::acc(opa, opb0, pacc[0]);
                else if constexpr (ph == 1 && i == 0) M::first(opa, opb0, sacc[NXT][0], mt[0]);
                else if constexpr (ph == 1) M::acc(opa, opb0, sacc[NXT][0]);
                else M::acca(opa, opb0, dqa[0][i % 4]);
            } else if constexpr (ph2 < 2) {
                // a row fragment: V rows (dP^T, ph2 == 0) or K rows (S^T, ph2 == 1)
                constexpr int off = kbr * 32 * 2 * D + (ph2 == 0 ? KT : 0);
                if constexpr (ph == 0 && i == 0) { if constexpr (WAITS) M::template r_first0<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, pacc[0]); else M::template r_first0_nw<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, pacc[0]); }
                else if constexpr (ph == 0) { if constexpr (WAITS) M::template r_acc<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, pacc[0]); else M::template r_acc_nw<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, pacc[0]); }
                else if constexpr (ph == 1 && i == 0) { if constexpr (WAITS) M::template r_first<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, sacc[NXT][0], mt[0]); else M::template r_first_nw<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, sacc[NXT][0], mt[0]); }
                else if constexpr (ph == 1) { if constexpr (WAITS) M::template r_acc<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, sacc[NXT][0]); else M::template r_acc_nw<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, sacc[NXT][0]); }
                else { if constexpr (WAITS) M::template r_acca<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, dqa[0][i % 4]); else M::template r_acca_nw<NWAIT, off>(kaddr[i2], ring[t0], opa, opb0, dqa[0][i % 4]); }
            } else {
                constexpr int off = KB * 32 * 2 * D + (i2 / 4) * 16 * 2 * D;
                s16x4 lo, hi;
                static_assert(ph >= 1, "transposed blocks are requested from the S^T and dQ^T groups");
                if constexpr (ph == 1) { if constexpr (WAITS) M::template t_acc<NWAIT, off>(tlo[i2 % 4], thi[i2 % 4], lo, hi, opa, opb0, sacc[NXT][0]); else M::template t_acc_nw<NWAIT, off>(tlo[i2 % 4], thi[i2 % 4], lo, hi, opa, opb0, sacc[NXT][0]); }
                else { if constexpr (WAITS) M::template t_acca<NWAIT, off>(tlo[i2 % 4], thi[i2 % 4], lo, hi, opa, opb0, dqa[0][i % 4]); else M::template t_acca_nw<NWAIT, off>(tlo[i2 % 4], thi[i2 % 4], lo, hi, opa, opb0, dqa[0][i % 4]); }
                ring[t0] = cat8(lo, hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            slice(integral_constant<int, 2 * g>{});
            __builtin_amdgcn_sched_barrier(0);
            const s16x8 opb1 = opb(integral_constant<int, 1>{});
            if constexpr (ph == 0 && i == 0) M::first0(opa, opb1, pacc[1]);
            else if constexpr (ph == 0) M::acc(opa, opb1, pacc[1]);
            else if constexpr (ph == 1 && i == 0) M::first(opa, opb1, sacc[NXT][1], mt[1]);
            else if constexpr (ph == 1) M::acc(opa, opb1, sacc[NXT][1]);
            else M::acca(opa, opb1, dqa[1][i % 4]);
            __builtin_amdgcn_sched_barrier(0);
            slice(integral_constant<int, 2 * g + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        for_each_const(group, std::make_integer_sequence<int, 24>{});
        // P^T(b + 1), rows 0 .. 31, is first read by the next iteration, behind mask_setup's branch: without a use in this
        // basic block hipcc sinks the fma / exp2 that make it out of their gaps to the head of that iteration
        asm volatile("" : "+v"(sacc[NXT][0]));
    };

    if (ntiles_w > 0) {
        // ---- lead-in: S^T(0) chains (block 0 of tile 0) and the first half of P^T(0); V rows of block 0 in flight
        using std::integral_constant;
        mask_setup(q0w, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing older may sit in the LDS queue: the counts are exact
        for_each_const([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            s16x8 kr = lds_b128_asm<0>(kaddr[ks]);
            lds_wait_for<0>(kr);
            if constexpr (ks == 0) { M::first(kr, qf[0][0], sacc[0][0], mt[0]); M::first(kr, qf[1][0], sacc[0][1], mt[1]); }
            else { M::acc(kr, qf[0][ks], sacc[0][0]); M::acc(kr, qf[1][ks], sacc[0][1]); }
        }, std::make_integer_sequence<int, NKS>{});
        // the requests of the first three dP^T groups ride here; they also put 3 + MFMAs' worth of distance behind the chains
        for_each_const([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            ring[G::slot(g)] = lds_b128_asm<KT>(kaddr[g]);
        }, std::make_integer_sequence<int, AHEAD>{});
        asm volatile("s_nop 15" : "+v"(sacc[0][0]), "+v"(sacc[0][1]));
        if (!(ABL & 1)) {   // what an iteration does for block b + 1; the last kLate exp2's belong to the next iteration
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float x = fmaf(sacc[0][0][i], c_log2, nl2[0]);
                sacc[0][0][i] = i < 16 - dqsched::kLate ? __builtin_amdgcn_exp2f(x) : x;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t_begin = 0;
    if (ABL & 32) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");
#pragma unroll 1
    for (int t = 0; t < ntiles_w; ++t) {
        mask_setup(q0w, BN * t + 32);          // of the block whose S^T chains run in the coming iteration: b + 1
        iter(std::integral_constant<int, 0>{}, t);
        mask_setup(q0w, BN * t + 64);
        iter(std::integral_constant<int, 1>{}, t);
        if (!(ABL & 2)) {
            wait_tiles();
            __builtin_amdgcn_s_barrier();
        }
    }
    if (ABL & 32) {
        unsigned long long t_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
        stamp_cycles += (unsigned)(t_end - t_begin);
        stamp_tiles += (unsigned)ntiles_w;
    }
    // causal: this wave's rows end before the workgroup's last tiles; keep feeding the other waves' tiles
    for (int t = ntiles_w; t < ntiles; ++t) {
        stage(t + 3);
        wait_tiles();
        __builtin_amdgcn_s_barrier();
    }
    gtile += ntiles;

    // ---- epilogue of the query tile: dQ = scale * dQ^T.  The accumulators were last written from asm: pad the
    // MFMA -> accumulator read hazard by hand, then everything below is compiler-visible.
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(dqa[0][0]), "+a"(dqa[0][1]), "+a"(dqa[0][2]), "+a"(dqa[0][3]), "+a"(dqa[1][0]),
                 "+a"(dqa[1][1]), "+a"(dqa[1][2]), "+a"(dqa[1][3]));
    dma_wait_all();   // the tiles past the end were requested too: nothing may still be writing LDS afterwards
    // whole-row stores through LDS (guide T21): behind the barrier every wave's DMA has landed and the tile buffers are dead;
    // a wave stages its two 32-row blocks in 16 KiB of its own
    __syncthreads();
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        u32x2 vals[NDB * 4];
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * db + g][0] = pack2_rn<Tag>(dqa[qb][db][4 * g + 0] * scale, dqa[qb][db][4 * g + 1] * scale);
                vals[4 * db + g][1] = pack2_rn<Tag>(dqa[qb][db][4 * g + 2] * scale, dqa[qb][db][4 * g + 3] * scale);
            }
        store_rows_via_lds<D>(smem + (2 * w + qb) * 32 * D * 2, vals, dq + base, q0w + 32 * qb, n, lane, D);
    }
    if (it + 1 < ntile_wg) __syncthreads();   // the tile buffers are about to be refilled
    }   // query tiles of this workgroup
    if ((ABL & 32) && L == 0 && w == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            reinterpret_cast<unsigned*>(dq)[0] = stamp_cycles;
            reinterpret_cast<unsigned*>(dq)[1] = stamp_tiles;
        }
    }
}

template <typename Tag>
static hipError_t launch_dq_w4_t(const BwdArgs& a, float* nlse, float* ndelta, hipStream_t st) {
    constexpr int D = 128, BM = 256;
    const int nqt = (int)((a.n + BM - 1) / BM);
    const size_t smem = 4 * 2 * 64 * D * 2;
    const float c = a.scale * 1.4426950408889634f;
    dim3 grid((unsigned)((a.causal ? (nqt + 1) / 2 : nqt) * a.bh));
    ProfScope ps(K_BWD_DQ_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k, (const uint16_t*)a.v,
                           (const uint16_t*)a.dout, (const uint16_t*)a.o, a.lse, nlse, ndelta, (uint16_t*)a.dq, (int)a.n, nqt, c,
                           a.scale, (int)(a.nk > 0 ? a.nk : a.n));
        return hipGetLastError();
    };
    if constexpr (std::is_same<Tag, bf16_tag>::value) {
        if (!a.causal) switch (option(OPT_DQ_ABL)) {   // profiling ablations: see the kernel's header comment
            case 1: return launch(bwd_dq_w4_kernel<Tag, false, 1>);
            case 2: return launch(bwd_dq_w4_kernel<Tag, false, 2>);
            case 4: return launch(bwd_dq_w4_kernel<Tag, false, 4>);
            case 32: return launch(bwd_dq_w4_kernel<Tag, false, 32>);
            case 33: return launch(bwd_dq_w4_kernel<Tag, false, 33>);
            case 34: return launch(bwd_dq_w4_kernel<Tag, false, 34>);
            case 36: return launch(bwd_dq_w4_kernel<Tag, false, 36>);
            case 39: return launch(bwd_dq_w4_kernel<Tag, false, 39>);
            default: break;
        }
    }
    return a.causal ? launch(bwd_dq_w4_kernel<Tag, true>) : launch(bwd_dq_w4_kernel<Tag, false>);
}

bool bwd_dq_w4_supported(int dtype, int64_t d) { return (dtype == 1 || dtype == 2) && d == 128; }

hipError_t launch_bwd_dq_w4(const BwdArgs& a, float* nlse, float* ndelta, hipStream_t st) {
    return a.dtype == 2 ? launch_dq_w4_t<bf16_tag>(a, nlse, ndelta, st) : launch_dq_w4_t<f16_tag>(a, nlse, ndelta, st);
}

}  // namespace fa
