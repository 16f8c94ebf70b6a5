// FlashAttention backward, dK/dV pass, one wave per SIMD (gfx950, bf16 / f16, head_dim 128).
//
//   dV[key] = sum_q P[q][key] dO[q],   dK[key] = scale * sum_q dS[q][key] Q[q]
//   P = exp(S - lse), dS = P * (dO V^T - delta)            (csrc/fa2/fa2_bwd.cu:91-104, the dK/dV half)
//
// Same products and the same orientation ("key on the lane") as fa_bwd_dkdv_mfma.hip; dV is bitwise the same, dK differs
// in the last bit (dS = P dP' is formed from the f32 P here, from the 16-bit P there).  A different machine mapping:
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

// The stream's instructions come from inline asm: ONE statement per operand group = [requests of group g + 2]
// [counted wait for group g's operands] [group g's first MFMA], so hipcc neither re-orders them nor pads the seams
// between them (it puts an s_nop after every asm statement whose outputs the next one reads).  Accumulators: "v" = pinned
// to the architectural VGPRs (S', dP': consumed by VALU code; in a 512-register kernel hipcc would give a builtin MFMA
// an AGPR accumulator and copy it around every use), "a" = accumulation registers (dK^T, dV^T: resident).
// hipcc pads nothing around asm: every consumer of an accumulator sits at least two MFMAs downstream of its last
// MFMA, and tools/mfma_hazard_audit.py checks the distances in the built code object (hipcc is free to move a tile).
#define FA_W4_WAIT(x) "s_waitcnt lgkmcnt(" x ")\n\t"
#define FA_W4_NOWAIT(x) ""
#define FA_W4_STREAM_IMPL(NAME, TAG, OPC, WAIT)                                                                                     \
    struct NAME##_##TAG {                                                                                             \
        /* request kinds: A = Q rows + K rows of both key blocks, B = one row fragment, T = two transposed 4-row blocks */ \
        template <int N> static __device__ __forceinline__ void fa_v(unsigned qa, unsigned ka, s16x8& r0, s16x8& r1, s16x8& r2, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:8192\n\t" WAIT("%8") OPC " %3, %6, %7, %3" \
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "+v"(c) : "v"(qa), "v"(ka), "v"(a), "v"(b), "n"(N));          \
        }                                                                                                               \
        template <int N> static __device__ __forceinline__ void fa_a(unsigned qa, unsigned ka, s16x8& r0, s16x8& r1, s16x8& r2, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:8192\n\t" WAIT("%8") OPC " %3, %6, %7, %3" \
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "+a"(c) : "v"(qa), "v"(ka), "v"(a), "v"(b), "n"(N));          \
        }                                                                                                               \
        /* KR: K rows of key block 0 stay in registers, a group of the S' chains requests Q rows + K rows of block 1 only */ \
        template <int N> static __device__ __forceinline__ void f2_v(unsigned qa, unsigned ka, s16x8& r0, s16x8& r2, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4 offset:8192\n\t" WAIT("%7") OPC " %2, %5, %6, %2" \
                         : "=&v"(r0), "=&v"(r2), "+v"(c) : "v"(qa), "v"(ka), "v"(a), "v"(b), "n"(N));                     \
        }                                                                                                               \
        template <int N> static __device__ __forceinline__ void f2_a(unsigned qa, unsigned ka, s16x8& r0, s16x8& r2, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4 offset:8192\n\t" WAIT("%7") OPC " %2, %5, %6, %2" \
                         : "=&v"(r0), "=&v"(r2), "+a"(c) : "v"(qa), "v"(ka), "v"(a), "v"(b), "n"(N));                     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void fb_v(unsigned qa, s16x8& r0, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\t" WAIT("%6") OPC " %1, %3, %4, %1"             \
                         : "=&v"(r0), "+v"(c) : "v"(qa), "v"(a), "v"(b), "n"(OFF), "n"(N));                             \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void fb_a(unsigned qa, s16x8& r0, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b128 %0, %2 offset:%5\n\t" WAIT("%6") OPC " %1, %3, %4, %1"             \
                         : "=&v"(r0), "+a"(c) : "v"(qa), "v"(a), "v"(b), "n"(OFF), "n"(N));                             \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void ft_v(unsigned lo_a, unsigned hi_a, s16x4& lo, s16x4& hi, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b64_tr_b16 %0, %3 offset:%7\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t" WAIT("%8") OPC " %2, %5, %6, %2" \
                         : "=&v"(lo), "=&v"(hi), "+v"(c) : "v"(lo_a), "v"(hi_a), "v"(a), "v"(b), "n"(OFF), "n"(N));     \
        }                                                                                                               \
        template <int N, int OFF> static __device__ __forceinline__ void ft_a(unsigned lo_a, unsigned hi_a, s16x4& lo, s16x4& hi, s16x8 a, s16x8 b, f32x16& c) { \
            asm volatile("ds_read_b64_tr_b16 %0, %3 offset:%7\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t" WAIT("%8") OPC " %2, %5, %6, %2" \
                         : "=&v"(lo), "=&v"(hi), "+a"(c) : "v"(lo_a), "v"(hi_a), "v"(a), "v"(b), "n"(OFF), "n"(N));     \
        }                                                                                                               \
        static __device__ __forceinline__ void v(s16x8 a, s16x8 b, f32x16& c) {                                         \
            asm volatile(OPC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));                                             \
        }                                                                                                               \
        static __device__ __forceinline__ void a(s16x8 a_, s16x8 b, f32x16& c) {                                        \
            asm volatile(OPC " %0, %1, %2, %0" : "+a"(c) : "v"(a_), "v"(b));                                            \
        }                                                                                                               \
    };
FA_W4_STREAM_IMPL(W4Stream, bf16, "v_mfma_f32_32x32x16_bf16", FA_W4_WAIT)
FA_W4_STREAM_IMPL(W4Stream, f16, "v_mfma_f32_32x32x16_f16", FA_W4_WAIT)
FA_W4_STREAM_IMPL(W4StreamNW, bf16, "v_mfma_f32_32x32x16_bf16", FA_W4_NOWAIT)   // the same statements without the wait
FA_W4_STREAM_IMPL(W4StreamNW, f16, "v_mfma_f32_32x32x16_f16", FA_W4_NOWAIT)
template <typename Tag, bool WAITS = true> struct W4Stream;
template <> struct W4Stream<bf16_tag, true> : W4Stream_bf16 {};
template <> struct W4Stream<f16_tag, true> : W4Stream_f16 {};
template <> struct W4Stream<bf16_tag, false> : W4StreamNW_bf16 {};
template <> struct W4Stream<f16_tag, false> : W4StreamNW_f16 {};

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

// The same from inline asm, for the loads inside the stream: the four reads land in the four quarters of the accumulator tuple
// (joined by shufflevectors: no copies), hipcc knows of no load and places no wait — its wait would drain the operand requests
// in flight — and the stream's counted waits cover them (38.7 -> 38.1 cycles per MFMA).
template <int OFF> __device__ __forceinline__ void lds_acc_init_asm(unsigned addr, f32x16& acc) {
    typedef float f32x8_t __attribute__((ext_vector_type(8)));
    f32x4 t0, t1, t2, t3;
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(addr), "n"(OFF), "n"(OFF + 32), "n"(OFF + 64), "n"(OFF + 96));
    const f32x8_t lo = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(t2, t3, 0, 1, 2, 3, 4, 5, 6, 7);
    acc = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}

// ABL != 0: profiling ablations (wrong results on purpose; option dkdv_abl):
//   bit 0: no vector slices (P, dS)   bit 1: no LDS-DMA in the stream, no tile wait, no barrier   bit 2: no LDS operand requests
//   bit 3: no row-constant loads      bit 4: no address updates
//   bit 5: shader-clock stamps around the block loop; wave 0 of workgroup 0 overwrites dk[0..7] with
//          (cycles of the loop, blocks) as two uint32 (tools/w4_cycles.py), and (DS variant) every wave of workgroup 0 leaves its stamps of
//          blocks 0 .. 23 — own stream done / behind the barrier — in dk[32 + 64 w ..] (tools/ds_store_cycles.py)
//   bit 6: (DS variant) no dS stores    bit 7: round 2's tile waits (one tile earlier than needed)
//   option value + 512: one key tile per workgroup under the mask too;  bit 11: no mask branch in the loop
// LDS-DMA pieces of the stream: the LDS address and the soffset come from scalar arithmetic on kernel arguments and the
// block counter (no v_readfirstlane feeds them), so the one wait state M0 needs is all the padding there is
// (fa_common.h's dma16_issue carries the five states a VALU-written SGPR operand would need).
__device__ __forceinline__ void dma16_issue_s(rsrc_s_t rsrc, unsigned lds_dst, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                 :: "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4_issue_s(rsrc_s_t rsrc, unsigned lds_dst, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %0, %2, %3 offen lds"
                 :: "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
}

// Placement of the block's vector work in its 64 MFMA gaps (slice S = the gap behind MFMA S), made by
// tools/gen_dkdv_schedule.py from the gap budgets: an MFMA leaves the wave about 24 cycles of issue, the operand
// requests in front of every even MFMA take 8 .. 16 of them.  MUL / EXP: one element of S' -> P; PC: a packed dword of P;
// SU: a packed dword of dS; ACC: row constants -> an initial accumulator of the next block; DMA: an LDS-DMA piece of the
// tile three blocks ahead; QADDR / TADDR / LADDR: operand addresses to the next tile's buffer.
namespace w4sched {
enum : unsigned char { NONE = 0, OP_MUL, OP_EXP, OP_PC, OP_SU, OP_ACC, OP_LADDR, OP_DMA, OP_QADDR, OP_TADDR };
struct Op { unsigned char op, a, b; };
constexpr Op MUL(int kb, int e) { return {OP_MUL, (unsigned char)kb, (unsigned char)e}; }
constexpr Op EXP(int kb, int e) { return {OP_EXP, (unsigned char)kb, (unsigned char)e}; }
constexpr Op PC(int kb, int m) { return {OP_PC, (unsigned char)kb, (unsigned char)m}; }
constexpr Op SU(int kb, int m) { return {OP_SU, (unsigned char)kb, (unsigned char)m}; }
constexpr Op ACC(int w) { return {OP_ACC, (unsigned char)w, 0}; }
constexpr Op DMA(int j) { return {OP_DMA, (unsigned char)j, 0}; }
constexpr Op QADDR(int i) { return {OP_QADDR, (unsigned char)i, 0}; }
constexpr Op TADDR(int j) { return {OP_TADDR, (unsigned char)j, 0}; }
constexpr Op LADDR{OP_LADDR, 0, 0};
constexpr int kWidth = 5;
// generated by tools/gen_dkdv_schedule.py: 0 cycles over budget in 0 gaps
// slice S (after MFMA S): up to 5 operations
constexpr Op kSched[64][kWidth] = {
    /*  0 (24) */ {DMA(0)},
    /*  1 ( 8) */ {LADDR},
    /*  2 (24) */ {DMA(1)},
    /*  3 ( 8) */ {},
    /*  4 (24) */ {DMA(2)},
    /*  5 ( 8) */ {},
    /*  6 (24) */ {DMA(3)},
    /*  7 ( 8) */ {},
    /*  8 (24) */ {DMA(4)},
    /*  9 (16) */ {},
    /* 10 (24) */ {QADDR(0)},
    /* 11 (16) */ {},
    /* 12 (24) */ {QADDR(1)},
    /* 13 (16) */ {},
    /* 14 (24) */ {QADDR(2)},
    /* 15 (16) */ {},
    /* 16 (24) */ {MUL(0,0), EXP(0,0), MUL(0,1), EXP(0,1)},
    /* 17 (16) */ {MUL(0,2), EXP(0,2), MUL(0,3)},
    /* 18 (24) */ {EXP(0,3), MUL(0,4), EXP(0,4), MUL(0,5)},
    /* 19 (16) */ {EXP(0,5), MUL(0,6), MUL(0,7)},
    /* 20 (24) */ {EXP(0,6), EXP(0,7), PC(0,0), PC(0,1)},
    /* 21 (16) */ {PC(0,2), PC(0,3), MUL(1,0), MUL(1,1)},
    /* 22 (24) */ {EXP(1,0), EXP(1,1), MUL(1,2), MUL(1,3)},
    /* 23 (16) */ {EXP(1,2), EXP(1,3)},
    /* 24 (24) */ {MUL(1,4), EXP(1,4), MUL(1,5), EXP(1,5)},
    /* 25 (12) */ {MUL(1,6), EXP(1,6)},
    /* 26 (24) */ {MUL(1,7), EXP(1,7), PC(1,0), PC(1,1), PC(1,2)},
    /* 27 (12) */ {PC(1,3), MUL(0,8), MUL(0,9)},
    /* 28 (24) */ {EXP(0,8), EXP(0,9), MUL(0,10), MUL(0,11)},
    /* 29 (12) */ {EXP(0,10), MUL(0,12)},
    /* 30 (24) */ {EXP(0,11), EXP(0,12), MUL(0,13), MUL(0,14)},
    /* 31 (12) */ {EXP(0,13), MUL(0,15)},
    /* 32 (24) */ {EXP(0,14), EXP(0,15), PC(0,4), PC(0,5)},
    /* 33 (12) */ {PC(0,6), PC(0,7), MUL(1,8)},
    /* 34 (24) */ {EXP(1,8), MUL(1,9), EXP(1,9), MUL(1,10)},
    /* 35 (12) */ {EXP(1,10), MUL(1,11)},
    /* 36 (24) */ {EXP(1,11), MUL(1,12), EXP(1,12), MUL(1,13)},
    /* 37 (12) */ {EXP(1,13), MUL(1,14)},
    /* 38 (24) */ {EXP(1,14), MUL(1,15), EXP(1,15), PC(1,4)},
    /* 39 (12) */ {PC(1,5), PC(1,6), PC(1,7)},
    /* 40 (24) */ {SU(0,0), SU(0,1)},
    /* 41 (12) */ {SU(0,2)},
    /* 42 (24) */ {SU(0,3), SU(1,0)},
    /* 43 (12) */ {SU(1,1)},
    /* 44 (24) */ {SU(1,2), SU(1,3)},
    /* 45 (12) */ {SU(0,4)},
    /* 46 (24) */ {SU(0,5), SU(0,6)},
    /* 47 (12) */ {SU(0,7)},
    /* 48 (24) */ {SU(1,4), SU(1,5)},
    /* 49 (12) */ {SU(1,6)},
    /* 50 (24) */ {SU(1,7), QADDR(3), QADDR(4), QADDR(5)},
    /* 51 (12) */ {QADDR(6), QADDR(7)},
    /* 52 (24) */ {ACC(0)},
    /* 53 (12) */ {},
    /* 54 (24) */ {ACC(2)},
    /* 55 (12) */ {},
    /* 56 (24) */ {ACC(1), TADDR(0)},
    /* 57 ( 8) */ {TADDR(1)},
    /* 58 (24) */ {ACC(3), TADDR(2)},
    /* 59 ( 8) */ {TADDR(3)},
    /* 60 (24) */ {},
    /* 61 ( 8) */ {},
    /* 62 (24) */ {},
    /* 63 ( 8) */ {},
};
}  // namespace w4sched

// DS: the variant that hands dS to the dQ product kernel (fa_bwd_dq_ds.hip) instead of leaving dQ to a pass that recomputes S
// and dP.  The packed dS of a block — already the B operand of the dK^T products, 4 x 16 bytes per lane — is stored as it
// is: tile (b,h; 32-query block qb; 32-key block kb) of 2 KiB at ds + (((b,h) nqb + qb) nkb32 + kb) 2048 bytes, inside it
// [16-query half s][key r][h][8 queries]: every store instruction writes 1 KiB contiguous, a wave 4 KiB per block, the
// workgroup 16 KiB.  Element j of lane (r, h) is query 16 s + 8 (j >> 2) + 4 h + (j & 3) of the block (the accumulator's
// register order); fa_bwd_dq_ds.hip reads the tiles back transposed (ds_read_b64_tr_b16).  Blocks the causal mask removes
// whole (queries before the wave's first key) are never written; masked elements of the others are written as 0.
// KR (round 3): K rows in registers — the non-causal kernels run at 210 VGPRs with none there, and the operand ring shrinks as the
// S' groups request fewer fragments, so there is room for all of them: KR = 1 key block 0 (2 fragments per S' group, 10-slot ring),
// 2 + four fragments of block 1 (12 slots), 3 every K fragment (Q rows only, 8 slots): 32 operand fragments per block from LDS
// instead of 48 (0.5 KB per MFMA instead of 0.75), 250 - 252 VGPRs, no scratch.
template <typename Tag, bool CAUSAL, int ABL = 0, int TPW = (CAUSAL ? 2 : 1), bool DS = false, int KR = 0>
__global__ __launch_bounds__(256, 1) void bwd_dkdv_w4_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                             const uint16_t* __restrict__ v,
                                                             const uint16_t* __restrict__ dout,
                                                             const float* __restrict__ nlse,
                                                             const float* __restrict__ ndelta, uint16_t* __restrict__ dk,
                                                             uint16_t* __restrict__ dv, int n, int nkt, float c_log2,
                                                             float scale, int nk /* keys; n = query rows; causal: nk >= n, diagonal at key = row + nk - n */,
                                                             uint16_t* __restrict__ ds = nullptr, int nqb = 0, int nkb32 = 0 /* DS: tile grid of the dS workspace */) {
    constexpr int D = 128, NKS = 8, NDB = 4, BK = 256, BQ = 32, NBUF = 4, K1R = KR == 3 ? 8 : (KR == 2 ? 4 : 0), RS = KR == 3 ? 8 : (KR == 2 ? 12 : (KR ? 10 : 16));
    // operand groups requested ahead of use (6 MFMAs).  4 live groups x 3 fragments = 12 ring slots; with 16 a request
    // never lands on a fragment the two MFMAs just issued are still reading (hipcc would pad that hazard with an s_nop)
    constexpr int AHEAD = 3;
    constexpr int K_BYTES = BK * D * 2;          // 64 KiB: the workgroup's K rows (B operand of S)
    constexpr int QT = BQ * D * 2;               // 8 KiB: one 32-row tile of Q (or dO)
    constexpr int BUF = 2 * QT + 1024;           // Q | dO | 64 x -lse/scale | 64 x -delta | 2 x 64 unused (see dma_piece)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Bs = smem + K_BYTES;                   // [NBUF][BUF]: tile t in buffer t % NBUF

    // A workgroup works through TPW key tiles of one (b,h).  Under the causal mask the pair is heavy + light
    // (tiles jp and nkt-1-jp): every workgroup then carries the same number of query blocks (as fa_bwd_dkdv_mfma.hip).
    const int npair = (nkt + TPW - 1) / TPW;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / npair;
    const int jp = L - bh * npair;
    auto tile_of = [&](int ip) -> int {
        if (TPW == 1) return jp;
        const int t = CAUSAL ? (ip == 0 ? jp : nkt - 1 - jp) : TPW * jp + ip;
        if (CAUSAL && ip > 0 && t <= jp) return -1;   // odd tile count: the middle tile is its own pair
        return t < nkt ? t : -1;
    };
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t base = (size_t)bh * n * D, kvbase = (size_t)bh * nk * D;   // q / dO rows; k / v / dk / dv rows
    const size_t rbase = (size_t)bh * n;
    const int coff = nk - n;
    int key0 = 0, kw0 = 0;                       // first key of the current tile / of this wave in it

    const rsrc_s_t k_rs = make_rsrc_s(k + kvbase, (unsigned)nk * D * 2);
    const rsrc_s_t q_rs = make_rsrc_s(q + base, (unsigned)n * D * 2);
    const rsrc_s_t o_rs = make_rsrc_s(dout + base, (unsigned)n * D * 2);
    const rsrc_s_t l_rs = make_rsrc_s(nlse + rbase, (unsigned)n * 4);
    const rsrc_s_t d_rs = make_rsrc_s(ndelta + rbase, (unsigned)n * 4);
    const rsrc_s_t lr_rs = (w & 1) ? d_rs : l_rs;   // this wave's row-constant source (wave-uniform select)
    const buf_rsrc_t v_rs = make_rsrc(v + kvbase, (unsigned)nk * D * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, D);

    int qs_first = 0, nblk = 0;                  // first query of the current tile's sweep, 32-query blocks in it
    const unsigned bbase = lds_addr_of(Bs);
    // one LDS-DMA piece (1 KiB = 4 rows) of tile t: J = 0, 1 -> Q, J = 2, 3 -> dO, J = 4 -> the wave's row constants
    auto dma_piece = [&](auto jc, int t) {
        constexpr int J = decltype(jc)::value;
        const int qs = qs_first + BQ * t;
        const unsigned b = bbase + (t & (NBUF - 1)) * BUF;
        if constexpr (J < 4) {
            const int pc = w + 4 * (J & 1);
            dma16_issue_s(J < 2 ? q_rs : o_rs, b + (J < 2 ? 0 : QT) + pc * 1024, dma_voff, (qs + 4 * pc) * 2 * D);
        } else {
            // row constants, 64 floats per piece (rows qs .. qs+63; the second half is never read); rows >= n read as 0:
            // harmless, their dO is 0.  Waves 0 / 1 bring -lse/scale / -delta; waves 2 / 3 issue the same pieces into an
            // unused area: the stream must not branch (hipcc sinks vector work across a branch, out of its MFMA gap) and
            // every wave then has the same number of pieces in flight for the counted waits.
            dma4_issue_s(lr_rs, b + 2 * QT + 256 * w, lane * 4, qs * 4);
        }
    };
    auto stage = [&](int t) { for_each_const([&](auto jc) { dma_piece(jc, t); }, std::make_integer_sequence<int, 5>{}); };
    // all but this wave's newest tile (4 pieces + 1 row-constant piece) have landed.  DS: the block's four dS stores were
    // issued behind those pieces and count too (vector-memory operations of all kinds retire in issue order); from the
    // stream's second block on the previous block's stores may still be in flight as well: they have a block to complete.
    // The stores cost the kernel 7 % wherever they sit in the block (stores into a private, L2-resident tile cost the same as
    // the real ones; not waiting for them saves nothing; in the nearly empty gaps of the dP' chains they cost the same:
    // profiles/r02_ds_handover.md) — about a third of it in cycles, the rest in clock: the chip is power-limited.
    auto wait_tiles = [&](bool first_block = false) {
        if (ABL & 128) {   // round 2's counts: the tile TWO blocks ahead has landed (one block of slack more than needed)
            if (!DS || (ABL & 64)) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (first_block) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        } else if (!DS || (ABL & 64)) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // this block's and the previous block's pieces may be in flight
        else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");                            // ... and the stores of this and the two blocks before
    };
    const int ds_voff = 32 * (lane & 31) + 16 * (lane >> 5);
    unsigned long long dsp = 0;                  // DS: this wave's two dS tiles of the current block (wave-uniform address)

    s16x8 vf[2][NKS];
    f32x16 dka[2][NDB], dva[2][NDB];
    unsigned kaddr[NKS], qaddr[NKS], tlo[NDB], thi[NDB], laddr;
    // first block this wave computes: earlier ones hold only queries before its first key (causal)
    int fb = 0;
    unsigned stamp_cycles = 0;
    // ---- the stream.  Operand groups g = 0 .. 31 of a block, two MFMAs each (key block 0, then 1):
    //   g =  0 ..  7  S'[kb]  += Q[ks] K[kb][ks]          reads: Q rows, K rows of block 0, K rows of block 1
    //   g =  8 .. 15  dP'[kb] += dO[ks] V[kb][ks]         reads: dO rows                    (V in registers)
    //   g = 16 .. 23  dV^T[kb][db] += dO^T[s][db] P[kb][s]    reads: two transposed 4-row blocks of dO
    //   g = 24 .. 31  dK^T[kb][db] += Q^T[s][db] dS[kb][s]    reads: two transposed 4-row blocks of Q
    // Group g + 3 is requested right before group g's MFMAs; groups 32 .. 34 are groups 0 .. 2 of the next block.
    s16x8 ring[RS];
    s16x8 kreg[KR ? NKS : 1];   // KR: K rows of key block 0, lane (r, h): K[kw0 + r][16 ks + 8 h ..]
    s16x8 kreg1[K1R ? K1R : 1]; // KR = 2: K rows of key block 1, k-steps 0 .. 3
    f32x16 sacc[2], pacc[2];
    u32x4 pp[2][2], sp[2][2];
    // fragments a group of the S' chains requests: Q rows, + K rows of block 0 unless they are in registers (KR), + K rows of
    // block 1 unless they are (KR = 2, k-steps 0 .. 3)
    struct G {
        static constexpr int f0(int i) { return 1 + (KR ? 0 : 1) + ((KR >= 2 && i < (KR == 3 ? 8 : 4)) ? 0 : 1); }
        static constexpr int reads(int g) { return (g % 32) < 8 ? f0(g % 32) : ((g % 32) < 16 ? 1 : 2); }
        static constexpr int opidx(int g) {
            int s = 0;
            for (int i = 0; i < ((g % 32) < 8 ? (g % 32) : 8); ++i) s += f0(i);
            return (g % 32) < 8 ? s : s + (g % 32) - 8;
        }
        static constexpr int slot(int g) { return opidx(g) % RS; }
    };
    static_assert((G::opidx(31) + 1) % RS == 0, "ring bookkeeping: a block's operand slots must be a multiple of the ring");
    auto fetch = [&](auto gc) {
        constexpr int g = decltype(gc)::value % 32, ph = g / 8, i = g % 8, s0 = G::slot(g);
        if constexpr (ph == 0) {
            ring[s0] = lds_b128_asm<0>(qaddr[i]);
            if constexpr (!KR) ring[(s0 + 1) % RS] = lds_b128_asm<0>(kaddr[i]);
            if constexpr (i >= K1R) ring[(s0 + G::f0(i) - 1) % RS] = lds_b128_asm<32 * 2 * D>(kaddr[i]);
        } else if constexpr (ph == 1) {
            ring[s0] = lds_b128_asm<QT>(qaddr[i]);
        } else {
            constexpr int off = (ph == 2 ? QT : 0) + (i / 4) * 16 * 2 * D;
            ring[s0] = cat8(lds_tr16_asm<off>(tlo[i % 4]), lds_tr16_asm<off>(thi[i % 4]));
        }
    };

    // Mask (diagonal blocks under the causal mask, keys past n): applied to the INITIAL accumulator of S' — a masked
    // element starts at -1e30, so P = exp2(c S') = 0 and dS = 0 — in a short wave-uniform branch between two blocks.
    // The stream itself has one form and no branch: two copies of it cost hipcc's register allocator 1.1 KB of spills,
    // and across a branch inside it hipcc sinks vector work out of its MFMA gap to the first use.  Register i holds query qs + 4 h + rc(i), rc(i) = (i & 3) + 8 (i >> 2); it is masked
    // when it precedes this lane's key (causal) or the key lies past n: rc(i) < thr, one per-lane threshold per key block.
    auto mask_init = [&](int blk) {   // blk: the block whose initial accumulators were just requested
        const int qs = qs_first + BQ * blk;
        const bool need_mask = (CAUSAL && (kw0 + 63 - coff > qs)) || (kw0 + 64 > nk);   // wave-uniform
        if (need_mask) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sacc[0]), "+v"(sacc[1]));   // the row constants were read from asm: hipcc places no wait
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int key = kw0 + 32 * kb + r;
                const int thr = key >= nk ? 64 : (CAUSAL ? key - coff - qs - 4 * h : -1);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) < thr) sacc[kb][i] = -1e30f;
            }
        }
    };
    auto block = [&](int blk) {
        const int dlt = ((blk + 1) & (NBUF - 1)) ? BUF : -(NBUF - 1) * BUF;   // to the next tile's buffer
        // Vector work, cut into single instructions' worth and placed by the table above (w4sched::kSched): MUL + EXP: one
        // element of S' -> P = exp2(c S') in place (the f32 P is what dS is made from, as in the dQ kernel);  PC: one packed
        // dword of P from two finished elements (at least one gap behind its exp2's: no dependency stall, no hazard nop);
        // SU: one packed dword of dS = P dP' (f32 P, one rounding).
        auto PC = [&](auto kbc, auto mc) {
            constexpr int kb = decltype(kbc)::value, m = decltype(mc)::value;
            pp[kb][m >> 2][m & 3] = pack2<Tag>(sacc[kb][2 * m], sacc[kb][2 * m + 1]);
        };
        auto SU = [&](auto kbc, auto mc) {
            constexpr int kb = decltype(kbc)::value, m = decltype(mc)::value;
            sp[kb][m >> 2][m & 3] = pack2<Tag>(sacc[kb][2 * m] * pacc[kb][2 * m], sacc[kb][2 * m + 1] * pacc[kb][2 * m + 1]);
        };
        using std::integral_constant;
        // The work that follows MFMA S of the block (S = 0 .. 63).  The table keeps the deadlines — P[kb][s] feeds MFMAs
        // 32 + 8 s ..., dS[kb][s] MFMAs 48 + 8 s ...; S'[kb] is complete after MFMA 14 + kb, dP'[kb] after MFMA 30 + kb, and a
        // consumer sits at least two MFMAs behind the chain it reads (hipcc pads nothing around the asm MFMAs;
        // tools/mfma_hazard_audit.py checks the built code) — and every gap within its issue budget: 42.1 -> 39.3 cycles per
        // MFMA against the first hand placement, bitwise the same results (profiles/r02_cycles_dkdv.md).
        // DS: the four stores of the block's packed dS, in the gaps behind the last SU (tools/gen_dkdv_schedule.py --ds
        // places them there and moves nothing else: DSST(0,0) 52, DSST(0,1) 53, DSST(1,0) 54, DSST(1,1) 55)
        auto DSST = [&](auto kbc, auto hc) {
            constexpr int kb = decltype(kbc)::value, half = decltype(hc)::value;
            const int vo = ds_voff;                 // (locals: hipcc does not capture a variable that only an asm operand names)
            const unsigned long long p = dsp;
            const u32x4 x = sp[kb][half];
            // nt: the tiles are read once, by another kernel (nt / sc1 / plain: 3.476 / 3.466 / 3.506 ms)
            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt" :: "v"(vo), "v"(x), "s"(p), "n"(2048 * kb + 1024 * half));
        };
        auto slice = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if constexpr (DS && !(ABL & 64) && S >= 52 && S < 56) DSST(integral_constant<int, ((S - 52) / 2)>{}, integral_constant<int, ((S - 52) % 2)>{});
            {   // ABL bits: see the kernel's header comment
                for_each_const([&](auto jc) {
                    constexpr w4sched::Op o = w4sched::kSched[S][decltype(jc)::value];
                    constexpr int a = o.a, b = o.b;
                    if constexpr (o.op == w4sched::OP_MUL && !(ABL & 1)) sacc[a][b] *= c_log2;
                    else if constexpr (o.op == w4sched::OP_EXP && !(ABL & 1)) sacc[a][b] = __builtin_amdgcn_exp2f(sacc[a][b]);
                    else if constexpr (o.op == w4sched::OP_PC && !(ABL & 1)) PC(integral_constant<int, a>{}, integral_constant<int, b>{});
                    else if constexpr (o.op == w4sched::OP_SU && !(ABL & 1)) SU(integral_constant<int, a>{}, integral_constant<int, b>{});
                    else if constexpr (o.op == w4sched::OP_ACC && !(ABL & 8)) {
                        if constexpr (a < 2) lds_acc_init_asm<0>(laddr, sacc[a]);
                        else lds_acc_init_asm<256>(laddr, pacc[a - 2]);
                    }
                    else if constexpr (o.op == w4sched::OP_DMA && !(ABL & 2)) dma_piece(integral_constant<int, a>{}, blk + 3);
                    else if constexpr (o.op == w4sched::OP_LADDR && !(ABL & 16)) laddr += dlt;
                    else if constexpr (o.op == w4sched::OP_QADDR && !(ABL & 16)) qaddr[a] += dlt;
                    else if constexpr (o.op == w4sched::OP_TADDR && !(ABL & 16)) { tlo[a] += dlt; thi[a] += dlt; }
                }, std::make_integer_sequence<int, w4sched::kWidth>{});
            }
        };
        auto group = [&](auto gc) {
            constexpr int g = decltype(gc)::value, ph = g / 8, i = g % 8, s0 = G::slot(g);
            // one counted wait per TWO groups: an even group also waits for the next group's operands (requested a group later:
            // a lead of two groups for those), an odd group issues no wait — one instruction less per four MFMAs (39.3 -> 38.7
            // cycles per MFMA)
            constexpr bool WAITS = (g % 2) == 0;
            // (the row-constant reads issued since group g + 1 was requested — in the slices behind MFMAs 2 g - 4 .. 2 g - 1 — are
            // in the queue too)
            constexpr int ACCS = [] {
                int c = 0;
                for (int S = 2 * g - 4; S <= 2 * g - 1; ++S)
                    for (int j = 0; j < w4sched::kWidth; ++j)
                        if (w4sched::kSched[(S + 64) % 64][j].op == w4sched::OP_ACC) c += 4;
                return c;
            }();
            constexpr int NWAIT = G::reads(g + 2) + G::reads(g + 3) + ((ABL & 8) ? 0 : ACCS);
            using M = W4Stream<Tag, WAITS>;
            constexpr int g2 = (g + AHEAD) % 32, ph2 = g2 / 8, i2 = g2 % 8, t0 = G::slot(g2);   // the group requested here
            // first operand of the MFMAs, second operand / accumulator of key block 0 and 1
            const s16x8 opa = ring[s0];
            s16x8 opb0, opb1;
            if constexpr (ph == 0) {
                opb0 = KR ? kreg[KR ? i : 0] : ring[(s0 + 1) % RS];
                opb1 = i < K1R ? kreg1[i < K1R ? i : 0] : ring[(s0 + G::f0(i) - 1) % RS];
            }
            else if constexpr (ph == 1) { opb0 = vf[0][i]; opb1 = vf[1][i]; }
            else if constexpr (ph == 2) { opb0 = *reinterpret_cast<s16x8*>(&pp[0][i / 4]); opb1 = *reinterpret_cast<s16x8*>(&pp[1][i / 4]); }
            else { opb0 = *reinterpret_cast<s16x8*>(&sp[0][i / 4]); opb1 = *reinterpret_cast<s16x8*>(&sp[1][i / 4]); }
            f32x16& acc0 = ph == 0 ? sacc[0] : (ph == 1 ? pacc[0] : (ph == 2 ? dva[0][i % 4] : dka[0][i % 4]));
            f32x16& acc1 = ph == 0 ? sacc[1] : (ph == 1 ? pacc[1] : (ph == 2 ? dva[1][i % 4] : dka[1][i % 4]));
            if constexpr (ABL & 4) {
                if constexpr (ph < 2) M::v(opa, opb0, acc0);
                else M::a(opa, opb0, acc0);
            } else if constexpr (ph2 == 0 && i2 < K1R) {   // Q rows only
                if constexpr (ph < 2) M::template fb_v<NWAIT, 0>(qaddr[i2], ring[t0], opa, opb0, acc0);
                else M::template fb_a<NWAIT, 0>(qaddr[i2], ring[t0], opa, opb0, acc0);
            } else if constexpr (ph2 == 0 && KR) {
                if constexpr (ph < 2) M::template f2_v<NWAIT>(qaddr[i2], kaddr[i2], ring[t0], ring[(t0 + 1) % RS], opa, opb0, acc0);
                else M::template f2_a<NWAIT>(qaddr[i2], kaddr[i2], ring[t0], ring[(t0 + 1) % RS], opa, opb0, acc0);
            } else if constexpr (ph2 == 0) {
                if constexpr (ph < 2) M::template fa_v<NWAIT>(qaddr[i2], kaddr[i2], ring[t0], ring[(t0 + 1) % RS], ring[(t0 + 2) % RS], opa, opb0, acc0);
                else M::template fa_a<NWAIT>(qaddr[i2], kaddr[i2], ring[t0], ring[(t0 + 1) % RS], ring[(t0 + 2) % RS], opa, opb0, acc0);
            } else if constexpr (ph2 == 1) {
                static_assert(ph < 2, "dO rows are requested during the S' and dP' chains");
                M::template fb_v<NWAIT, QT>(qaddr[i2], ring[t0], opa, opb0, acc0);
            } else {
                constexpr int off = (ph2 == 2 ? QT : 0) + (i2 / 4) * 16 * 2 * D;
                s16x4 lo, hi;
                if constexpr (ph < 2) M::template ft_v<NWAIT, off>(tlo[i2 % 4], thi[i2 % 4], lo, hi, opa, opb0, acc0);
                else M::template ft_a<NWAIT, off>(tlo[i2 % 4], thi[i2 % 4], lo, hi, opa, opb0, acc0);
                ring[t0] = cat8(lo, hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            slice(integral_constant<int, 2 * g>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ph < 2) M::v(opa, opb1, acc1);
            else M::a(opa, opb1, acc1);
            __builtin_amdgcn_sched_barrier(0);
            slice(integral_constant<int, 2 * g + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        for_each_const(group, std::make_integer_sequence<int, 32>{});
    };

#pragma unroll 1
    for (int ip = 0; ip < TPW; ++ip) {
    const int kt = tile_of(ip);
    if (TPW > 1 && kt < 0) break;
    key0 = kt * BK;
    kw0 = key0 + 64 * w;
    qs_first = CAUSAL ? (max(0, key0 - coff) / BQ) * BQ : 0;   // earlier queries see none of this tile's keys
    fb = CAUSAL ? (max(0, kw0 - coff) - qs_first) / BQ : 0;    // (nq == nk: 2 w)
    nblk = (n - qs_first + BQ - 1) / BQ;
    if (DS) dsp = (unsigned long long)(uintptr_t)ds + (((unsigned long long)bh * nqb + qs_first / BQ + fb) * nkb32 + (kw0 >> 5)) * 2048ull;
    // ---- prologue of a key tile: K tile, V fragments (B operand of dP = dO V^T), query tiles 0 .. 2
    // TPW > 1: the lane-constant address arithmetic of the prologue and the epilogue does not depend on the tile, and hipcc
    // would hoist it out of the tile loop and keep ~40 registers of it alive through the stream (256 VGPRs + scratch);
    // an opaque copy of the lane id per tile keeps it where it is used
    int lane_p = lane;
    if (TPW > 1) asm volatile("" : "+v"(lane_p));
    const int r_p = lane_p & 31, h_p = lane_p >> 5;
    dma_stage_tile<D, BK, 4>(k_rs, Ks, key0, TPW > 1 ? dma_lane_voff<D>(lane_p, w, D) : dma_voff, w);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) vf[kb][ks] = buf_load_frag(v_rs, frag_off(kw0 + 32 * kb + r_p, 16 * ks + 8 * h_p, D, false));
    stage(0);
    stage(1);
    stage(2);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < NDB; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dka[kb][t][i] = 0.f; dva[kb][t][i] = 0.f; }
    // pin the zeroing HERE, under the prologue's loads: hipcc sinks it to the stream's entry otherwise — 256 v_accvgpr_write
    // (2 500 cycles) between a wave's last feed-only barrier and its first block, with the other waves waiting at the next barrier
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
        asm volatile("" : "+a"(dka[kb][0]), "+a"(dka[kb][1]), "+a"(dka[kb][2]), "+a"(dka[kb][3]), "+a"(dva[kb][0]), "+a"(dva[kb][1]), "+a"(dva[kb][2]), "+a"(dva[kb][3]));
    // lane-constant operand addresses (LDS bytes).  Q / dO rows and K rows: row r, chunk 2 ks + h; transposed reads:
    // 4-row blocks at rows 4 h + tq (+8), chunks 4 db + 2 g16 + (tp >> 1); everything else is an immediate offset
    // (the swizzle depends on the row modulo 16 only).  The tile addresses move from buffer to buffer during the stream.
    {
        const unsigned kbase = lds_addr_of(Ks) + 64 * w * 2 * D, b0 = bbase + (fb & (NBUF - 1)) * BUF;
        const int li = lane_p & 15, g16 = (lane_p >> 4) & 1, tq = li >> 2, tp = li & 3;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int off = TileSwz<D>::off(r_p, 2 * ks + h_p);
            kaddr[ks] = kbase + off;
            qaddr[ks] = b0 + off;
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int ch = 4 * db + 2 * g16 + (tp >> 1);
            tlo[db] = b0 + TileSwz<D>::off(4 * h_p + tq, ch) + 8 * (tp & 1);
            thi[db] = b0 + TileSwz<D>::off(4 * h_p + tq + 8, ch) + 8 * (tp & 1);
        }
        laddr = b0 + 2 * QT + 16 * h_p;
    }
    dma_wait_all();
    __syncthreads();
    if constexpr (KR) {   // the K tile has landed: this lane's fragments of key block 0 (compiler-visible reads: hipcc waits before their first use)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kreg[ks] = lds_b128_at(kaddr[ks]);
#pragma unroll
        for (int ks = 0; ks < K1R; ++ks) kreg1[ks] = lds_b128_at(kaddr[ks] + 32 * 2 * D);
    }
    // the V fragments are first used inside the stream: make hipcc wait for them here, not in the loop (its vmcnt wait
    // there would also drain the LDS-DMA of the tiles in flight)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ks += 4)
            asm volatile("" : "+v"(vf[kb][ks]), "+v"(vf[kb][ks + 1]), "+v"(vf[kb][ks + 2]), "+v"(vf[kb][ks + 3]));

    // feed-only blocks (causal: queries before this wave's first key): the wave's share of the LDS-DMA and the barriers.
    // The set-up of the wave's first block — row constants into the accumulators, the mask, operand groups 0 .. 2 in flight —
    // sits in FRONT of the last feed-only barrier (tile fb is visible since the barrier of iteration fb - 2, or since the
    // prologue when fb <= 2): behind it the other waves, already streaming, would wait for the late starter at the next barrier.
    const int nfeed = min(fb, nblk);
    for (int blk = 0;; ++blk) {
        const bool last = blk >= nfeed - 1;   // (also when there is no feed-only block)
        if (blk < nfeed) stage(blk + 3);
        if (last && fb < nblk) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing older may sit in the LDS queue: the counts are exact
            lds_acc_init<0>(laddr, sacc[0]);
            lds_acc_init<0>(laddr, sacc[1]);
            mask_init(fb);
            lds_acc_init<256>(laddr, pacc[0]);
            lds_acc_init<256>(laddr, pacc[1]);
            fetch(std::integral_constant<int, 0>{});
            fetch(std::integral_constant<int, 1>{});
            fetch(std::integral_constant<int, 2>{});
        }
        if (blk < nfeed) {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // (no stores here: the count of the plain kernel)
            __builtin_amdgcn_s_barrier();
        }
        if (last) break;
    }
    if (fb < nblk) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t_begin = 0;
        if (ABL & 32) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin)::"memory");
#pragma unroll 1
        for (int blk = fb; blk < nblk; ++blk) {
            block(blk);
            if (DS && (ABL & 32) && L == 0 && blk < 24) {   // per-wave stamps of blocks 0 .. 23 of workgroup 0's last tile: own work done, dk[32 + 64 w + 2 blk]
                unsigned long long t_now;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory");
                if (lane == 0) reinterpret_cast<unsigned*>(dk)[32 + 64 * w + 2 * blk] = (unsigned)t_now;
            }
            if (!(ABL & 2048)) mask_init(blk + 1);   // on the next block's initial accumulators, outside the stream (the only branch)
            if (DS) dsp += (unsigned long long)nkb32 * 2048;
            if (!(ABL & 2)) {
                wait_tiles(blk == fb);
                __builtin_amdgcn_s_barrier();
            }
            if (DS && (ABL & 32) && L == 0 && blk < 24) {   // ... and behind the barrier: dk[32 + 64 w + 2 blk + 1]
                unsigned long long t_now;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory");
                if (lane == 0) reinterpret_cast<unsigned*>(dk)[32 + 64 * w + 2 * blk + 1] = (unsigned)t_now;
            }
        }
        if (ABL & 32) {
            unsigned long long t_end;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
            stamp_cycles = (unsigned)(t_end - t_begin);
        }
    }

    // ---- epilogue: dK = scale * dK^T (transposed back on the store), dV.  The resident accumulators were last
    // written from asm: pad the MFMA -> accumulator read hazard by hand, then everything below is compiler-visible.
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+a"(dka[0][0]), "+a"(dka[0][1]), "+a"(dka[0][2]), "+a"(dka[0][3]), "+a"(dka[1][0]), "+a"(dka[1][1]),
                   "+a"(dka[1][2]), "+a"(dka[1][3]), "+a"(dva[0][0]), "+a"(dva[0][1]), "+a"(dva[0][2]), "+a"(dva[0][3]),
                   "+a"(dva[1][0]), "+a"(dva[1][1]), "+a"(dva[1][2]), "+a"(dva[1][3]));
    dma_wait_all();   // nothing of this workgroup may still be writing LDS when the next one takes the CU
    // a wave only ever read its own 64 K rows: that slice of the K tile is its staging area for whole-row stores
    char* stg = Ks + w * 64 * D * 2;
    int lane_e = lane;
    if (TPW > 1) asm volatile("" : "+v"(lane_e));
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
        store_rows_via_lds<D>(stg, vals, dk + kvbase, kw0 + 32 * kb, nk, lane_e, D);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * db + g][0] = pack2_rn<Tag>(dva[kb][db][4 * g + 0], dva[kb][db][4 * g + 1]);
                vals[4 * db + g][1] = pack2_rn<Tag>(dva[kb][db][4 * g + 2], dva[kb][db][4 * g + 3]);
            }
        store_rows_via_lds<D>(stg + 32 * D * 2, vals, dv + kvbase, kw0 + 32 * kb, nk, lane_e, D);
    }
    if (TPW > 1) __syncthreads();   // the K tile and the query-tile buffers are about to be refilled
    }   // key tiles of this workgroup
    if ((ABL & 32) && L == 0 && w == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            reinterpret_cast<unsigned*>(dk)[0] = stamp_cycles;
            reinterpret_cast<unsigned*>(dk)[1] = (unsigned)(nblk - fb);
        }
    }
}

template <typename Tag>
static hipError_t launch_dkdv_w4_t(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st, void* ds) {
    constexpr int D = 128, BK = 256;
    const int64_t nk = a.nk > 0 ? a.nk : a.n;
    const int nkt = (int)((nk + BK - 1) / BK);
    const size_t smem = (size_t)BK * D * 2 + 4 * (2 * 32 * D * 2 + 1024);
    const float c = a.scale * 1.4426950408889634f;
    // key tiles per workgroup: 2 under the causal mask (heavy + light pair), else 1
    const bool tpw1 = a.causal && (option(OPT_DKDV_ABL) & 512);   // experiment: one key tile per workgroup under the mask too
    dim3 grid((unsigned)((a.causal && !tpw1 ? (nkt + 1) / 2 : nkt) * a.bh));
    ProfScope ps(K_BWD_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k, (const uint16_t*)a.v,
                           (const uint16_t*)a.dout, nlse, ndelta, (uint16_t*)a.dk, (uint16_t*)a.dv, (int)a.n, nkt, c, a.scale, (int)nk,
                           (uint16_t*)ds, ds_tile_rows(a.n), ds_tile_cols(nk));
        return hipGetLastError();
    };
    // K rows in registers: the kernels keep ALL of the wave's K fragments there (KR = 3: 250 - 256 VGPRs, no scratch;
    // non-causal 1.738 ms per launch against 1.744 with block 0 + half of block 1, 1.751 with block 0 only, 1.781 with none,
    // interleaved; config 3 (causal) 3.294 / 3.296 / 3.323 / 3.367).
    // Option dkdv_kreg: 2 = none (the round-2 form), 3 = key block 0 only, 4 = block 0 + four fragments of block 1.
    const int kro = option(OPT_DKDV_KREG);
    const int kr = kro == 2 ? 0 : (kro == 3 ? 1 : (kro == 4 ? 2 : 3));
    if (ds && option(OPT_DKDV_ABL) >= 32) {
        if constexpr (std::is_same<Tag, bf16_tag>::value) switch (option(OPT_DKDV_ABL)) {
            case 64: return a.causal ? launch(bwd_dkdv_w4_kernel<Tag, true, 64, 2, true, 3>) : launch(bwd_dkdv_w4_kernel<Tag, false, 64, 1, true, 3>);
            case 128: return a.causal ? launch(bwd_dkdv_w4_kernel<Tag, true, 128, 2, true, 3>) : launch(bwd_dkdv_w4_kernel<Tag, false, 128, 1, true, 3>);
            case 32: return a.causal ? launch(bwd_dkdv_w4_kernel<Tag, true, 32, 2, true, 3>) : launch(bwd_dkdv_w4_kernel<Tag, false, 32, 1, true, 3>);
            case 96: return a.causal ? launch(bwd_dkdv_w4_kernel<Tag, true, 96, 2, true, 3>) : launch(bwd_dkdv_w4_kernel<Tag, false, 96, 1, true, 3>);
            case 544: if (a.causal) return launch(bwd_dkdv_w4_kernel<Tag, true, 32, 1, true, 3>); break;
            case 608: if (a.causal) return launch(bwd_dkdv_w4_kernel<Tag, true, 96, 1, true, 3>); break;
            case 2080: if (a.causal) return launch(bwd_dkdv_w4_kernel<Tag, true, 2080, 2, true, 3>); break;   // stamps, no mask branch in the loop
            case 512: if (a.causal) return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 1, true, 3>); break;
            default: break;
        }
    }
    if (ds) {
        // (causal + dS stores + f16: KR = 3 is one register over — 16 bytes of scratch, 1.092 against 1.066 ms: KR = 2 there)
        if (a.causal) switch ((kr == 3 && std::is_same<Tag, f16_tag>::value) ? 2 : kr) {
            case 0: return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 2, true, 0>);
            case 1: return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 2, true, 1>);
            case 2: return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 2, true, 2>);
            default: return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 2, true, 3>);
        }
        switch (kr) {
            case 0: return launch(bwd_dkdv_w4_kernel<Tag, false, 0, 1, true, 0>);
            case 1: return launch(bwd_dkdv_w4_kernel<Tag, false, 0, 1, true, 1>);
            case 2: return launch(bwd_dkdv_w4_kernel<Tag, false, 0, 1, true, 2>);
            default: return launch(bwd_dkdv_w4_kernel<Tag, false, 0, 1, true, 3>);
        }
    }
    if constexpr (std::is_same<Tag, bf16_tag>::value) {
        if (option(OPT_DKDV_ABL) == 128) return a.causal ? launch(bwd_dkdv_w4_kernel<Tag, true, 128, 2, false, 3>) : launch(bwd_dkdv_w4_kernel<Tag, false, 128, 1, false, 3>);
        if (!a.causal) switch (option(OPT_DKDV_ABL)) {   // profiling ablations: see the kernel's header comment
            case 1: return launch(bwd_dkdv_w4_kernel<Tag, false, 1>);
            case 2: return launch(bwd_dkdv_w4_kernel<Tag, false, 2>);
            case 4: return launch(bwd_dkdv_w4_kernel<Tag, false, 4>);
            case 8: return launch(bwd_dkdv_w4_kernel<Tag, false, 8>);
            case 16: return launch(bwd_dkdv_w4_kernel<Tag, false, 16>);
            case 3: return launch(bwd_dkdv_w4_kernel<Tag, false, 3>);
            case 31: return launch(bwd_dkdv_w4_kernel<Tag, false, 31>);
            case 32: return launch(bwd_dkdv_w4_kernel<Tag, false, 32>);
            case 33: return launch(bwd_dkdv_w4_kernel<Tag, false, 33>);
            case 34: return launch(bwd_dkdv_w4_kernel<Tag, false, 34>);
            case 36: return launch(bwd_dkdv_w4_kernel<Tag, false, 36>);
            case 40: return launch(bwd_dkdv_w4_kernel<Tag, false, 40>);
            case 59: return launch(bwd_dkdv_w4_kernel<Tag, false, 59>);
            case 63: return launch(bwd_dkdv_w4_kernel<Tag, false, 63>);
            case 27: return launch(bwd_dkdv_w4_kernel<Tag, false, 27>);
            default: break;
        }
    }
    if (a.causal) switch (kr) {
        case 0: return launch(bwd_dkdv_w4_kernel<Tag, true>);
        case 1: return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 2, false, 1>);
        case 2: return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 2, false, 2>);
        default: return launch(bwd_dkdv_w4_kernel<Tag, true, 0, 2, false, 3>);
    }
    switch (kr) {
        case 0: return launch(bwd_dkdv_w4_kernel<Tag, false>);
        case 1: return launch(bwd_dkdv_w4_kernel<Tag, false, 0, 1, false, 1>);
        case 2: return launch(bwd_dkdv_w4_kernel<Tag, false, 0, 1, false, 2>);
        default: return launch(bwd_dkdv_w4_kernel<Tag, false, 0, 1, false, 3>);
    }
}

bool bwd_dkdv_w4_supported(int dtype, int64_t d) { return (dtype == 1 || dtype == 2) && d == 128; }

hipError_t launch_bwd_dkdv_w4(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st, void* ds) {
    return a.dtype == 2 ? launch_dkdv_w4_t<bf16_tag>(a, nlse, ndelta, st, ds) : launch_dkdv_w4_t<f16_tag>(a, nlse, ndelta, st, ds);
}

}  // namespace fa
