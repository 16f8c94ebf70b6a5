// Shared device helpers for the gfx950 FlashAttention kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace fa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct bf16_tag {};
struct f16_tag {};

// ---- scalar element conversion (generic path) ----
template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<__half>(__half x) { return __half2float(x); }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 x) { return __bfloat162float(x); }

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ __half from_f32<__half>(float x) { return __float2half_rn(x); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float x) { return __float2bfloat16(x); }

// ---- packed 16-bit conversion (fast path); returns two 16-bit values in one dword, lo in bits 0..15 ----
template <typename Tag> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_tag>(float lo, float hi) {
    // Compiler-visible conversion (hipcc emits v_cvt_pk_bf16_f32 for it).  Do NOT hand-write the instruction in
    // inline asm: hipcc pads no hazards around asm, and an asm VALU that reads an MFMA accumulator inside the
    // MFMA's result latency gets stale data (seen as a rare wrong dV tile when the epilogue packed dV^T directly).
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, v);
}
template <> __device__ __forceinline__ uint32_t pack2<f16_tag>(float lo, float hi) {
    __half2 v = __floats2half2_rn(lo, hi);
    return *reinterpret_cast<uint32_t*>(&v);
}
// round-to-nearest-even variant for values that are stored as results
template <typename Tag> __device__ __forceinline__ uint32_t pack2_rn(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2_rn<bf16_tag>(float lo, float hi) { return pack2<bf16_tag>(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2_rn<f16_tag>(float lo, float hi) {
    __half2 v = __floats2half2_rn(lo, hi);
    return *reinterpret_cast<uint32_t*>(&v);
}

template <typename Tag> __device__ __forceinline__ float unpack_lo(uint32_t v);
template <typename Tag> __device__ __forceinline__ float unpack_hi(uint32_t v);
template <> __device__ __forceinline__ float unpack_lo<bf16_tag>(uint32_t v) { return __uint_as_float(v << 16); }
template <> __device__ __forceinline__ float unpack_hi<bf16_tag>(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
template <> __device__ __forceinline__ float unpack_lo<f16_tag>(uint32_t v) {
    return __half2float(__ushort_as_half((unsigned short)(v & 0xffffu)));
}
template <> __device__ __forceinline__ float unpack_hi<f16_tag>(uint32_t v) {
    return __half2float(__ushort_as_half((unsigned short)(v >> 16)));
}

// packed (p_lo * a, p_hi * b) where p is a packed 16-bit pair: dS = P * dP' with the 16-bit P that also feeds dV.
template <typename Tag> __device__ __forceinline__ uint32_t mul_pack(uint32_t p, float a, float b);
template <> __device__ __forceinline__ uint32_t mul_pack<bf16_tag>(uint32_t p, float a, float b) {
    return pack2<bf16_tag>(unpack_lo<bf16_tag>(p) * a, unpack_hi<bf16_tag>(p) * b);
}
template <> __device__ __forceinline__ uint32_t mul_pack<f16_tag>(uint32_t p, float a, float b) {
    // f32 multiply of the f16 P by the f32 dP', one rounding to f16 (v_fma_mix: two instructions per pair, no extra
    // registers).  Rounding dP' = dO V^T - delta to f16 first would overflow at |dP'| > 65504 although dS = P dP'
    // itself is small (P ~ 1/N).  Inline asm (hipcc does not select the mix forms from source): when a, b are MFMA
    // results the caller puts mfma_result_fence() on them first — hipcc pads no hazards in front of asm.
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %3, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(r) : "v"(p), "v"(a), "v"(b));
    return r;
}

// ---- MFMA wrappers: 32x32x16, 16-bit inputs, f32 accumulate ----
template <typename Tag> __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma32<bf16_tag>(s16x8 a, s16x8 b, f32x16 c) {
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf8*>(&a), *reinterpret_cast<bf8*>(&b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<f16_tag>(s16x8 a, s16x8 b, f32x16 c) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<h8*>(&a), *reinterpret_cast<h8*>(&b), c, 0, 0, 0);
}

// Same MFMA with the accumulator pinned to the ARCHITECTURAL VGPR file (inline asm, "+v").
// Why: in a kernel that needs more than 256 registers hipcc selects the AGPR form for every builtin MFMA, so
// short-lived accumulators that VALU code consumes (S, dP, dQ tiles) would compete with the resident dK/dV
// accumulators for the 256 AGPRs and bounce through v_accvgpr_read/write.  hipcc pads no hazards inside asm:
// the leading s_nop 1 covers a VALU-written operand, and mfma_result_fence() must follow the last MFMA of a
// chain before anything but a chained MFMA reads the result (8-pass XDL -> 12 wait states).
template <typename Tag> __device__ __forceinline__ void mfma32_v(s16x8 a, s16x8 b, f32x16& c);
template <> __device__ __forceinline__ void mfma32_v<bf16_tag>(s16x8 a, s16x8 b, f32x16& c) {
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma32_v<f16_tag>(s16x8 a, s16x8 b, f32x16& c) {
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_result_fence(f32x16& c) { asm volatile("s_nop 11" : "+v"(c)); }
__device__ __forceinline__ void mfma_result_fence(f32x16& c, f32x16& d) { asm volatile("s_nop 11" : "+v"(c), "+v"(d)); }

// Multiply an accumulator tile by a per-lane scalar while it stays in the accumulation registers.  For the kernels
// that run one wave per SIMD with their MFMA accumulators in AGPRs (D = 256): a plain `c *= alpha` is a VALU use, for
// which hipcc copies every accumulator to a VGPR and back on EVERY trip of the tile loop, taken or not.  The caller
// pads the MFMA -> v_accvgpr_read hazard (acc_scale_begin) because hipcc inserts no wait states around inline asm.
__device__ __forceinline__ void acc_scale_begin() { asm volatile("s_nop 15\n\ts_nop 3"); }
__device__ __forceinline__ void acc_scale_end() { asm volatile("s_nop 4"); }
__device__ __forceinline__ void acc_scale(f32x16& c, float alpha) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float x = c[i], t;
        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                     : "+a"(x), "=&v"(t) : "v"(alpha));
        c[i] = x;
    }
}

__device__ __forceinline__ float wave_half_swap(float x) {  // value held by lane ^ 32
    return __shfl_xor(x, 32, 64);
}

// ---- LDS tile images ---------------------------------------------------------------------------
// Byte offset of 16-byte chunk `ch` of row `row` inside a [rows][D] 16-bit LDS tile.  One image serves both kinds
// of MFMA operand read and is conflict-free for each:
//   * row-wise ds_read_b128 (32 lanes = 32 different rows, same chunk),
//   * transposed ds_read_b64_tr_b16 (per half wave: 4 consecutive rows x 64 B).
template <int D> struct TileSwz;
template <> struct TileSwz<128> {
    static __device__ __forceinline__ int off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
};
template <> struct TileSwz<256> {   // 512-byte rows: the bank of a chunk is its index mod 16, so the 128-wide pattern repeats
    static __device__ __forceinline__ int off(int row, int ch) { return 512 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
};
template <> struct TileSwz<64> {
    static __device__ __forceinline__ int off(int row, int ch) { return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))); }
};
typedef short lds_s16x4_t __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major
// (lane i of the group gets column i of the 4 rows; lane 4q+p supplies the address of row q, columns 4p..4p+3).
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t __attribute__((address_space(3)))*)(p));
}
// the same reads from a 32-bit LDS byte address kept in a VGPR (lane-constant operand addresses are computed once and
// advanced through the instruction's immediate offset; from generic pointers hipcc rebuilds every address separately)
__device__ __forceinline__ s16x4 lds_tr16_at(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t __attribute__((address_space(3)))*)(uintptr_t)addr);
}
__device__ __forceinline__ s16x8 lds_b128_at(unsigned addr) {
    return *(const s16x8 __attribute__((address_space(3)))*)(uintptr_t)addr;
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression in the body
template <typename F, int... Js>
__device__ __forceinline__ void for_each_const(F&& f, std::integer_sequence<int, Js...>) {
    (f(std::integral_constant<int, Js>{}), ...);
}
// Hand-timed operand reads for the phases in which a wave is alone on its SIMD's matrix pipe: the read and its wait are
// inline asm, so hipcc neither sinks the read to its use nor inserts waits of its own (it turns counted waits into
// lgkmcnt(0) around ds_read_b64_tr_b16 pairs, which drains the prefetch every few MFMAs).  OFF = immediate byte offset.
template <int OFF> __device__ __forceinline__ s16x4 lds_tr16_asm(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ s16x8 lds_b128_asm(unsigned addr) {
    s16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// wait until at most N LDS operations issued after the ones that fill `x` are outstanding; ties x to the wait
template <int N> __device__ __forceinline__ void lds_wait_for(s16x8& x) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N));
}
// wait + MFMA as ONE asm statement (no compiler-inserted s_nop between them, one issue slot less per step).  The
// caller owns the hazards: the last MFMA of a stream is followed by mfma_stream_fence() before anything but a chained
// MFMA touches the accumulators (hipcc pads nothing around inline asm; tools/mfma_hazard_audit.py checks the .so).
// `first` writes its destination before it has read every source: early clobber, as LLVM marks these MFMAs.
#define FA_MFMA_WAIT_IMPL(TAG, OPC)                                                                                     \
    template <int N> struct MfmaWait_##TAG {                                                                            \
        static __device__ __forceinline__ void acc(s16x8 a, s16x8 b, f32x16& c) {                                       \
            asm volatile("s_waitcnt lgkmcnt(%3)\n\t" OPC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b), "n"(N));           \
        }                                                                                                               \
        static __device__ __forceinline__ void first(s16x8 a, s16x8 b, f32x16& c) {                                     \
            asm volatile("s_waitcnt lgkmcnt(%3)\n\t" OPC " %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b), "n"(N));            \
        }                                                                                                               \
    };
FA_MFMA_WAIT_IMPL(bf16, "v_mfma_f32_32x32x16_bf16")
FA_MFMA_WAIT_IMPL(f16, "v_mfma_f32_32x32x16_f16")
// the same MFMAs without a wait (their operand's arrival is covered by an earlier counted wait)
#define FA_MFMA_NOWAIT_IMPL(TAG, OPC)                                                                                   \
    struct MfmaNoWait_##TAG {                                                                                           \
        static __device__ __forceinline__ void acc(s16x8 a, s16x8 b, f32x16& c) {                                       \
            asm volatile(OPC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));                                             \
        }                                                                                                               \
        static __device__ __forceinline__ void first(s16x8 a, s16x8 b, f32x16& c) {                                     \
            asm volatile(OPC " %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));                                              \
        }                                                                                                               \
    };
FA_MFMA_NOWAIT_IMPL(bf16, "v_mfma_f32_32x32x16_bf16")
FA_MFMA_NOWAIT_IMPL(f16, "v_mfma_f32_32x32x16_f16")
template <typename Tag> struct MfmaNoWait;
template <> struct MfmaNoWait<bf16_tag> : MfmaNoWait_bf16 {};
template <> struct MfmaNoWait<f16_tag> : MfmaNoWait_f16 {};
template <typename Tag, int N> struct MfmaWait;
template <int N> struct MfmaWait<bf16_tag, N> : MfmaWait_bf16<N> {};
template <int N> struct MfmaWait<f16_tag, N> : MfmaWait_f16<N> {};
// 8-pass XDL result -> VALU read: 12 wait states after the last MFMA of an inline-asm stream (one s_nop, tied to every
// accumulator the stream wrote so that no reader is scheduled above it)
template <int NA, int NB> __device__ __forceinline__ void mfma_stream_fence(f32x16 (&a)[NA], f32x16 (&b)[NB]) {
    static_assert(NA == 4 && (NB == 4 || NB == 2), "operand list is written out for KB = 4 and NDV = 4 | 2");
    if constexpr (NB == 4)
        asm volatile("s_nop 11" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else
        asm volatile("s_nop 11" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
}
template <int NB> __device__ __forceinline__ void mfma_stream_fence(f32x16 (&a)[2], f32x16 (&b)[NB]) {
    if constexpr (NB == 4) asm volatile("s_nop 11" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else asm volatile("s_nop 11" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
}
__device__ __forceinline__ s16x8 cat8(s16x4 lo, s16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// ---- LDS-DMA staging of a swizzled tile -----------------------------------------------------------
// buffer_load_dwordx4 ... lds writes LDS at (wave-uniform base) + 16 * lane, so the LDS image stays lane-linear
// and the XOR swizzle is applied to the per-lane SOURCE address instead: slot s of a row holds chunk s ^ f(row),
// and f is its own inverse.  One wave-instruction moves 1 KiB = RPP = 512 / D rows.  Wave w of NW issues pieces
// w, w + NW, ...; RPP * NW is a multiple of 16 rows, so f(row) is the same for every piece of a wave and the
// per-lane offset is computed once.  Rows past the tensor's end read as zero through the buffer range check.
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// 16 bytes at byte offset `off` of the tensor behind rsrc; zeros when the range check fails (rows past the end)
__device__ __forceinline__ s16x8 buf_load_frag(buf_rsrc_t rsrc, int off) {
    u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    return *reinterpret_cast<s16x8*>(&t);
}
// Padded head dims: a tensor row holds `dr` <= D elements (dr a multiple of 8, so 16-byte chunks are either whole or
// absent); tiles and MFMA shapes stay D wide and the missing chunks are requested at kOobOff, an offset the buffer
// range check rejects, so they arrive as zeros like the rows past the tensor's end.  dr == D folds to the plain form.
constexpr int kOobOff = (int)0x80000000u;
__device__ __forceinline__ int frag_off(int row, int col, int dr, bool pad) {
    return (pad && col >= dr) ? kOobOff : (row * dr + col) * 2;
}
template <int D>
__device__ __forceinline__ int dma_lane_voff(int lane, int w, int dr = D) {
    constexpr int RPP = 512 / D, CPR = D / 8;
    const int rl = lane / CPR, slot = lane - rl * CPR;
    const int row = (RPP * w + rl) & 15;
    const int ch = (TileSwz<D>::off(row, slot) - 2 * D * row) >> 4;
    return (8 * ch < dr) ? rl * 2 * dr + 16 * ch : kOobOff;
}
// The DMA itself is issued from inline asm, so hipcc does not see it: with the builtin form hipcc treats every later
// ds_read_b64_tr_b16 as possibly aliasing the pending LDS write and parks the wave on s_waitcnt vmcnt(0) in the
// middle of the tile (the DMA's whole latency exposed).  The price: nothing waits for the data unless we do —
// dma_wait_all() (s_waitcnt vmcnt(0)) must precede the barrier after which other waves read the tile.
// M0 carries the LDS base for the instruction.  hipcc reserves M0 and keeps nothing in it across statements, so it is
// written here and not restored.  The s_mov + s_nop 3 are the five wait states a buffer instruction needs after a
// v_readfirstlane wrote one of its SGPR operands (the soffset), which also covers the one state M0 needs.
typedef u32x4 rsrc_s_t;   // buffer descriptor held in SGPRs (every word wave-uniform)
__device__ __forceinline__ rsrc_s_t make_rsrc_s(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    rsrc_s_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
// LDS byte address of a pointer into the workgroup's shared memory: the low half of its flat address (the high half
// is the shared aperture).  Deliberately not a flat -> local address-space cast: its null check (high half != 0),
// once the high half folds to the aperture register, is mis-selected by this hipcc ("Operand has incorrect register
// class: V_CMP_NE_U32 0, src_shared_base").
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
}
__device__ __forceinline__ void dma16_issue(rsrc_s_t rsrc, unsigned lds_dst, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                 :: "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4_issue(rsrc_s_t rsrc, unsigned lds_dst, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 3\n\tbuffer_load_dword %0, %2, %3 offen lds"
                 :: "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// stage rows [row0, row0 + ROWS) of the tensor behind `rsrc` (row stride 2*dr bytes) into the LDS tile at `tile`
// (D = 256 with 4 waves: a wave's pieces alternate between two row classes mod 16, RPP * NW = 8; `voff_b` =
// dma_lane_voff<D>(lane, w + NW, dr) serves the odd ones.)
template <int D, int ROWS, int NW>
__device__ __forceinline__ void dma_stage_tile(rsrc_s_t rsrc, char* tile, int row0, int voff, int w, int dr = D,
                                               int voff_b = 0) {
    constexpr int RPP = 512 / D, PIECES = ROWS / RPP, PER_WAVE = PIECES / NW;
    static_assert(PIECES % NW == 0 && ((RPP * NW) % 16 == 0 || RPP * NW == 8), "tile does not split evenly over the waves");
    const unsigned t0 = lds_addr_of(tile);
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int pc = w + NW * j;
        const int vo = ((RPP * NW) % 16 != 0 && (j & 1)) ? voff_b : voff;
        dma16_issue(rsrc, t0 + pc * 1024, vo, __builtin_amdgcn_readfirstlane((row0 + RPP * pc) * 2 * dr));
    }
}

// ---- epilogue: a wave's 32 x D result tile, held "row on the lane" (lane (r, h) owns 4-element pieces at columns
// 32 b + 8 g + 4 h of row r), goes through a wave-private LDS region and leaves as whole rows: 1 KiB contiguous per
// store instruction instead of 64 scattered 16-byte pieces (the per-workgroup tail is store-issue bound).
// `lrow`: the row of the 32-row tile this lane's values belong to (default: lane & 31; the dQ product kernel's lanes hold their
// rows in the order of the kernel that wrote its operand tiles).
template <int D>
__device__ __forceinline__ void store_rows_via_lds(char* wl, const u32x2 (&vals)[(D / 32) * 4], uint16_t* gdst, int row0,
                                                   int n, int lane, int dr = D, int lrow = -1) {
    constexpr int CPR = D / 8, RPI = 512 / D, ROWB = 2 * D;   // chunks per row, rows per 1-KiB store, row bytes
    const int r = lrow < 0 ? (lane & 31) : lrow, h = lane >> 5;
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<u32x2*>(wl + r * ROWB + 16 * ((4 * b + g) ^ (r & (CPR - 1) & 15)) + 8 * h) = vals[4 * b + g];
    const int rl = lane / CPR, cc = lane - rl * CPR;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int row = RPI * i + rl;
        const u32x4 v = *reinterpret_cast<const u32x4*>(wl + row * ROWB + 16 * (cc ^ (row & (CPR - 1) & 15)));
        if (row0 + row < n && 8 * cc < dr) *reinterpret_cast<u32x4*>(gdst + (size_t)(row0 + row) * dr + 8 * cc) = v;
    }
}

// XCD-aware block remap: blocks that share blockIdx % 8 share an XCD (and its L2) under the
// observed round-robin placement (speed only, never correctness).  Returns the logical id such that
// consecutive logical ids [c*per, (c+1)*per) run on one XCD.  Bijective for any nblk.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace fa
