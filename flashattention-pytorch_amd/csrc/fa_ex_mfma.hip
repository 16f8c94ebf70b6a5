// Extended attention on the 16-bit MFMA (bf16 / f16 tensors, head_dim a multiple of 8 up to 128): the notebook model's
// extras (fa_ex.hip has the list and the reference lines) on the decomposition of the plain kernels —
//   forward   (fa_fwd_mfma.hip)      : 8 waves x 32 query rows, K/V tiles of 128 keys by LDS-DMA, S^T = K Q^T with the query on
//                                      the lane, online softmax in registers, O^T += V^T P^T;
//   dK/dV     (fa_bwd_dkdv_mfma.hip) : 8 waves x 32 keys, Q/dO tiles of 64 rows, the key on the lane;
//   dQ        (fa_bwd_dq_mfma.hip)   : 8 waves x 32 query rows, K/V tiles of 64 keys, the query on the lane;
// with Nq != Nk (separate row counts, causal diagonal shifted by Nk - Nq), and per element
//   FEAT bit 0  dense mask and / or block-sparse mask (either pointer may be null at run time),
//   FEAT bit 1  dropout (counter-based, fa_ex_common.h: a lane makes one splitmix64 value per two of its elements).
// Dense mask bytes are fetched with range-checked buffer loads (rows / bytes past the mask read as 0 = masked); when Nk, the
// mask pointer and the (b,h) stride are multiples of 4 a lane of the query-on-the-lane kernels takes the 4 keys of a
// register group with one dword load.  The block-sparse mask needs br, bc multiples of 32 here (a wave's 32 x 32 block
// then has ONE entry; other block shapes take the exact-f32 kernels): tiles without a live entry are skipped before
// they are loaded, live tiles mask their dead 32 x 32 blocks.  A row without a visible key: o = 0, lse = -inf, dQ = 0.
#include "fa_common.h"
#include "fa_ex_common.h"
#include "fa_kernels.h"

namespace fa {

namespace {

constexpr int kFeatMask = 1, kFeatDrop = 2;

// rc(i): row (or key) offset inside a 32-wide block of accumulator register i, before the 4 * (lane >> 5) term
__device__ __forceinline__ constexpr int rc_of(int i) { return (i & 3) + 8 * (i >> 2); }

// Which tiles does the block-sparse mask leave alive?  A workgroup walks tiles along ONE axis (key tiles of T keys in
// the forward and dQ kernels, query tiles of T rows in the dK/dV kernel) against a fixed range of the other axis.  One
// probe = one byte load per lane + a ballot, and covers TP = 64 / E consecutive tiles, E = block-mask entries a tile can
// touch (at most 32: tiles are at most 256 x 128, blocks at least 32 x 32); the window is kept in scalar registers, so
// next() costs a probe only every TP tiles (128 x 128 blocks: every 32 key tiles).  All results are wave-uniform.
template <bool KEYS_VARY, int T>
struct LiveScan {
    const uint8_t* bm;
    int nbc, blk_var, blk_fix;     // block size along the walked / the fixed axis
    int fb0, nfix;                 // first block and number of blocks of the fixed range
    int nvar, E, TP;               // blocks of the walked axis a tile can touch; entries per tile; tiles per probe
    int origin, alim, ntiles;      // tile t covers [origin + t T, origin + (t + 1) T) clipped to alim
    int base;                      // first tile of the cached window (-1: none)
    unsigned long long bits;
    int lane;

    __device__ __forceinline__ void init(const ExParams& p, int f0, int f1, int origin_, int alim_, int ntiles_, int lane_) {
        bm = p.bmask; nbc = p.nbc;
        blk_var = KEYS_VARY ? p.bc : p.br;
        blk_fix = KEYS_VARY ? p.br : p.bc;
        fb0 = f0 / blk_fix;
        nfix = (f1 - 1) / blk_fix - fb0 + 1;
        nvar = (T % blk_var == 0) ? T / blk_var : ((blk_var % T == 0 && origin_ % T == 0) ? 1 : (T - 1) / blk_var + 2);
        E = nfix * nvar;
        TP = 64 / E;
        origin = origin_; alim = alim_; ntiles = ntiles_; lane = lane_;
        base = -1; bits = 0;
    }
    __device__ __forceinline__ void probe(int t0) {
        const int tau = lane / E, e = lane - tau * E;
        const int iv = e / nfix, jf = e - iv * nfix;
        const int tile = t0 + tau;
        bool hit = false;
        if (tau < TP && tile < ntiles) {
            const int a0 = origin + tile * T;
            const int vb0 = a0 / blk_var, vb1 = (min(a0 + T, alim) - 1) / blk_var;
            if (vb0 + iv <= vb1) hit = (KEYS_VARY ? bm[(fb0 + jf) * nbc + vb0 + iv] : bm[(vb0 + iv) * nbc + fb0 + jf]) != 0;
        }
        bits = __ballot(hit);
        base = t0;
    }
    // first live tile >= t (ntiles if none)
    __device__ __forceinline__ int next(int t) {
        const unsigned long long emask = E >= 64 ? ~0ull : ((1ull << E) - 1ull);
        while (t < ntiles) {
            if (base < 0 || t < base || t >= base + TP) probe(t);
            if ((bits >> ((t - base) * E)) & emask) return t;
            ++t;
        }
        return t;
    }
};

struct MaskSrc {
    buf_rsrc_t rs;
    bool on, dwords, quads;
};
__device__ __forceinline__ MaskSrc make_mask_src(const ExParams& p, int bh) {
    MaskSrc m;
    m.on = p.mask != nullptr;
    const uint8_t* base = m.on ? p.mask + (size_t)bh * p.mask_bh : nullptr;
    m.rs = make_rsrc(base, m.on ? (unsigned)p.nq * (unsigned)p.nk : 0u);
    m.dwords = m.on && (p.nk & 3) == 0 && (p.mask_bh & 3) == 0 && (((uintptr_t)p.mask) & 3) == 0;
    m.quads = m.on && (p.nk & 15) == 0 && (p.mask_bh & 15) == 0 && (((uintptr_t)p.mask) & 15) == 0;
    return m;
}
__device__ __forceinline__ unsigned mask_load_b32(const MaskSrc& m, int off) { return __builtin_amdgcn_raw_buffer_load_b32(m.rs, off, 0, 0); }
__device__ __forceinline__ unsigned mask_load_b8(const MaskSrc& m, int off) { return (unsigned)(__builtin_amdgcn_raw_buffer_load_b8(m.rs, off, 0, 0) & 0xff); }

// 16-bit visibility mask (bit i = register i visible) of one 32 x 32 block for a lane of a query-on-the-lane kernel:
// the lane's row is `row`, register i holds key kcol + rc(i)
__device__ __forceinline__ void mask_words_q(const MaskSrc& m, int row, int nk, int kcol, unsigned (&wd)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) wd[g] = mask_load_b32(m, row * nk + kcol + 8 * g);
}
// The same words from ONE 16-byte load per lane (round 3; Nk, pointer and (b,h) stride multiples of 16): the two lanes of a row
// (r, h = 0 / 1) fetch bytes 16 h .. 16 h + 15 of the row's 32 and trade the dwords the other one's registers stand for — lane
// (r, 0) holds keys 0-3, 8-11, 16-19, 24-27 of the block, lane (r, 1) keys 4-7, 12-15, 20-23, 28-31 — with two
// v_permlane32_swap (upper half of the first operand <-> lower half of the second).  A quarter of the load instructions, the
// same cache lines: the address unit walks 64 lanes per instruction either way.
__device__ __forceinline__ void mask_words_q16(const MaskSrc& m, int row, int nk, int kblk, int h, unsigned (&wd)[4]) {
    const u32x4 w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(m.rs, row * nk + kblk + 16 * h, 0, 0));
    const auto s01 = __builtin_amdgcn_permlane32_swap(w[0], w[1], false, false);   // [0]: lo own w0 | hi <- lo's w1;  [1]: lo <- hi's w0 | hi own w1
    const auto s23 = __builtin_amdgcn_permlane32_swap(w[2], w[3], false, false);
    wd[0] = s01[0]; wd[1] = s23[0]; wd[2] = s01[1]; wd[3] = s23[1];
}
__device__ __forceinline__ unsigned bits_of_words(const unsigned (&wd)[4]) {
    // Structured masks (a causal or windowed mask handed over as a dense one) are all-visible or all-masked over most
    // 32 x 32 blocks: two wave-uniform tests on the words save the per-byte work there (6 instructions per element)
    unsigned zero_byte = 0;   // bit 7 of every byte that is 0
#pragma unroll
    for (int g = 0; g < 4; ++g) zero_byte |= (wd[g] - 0x01010101u) & ~wd[g] & 0x80808080u;
    if (__all(zero_byte == 0)) return 0xffffu;
    if (__all((wd[0] | wd[1] | wd[2] | wd[3]) == 0)) return 0u;
    // bit 4 g + j = byte j of word g is non-zero.  Branch-free on the whole word (round 3; was a shift, mask, compare and select
    // per byte): bit 7 of every non-zero byte, then the four bits gathered with three shifts — 11 instructions per four
    // elements instead of 16.
    unsigned bits = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const unsigned nz = (((wd[g] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | wd[g]) & 0x80808080u;
        bits |= (((nz >> 7) | (nz >> 14) | (nz >> 21) | (nz >> 28)) & 0xfu) << (4 * g);
    }
    return bits;
}
// s where bit i of `vis` is set, -inf where it is not: a sign-extended one-bit field as the select mask of a bit-field insert
// (two instructions, no compare / VCC round trip)
__device__ __forceinline__ float keep_or_minus_inf(float s, unsigned vis, int i) {
    const int m = ((int)(vis << (31 - i))) >> 31;                     // v_bfe_i32: 0 or -1
    return __uint_as_float((__float_as_uint(s) & (unsigned)m) | (0xff800000u & ~(unsigned)m));   // v_bfi_b32
}
__device__ __forceinline__ unsigned dense_bits_q(const MaskSrc& m, int row, int nk, int kcol /* block's first key + 4 h */, int h) {
    if (m.quads) {
        unsigned wd[4];
        mask_words_q16(m, row, nk, kcol - 4 * h, h, wd);
        return bits_of_words(wd);
    }
    if (m.dwords) {
        unsigned wd[4];
        mask_words_q(m, row, nk, kcol, wd);
        return bits_of_words(wd);
    }
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) bits |= mask_load_b8(m, row * nk + kcol + rc_of(i)) ? (1u << i) : 0u;
    return bits;
}
// the same for a lane of the key-on-the-lane kernel: the lane's key is `key`, register i holds row rb0 + 4 h + rc(i)
__device__ __forceinline__ unsigned dense_bits_k(const MaskSrc& m, int rrow, int nk, int key) {
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) bits |= mask_load_b8(m, (rrow + rc_of(i)) * nk + key) ? (1u << i) : 0u;
    return bits;
}

// The same through LDS (round 3; Nk, pointer and (b,h) stride multiples of 16): the block's 32 x 32 mask bytes are exactly one
// 16-byte load per lane (lane l: row l >> 1, bytes 16 (l & 1) ..), written to a 1 KiB image [row][32 bytes] of the wave's own,
// from which the lane reads its key's column — 16 byte reads from LDS instead of 16 byte loads from memory.  Structured masks
// leave on the words as loaded: all bytes set / all bytes clear over the wave are decided before the round trip.  A wave's
// LDS instructions execute in order, so the reads see the wave's own writes without a barrier.
__device__ __forceinline__ unsigned dense_bits_k_lds(const MaskSrc& m, int rb0, int nk, int kw0, int lane, char* img /* this wave's 1 KiB */) {
    asm volatile("" : "+v"(lane));   // the lane-constant addresses below are made per call: hoisted, they are three registers this kernel does not have
    const u32x4 w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(m.rs, (rb0 + (lane >> 1)) * nk + kw0 + 16 * (lane & 1), 0, 0));
    unsigned zero_byte = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) zero_byte |= (w[g] - 0x01010101u) & ~w[g] & 0x80808080u;
    if (__all(zero_byte == 0)) return 0xffffu;
    if (__all((w[0] | w[1] | w[2] | w[3]) == 0)) return 0u;
    *reinterpret_cast<u32x4*>(img + 16 * lane) = w;
    const int c = lane & 31, h = lane >> 5;
    unsigned bits = 0;
    const char* col = img + 128 * h + c;   // row 4 h + rc(i), rc(i) = (i & 3) + 8 (i >> 2): immediate offsets from one address
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bits |= (*reinterpret_cast<const volatile uint8_t*>(col + 32 * (j + 8 * g)) ? 1u : 0u) << (4 * g + j);
        asm volatile("" : "+v"(bits));   // four reads in flight at a time: the kernel has no registers to spare
    }
    return bits;
}

// keep bits (bit i = register i kept) of one 32 x 32 block.  Query on the lane: row fixed, keys kcol + rc(i) — registers
// 4g+0, 4g+1 are one key pair, 4g+2, 4g+3 the next: 8 values per block.
__device__ __forceinline__ unsigned keep_bits_q(const ExParams& p, unsigned hi, int row, int kcol) {
    unsigned bits = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const unsigned long long z = ex_hash(hi, (unsigned)(kcol + 8 * g + 2 * pr) >> 1, p.seedmix);
            const unsigned wd = (row & 1) ? (unsigned)(z >> 32) : (unsigned)z;
            bits |= ((wd & 0xffffu) >= p.drop_thr) ? (1u << (4 * g + 2 * pr)) : 0u;
            bits |= ((wd >> 16) >= p.drop_thr) ? (1u << (4 * g + 2 * pr + 1)) : 0u;
        }
    return bits;
}
// Key on the lane: key fixed, rows rrow + rc(i) — registers 4g+0, 4g+1 are one row pair
__device__ __forceinline__ unsigned keep_bits_k(const ExParams& p, unsigned hi_bh, int rrow, int key) {
    unsigned bits = 0;
    const int sh = 16 * (key & 1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const unsigned long long z = ex_hash(hi_bh + ((unsigned)(rrow + 8 * g + 2 * pr) >> 1), (unsigned)key >> 1, p.seedmix);
            bits |= ((((unsigned)z >> sh) & 0xffffu) >= p.drop_thr) ? (1u << (4 * g + 2 * pr)) : 0u;
            bits |= ((((unsigned)(z >> 32) >> sh) & 0xffffu) >= p.drop_thr) ? (1u << (4 * g + 2 * pr + 1)) : 0u;
        }
    return bits;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ forward
template <typename Tag, int D, int FEAT>
__global__ __launch_bounds__(512, 2) void exm_fwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                         const uint16_t* __restrict__ v, uint16_t* __restrict__ o,
                                                         float* __restrict__ lse, ExParams p, float c_log2) {
    constexpr int NW = 8, BM = 32 * NW, KB = 4, BN = 32 * KB, NKS = D / 16, NDV = D / 32, TILE_BYTES = BN * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K tile | V tile]
    const int DR = p.d, nq = p.nq, nk = p.nk;
    const int nqt = (nq + BM - 1) / BM;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nqt;
    const int q0 = (L - bh * nqt) * BM;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t qbase = (size_t)bh * nq * DR, kbase = (size_t)bh * nk * DR;
    const int qrow = q0 + 32 * w + r;

    const buf_rsrc_t q_rs = make_rsrc(q + qbase, (unsigned)nq * DR * 2);
    s16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = buf_load_frag(q_rs, frag_off(qrow, 16 * ks + 8 * h, DR, true));

    const rsrc_s_t k_rs = make_rsrc_s(k + kbase, (unsigned)nk * DR * 2);
    const rsrc_s_t v_rs = make_rsrc_s(v + kbase, (unsigned)nk * DR * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, DR);
    auto stage = [&](int buf, int k0) {
        char* kb_ = smem + buf * 2 * TILE_BYTES;
        dma_stage_tile<D, BN, NW>(k_rs, kb_, k0, dma_voff, w, DR);
        dma_stage_tile<D, BN, NW>(v_rs, kb_ + TILE_BYTES, k0, dma_voff, w, DR);
    };
    const MaskSrc msk = make_mask_src(p, bh);
    const bool use_bm = (FEAT & kFeatMask) && p.bmask != nullptr;
    const bool drop = (FEAT & kFeatDrop) && p.p_drop > 0.f;
    const unsigned hi = (unsigned)bh * p.nqh + ((unsigned)qrow >> 1);
    const int rbw = min(q0 + 32 * w, nq - 1) / p.br;   // block row of this wave's 32 rows (br is a multiple of 32)

    f32x16 oacc[NDV];
#pragma unroll
    for (int t = 0; t < NDV; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // keys past the last row's diagonal are masked for every row of the tile (of the wave)
    const int kend = p.causal ? max(0, min(nk, q0 + BM + p.coff)) : nk;
    const int kend_w = p.causal ? max(0, min(nk, q0 + 32 * w + 32 + p.coff)) : nk;
    const int ntiles = (kend + BN - 1) / BN, ntiles_w = (kend_w + BN - 1) / BN;
    LiveScan<true, BN> scan;
    if (use_bm) scan.init(p, q0, min(q0 + BM, nq), 0, nk, ntiles, lane);
    auto next_live = [&](int t) { return use_bm ? scan.next(t) : t; };
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;

    int t = next_live(0), cur = 0;
    if (t < ntiles) stage(0, t * BN);
    dma_wait_all();
    __syncthreads();
    // two loops instead of an `if` inside one (a conditional accumulate makes hipcc carry the accumulators through
    // copies): tiles this wave computes, then the ones it only helps to load
    // Dense mask: its loads are ordinary VMEM loads, and VMEM returns in order — were the next tile's LDS-DMA issued
    // first, the wait for the mask words would also be a wait for that whole tile.  So the DMA goes out when the mask has
    // been read (it still has the tile's products to land).  Fetching the words a tile ahead instead (16 more live
    // registers) was measured slower: 1.15 vs 1.08 ms forward at BH 64, N 4096, half the pairs masked.
    const bool late_stage = (FEAT & kFeatMask) && msk.on;
    while (t < ntiles_w) {
        const int tn = next_live(t + 1);
        if (!late_stage && tn < ntiles) stage(cur ^ 1, tn * BN);
        const int k0 = t * BN;
        const char* Kt = smem + cur * 2 * TILE_BYTES;
        const char* Vt = Kt + TILE_BYTES;
        {
            // visibility / keep bits of this lane's 4 x 16 elements, requested ahead of the S MFMAs
            unsigned vis[KB], kp[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                vis[kb] = 0xffffu;
                kp[kb] = 0xffffu;
                if constexpr (FEAT & kFeatMask) {
                    if (msk.on) vis[kb] = dense_bits_q(msk, qrow, nk, k0 + 32 * kb + 4 * h, h);
                    if (use_bm && p.bmask[rbw * p.nbc + min(k0 + 32 * kb, nk - 1) / p.bc] == 0) vis[kb] = 0;
                }
                if constexpr (FEAT & kFeatDrop) {
                    if (drop) kp[kb] = keep_bits_q(p, hi, qrow, k0 + 32 * kb + 4 * h);
                }
            }
            // a tile of which this wave sees nothing (the upper triangle of a causal mask handed over as a dense one, the
            // dead blocks of a block-sparse tile) is not computed: wave-uniform
            bool any_vis = true;
            if constexpr (FEAT & kFeatMask) {
                unsigned all = 0;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) all |= vis[kb];
                any_vis = __any(all != 0) != 0;
                if (late_stage && tn < ntiles) stage(cur ^ 1, tn * BN);   // the mask words have arrived
            }
            if (any_vis) {
            f32x16 sacc[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const s16x8 a = *reinterpret_cast<const s16x8*>(Kt + TileSwz<D>::off(32 * kb + r, 2 * ks + h));
                    sacc[kb] = mfma32<Tag>(a, qf[ks], sacc[kb]);
                }
            }
            // ---- causal diagonal / ragged last tile: key index of register i is k0 + 32 kb + 4 h + rc(i)
            const bool need_mask = (p.causal && (k0 + BN - 1 > q0 + 32 * w + p.coff)) || (k0 + BN > nk);
            if (need_mask) {
                const int lim = p.causal ? min(qrow + p.coff, nk - 1) : nk - 1;   // last visible key of this lane's row
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int thr = lim - (k0 + 32 * kb + 4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (rc_of(i) > thr) sacc[kb][i] = -INFINITY;
                }
            }
            if constexpr (FEAT & kFeatMask) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    if (__any(vis[kb] != 0xffffu)) {   // wave-uniform
#pragma unroll
                        for (int i = 0; i < 16; ++i) sacc[kb][i] = keep_or_minus_inf(sacc[kb][i], vis[kb], i);
                    }
            }
            // ---- online softmax (fa_fwd_mfma.hip), with rows that have not met a visible key yet (m = -inf)
            float mx = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kb][i]);
            mx = fmaxf(mx, wave_half_swap(mx));
            const float m_new = fmaxf(m_run, mx);
            float mc;
            // lazy rescale: keep the stale max while no row has grown past it by more than 2^8; -inf - -inf = NaN counts
            // as "rescale", so a wave with a dead row takes the exact path
            const bool rescale = __any(!((m_new - m_run) * c_log2 <= 8.0f)) != 0;
            if (rescale) {
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * c_log2);
                mc = m_use * c_log2;
                m_run = m_new;
#pragma unroll
                for (int t2 = 0; t2 < NDV; ++t2)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[t2][i] *= alpha;
                l_run *= alpha;
            } else {
                mc = m_run * c_log2;
            }
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float pe = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], c_log2, -mc));
                    rs += pe;   // the denominator counts every visible key, dropped or not
                    if constexpr (FEAT & kFeatDrop) pe = ((kp[kb] >> i) & 1u) ? pe * p.keep_scale : 0.f;
                    sacc[kb][i] = pe;
                }
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
                        const s16x4 lo = lds_tr16(Vt + TileSwz<D>::off(key_a, ch) + 8 * (tp & 1));
                        const s16x4 hi4 = lds_tr16(Vt + TileSwz<D>::off(key_a + 8, ch) + 8 * (tp & 1));
                        oacc[dvb] = mfma32<Tag>(cat8(lo, hi4), pb, oacc[dvb]);
                    }
                }
            }
            l_run += rs;
            }   // any_vis
        }
        dma_wait_all();
        __syncthreads();
        cur ^= 1;
        t = tn;
    }
    while (t < ntiles) {
        const int tn = next_live(t + 1);
        if (tn < ntiles) stage(cur ^ 1, tn * BN);
        dma_wait_all();
        __syncthreads();
        cur ^= 1;
        t = tn;
    }

    // ---- epilogue: normalise, store O and lse.  Every wave is past the last barrier and nothing is in flight: each wave
    // stages its rows in 32 x D x 2 bytes of buffer 0
    const float l_tot = l_run + wave_half_swap(l_run);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    u32x2 vals[NDV * 4];
#pragma unroll
    for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            vals[4 * dvb + g][0] = pack2_rn<Tag>(oacc[dvb][4 * g + 0] * inv, oacc[dvb][4 * g + 1] * inv);
            vals[4 * dvb + g][1] = pack2_rn<Tag>(oacc[dvb][4 * g + 2] * inv, oacc[dvb][4 * g + 3] * inv);
        }
    store_rows_via_lds<D>(smem + w * 32 * D * 2, vals, o + qbase, q0 + 32 * w, nq, lane, DR);
    if (qrow < nq && h == 0) lse[(size_t)bh * nq + qrow] = l_tot > 0.f ? m_run * p.scale + logf(l_tot) : -INFINITY;
}

// ------------------------------------------------------------------------------------------------ row constants
// nlse = -lse / scale (0 for a row without a visible key: every element of such a row is masked by a select, and an
// infinite initial accumulator would only make NaNs on the way), ndelta = -rowsum(dO * O).  16 lanes per row.
template <typename Tag>
__global__ __launch_bounds__(256) void exm_prep_kernel(const uint16_t* __restrict__ o, const uint16_t* __restrict__ dout,
                                                       const float* __restrict__ lse, float* __restrict__ nlse,
                                                       float* __restrict__ ndelta, long long rows, int d, float inv_scale) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    float s = 0.f;
    if (row < rows && 8 * sub < d) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(o + row * d + 8 * sub);
        const u32x4 b = *reinterpret_cast<const u32x4*>(dout + row * d + 8 * sub);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += unpack_lo<Tag>(a[j]) * unpack_lo<Tag>(b[j]) + unpack_hi<Tag>(a[j]) * unpack_hi<Tag>(b[j]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (row < rows && sub == 0) {
        const float l = lse[row];
        nlse[row] = (l == -INFINITY) ? 0.f : -l * inv_scale;
        ndelta[row] = -s;
    }
}

// ------------------------------------------------------------------------------------------------ dK / dV
// M16: the dense mask (if any) is 16-byte aligned in every respect — its bytes come through the wave's LDS image
// (dense_bits_k_lds); a separate instantiation, not a run-time choice: with both loaders in one kernel the masked form spills
template <typename Tag, int D, int FEAT, bool M16 = false>
__global__ __launch_bounds__(512, 2) void exm_dkdv_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                          const uint16_t* __restrict__ v, const uint16_t* __restrict__ dout,
                                                          const float* __restrict__ nlse, const float* __restrict__ ndelta,
                                                          uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, ExParams p,
                                                          float c_log2) {
    constexpr int NW = 8, BK = 32 * NW, BQ = 64, NKS = D / 16, NDB = D / 32;
    constexpr int K_BYTES = BK * D * 2, Q_BYTES = BQ * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                      // [256][D]
    char* Qs = Ks + K_BYTES;              // [2][64][D]
    char* Os = Qs + 2 * Q_BYTES;          // [2][64][D]   (dO)
    float* Ls = reinterpret_cast<float*>(Os + 2 * Q_BYTES);  // [2][ 64 x -lse/scale | 64 x -delta ]
    const int DR = p.d, nq = p.nq, nk = p.nk;
    const int nkt = (nk + BK - 1) / BK;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nkt;
    const int key0 = (L - bh * nkt) * BK;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t qbase = (size_t)bh * nq * DR, kbase = (size_t)bh * nk * DR, rbase = (size_t)bh * nq;
    const int kw0 = key0 + 32 * w, key = kw0 + r;

    const rsrc_s_t k_rs = make_rsrc_s(k + kbase, (unsigned)nk * DR * 2);
    const rsrc_s_t q_rs = make_rsrc_s(q + qbase, (unsigned)nq * DR * 2);
    const rsrc_s_t o_rs = make_rsrc_s(dout + qbase, (unsigned)nq * DR * 2);
    const rsrc_s_t l_rs = make_rsrc_s(nlse + rbase, (unsigned)nq * 4);
    const rsrc_s_t d_rs = make_rsrc_s(ndelta + rbase, (unsigned)nq * 4);
    const buf_rsrc_t v_rs = make_rsrc(v + kbase, (unsigned)nk * DR * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, DR);
    auto stage = [&](int buf, int qs) {
        dma_stage_tile<D, BQ, NW>(q_rs, Qs + buf * Q_BYTES, qs, dma_voff, w, DR);
        dma_stage_tile<D, BQ, NW>(o_rs, Os + buf * Q_BYTES, qs, dma_voff, w, DR);
        // row constants: 64 floats each, one 4-byte LDS-DMA per lane (rows >= nq read as 0: harmless, their dO is 0)
        if (w == 0) dma4_issue(l_rs, lds_addr_of(Ls + buf * 128), lane * 4, __builtin_amdgcn_readfirstlane(qs * 4));
        if (w == 1) dma4_issue(d_rs, lds_addr_of(Ls + buf * 128 + 64), lane * 4, __builtin_amdgcn_readfirstlane(qs * 4));
    };
    const MaskSrc msk = make_mask_src(p, bh);
    const bool use_bm = (FEAT & kFeatMask) && p.bmask != nullptr;
    const bool drop = (FEAT & kFeatDrop) && p.p_drop > 0.f;
    const int cbw = min(kw0, nk - 1) / p.bc;   // block column of this wave's 32 keys (bc is a multiple of 32)

    dma_stage_tile<D, BK, NW>(k_rs, Ks, key0, dma_voff, w, DR);
    s16x8 vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) vf[ks] = buf_load_frag(v_rs, frag_off(key, 16 * ks + 8 * h, DR, true));

    f32x16 dka[NDB], dva[NDB];
#pragma unroll
    for (int t = 0; t < NDB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dka[t][i] = 0.f; dva[t][i] = 0.f; }

    // key j is visible from row j - coff on: earlier query tiles see none of this workgroup's (this wave's) keys
    const int qs_first = p.causal ? (max(0, key0 - p.coff) / BQ) * BQ : 0;
    const int ntile = qs_first < nq ? (nq - qs_first + BQ - 1) / BQ : 0;
    const int it_first = p.causal ? max(0, kw0 - p.coff) / BQ - qs_first / BQ : 0;
    LiveScan<false, BQ> scan;
    if (use_bm) scan.init(p, key0, min(key0 + BK, nk), qs_first, nq, ntile, lane);
    auto next_live = [&](int it) { return use_bm ? scan.next(it) : it; };
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;

    int it = next_live(0), cur = 0;
    if (it < ntile) stage(0, qs_first + it * BQ);
    dma_wait_all();
    __syncthreads();
    // feed-only iterations first (tiles before this wave's first visible row), then the computing ones
    while (it < min(it_first, ntile)) {
        const int itn = next_live(it + 1);
        if (itn < ntile) stage(cur ^ 1, qs_first + itn * BQ);
        dma_wait_all();
        __syncthreads();
        cur ^= 1;
        it = itn;
    }
    while (it < ntile) {
        const int itn = next_live(it + 1);
        if (itn < ntile) stage(cur ^ 1, qs_first + itn * BQ);
        const int qs = qs_first + it * BQ;
        const char* Qt = Qs + cur * Q_BYTES;
        const char* Ot = Os + cur * Q_BYTES;
        const float* Lt = Ls + cur * 128;
#pragma unroll
        for (int qb = 0; qb < BQ / 32; ++qb) {
            const int rb0 = qs + 32 * qb;              // first row of the block; register i holds row rb0 + 4 h + rc(i)
            unsigned vis = 0xffffu, kp = 0xffffu;
            if constexpr (FEAT & kFeatMask) {
                if constexpr (M16) { if (msk.on) vis = dense_bits_k_lds(msk, rb0, nk, kw0, lane, reinterpret_cast<char*>(Ls + 2 * 128) + 1024 * w); }
                else if (msk.on) vis = dense_bits_k(msk, rb0 + 4 * h, nk, key);
                if (use_bm && p.bmask[(min(rb0, nq - 1) / p.br) * p.nbc + cbw] == 0) vis = 0;
            }
            if constexpr (FEAT & kFeatDrop) {
                if (drop) kp = keep_bits_k(p, (unsigned)bh * p.nqh, rb0 + 4 * h, key);
            }
            // a block of which this wave sees nothing is not computed (wave-uniform; see the forward kernel)
            if ((FEAT & kFeatMask) && !__any(vis != 0)) continue;
            // masked: the row precedes the key's first visible row (causal), or the key lies past nk: rc(i) < thr
            const bool need_mask = (p.causal && (kw0 + 31 - p.coff > rb0)) || (kw0 + 32 > nk);
            const int thr = !need_mask ? -1 : (key >= nk ? 64 : (p.causal ? key - p.coff - rb0 - 4 * h : -1));
            int kofs = 32 * w * 2 * D;
            asm volatile("" : "+v"(kofs));
            u32x4 pp[2], sp[2];
            [[maybe_unused]] u32x4 pu[2];   // dropout: the un-dropped 16-bit P (dS = P (dP_drop - delta))
            {
                f32x16 sacc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(Lt + 32 * qb + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc[4 * g + j] = a[j];
                }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const int ro = TileSwz<D>::off(r, 2 * ks + h);
                    const s16x8 qa = *reinterpret_cast<const s16x8*>(Qt + 32 * qb * 2 * D + ro);
                    const s16x8 kf = *reinterpret_cast<const s16x8*>(Ks + ro + kofs);
                    sacc = mfma32<Tag>(qa, kf, sacc);
                }
                // wave-uniform: blocks that every lane sees whole (most of a structured mask) skip the selects
                const bool plain = !need_mask && (!(FEAT & kFeatMask) || !__any(vis != 0xffffu));
                if (plain) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[i] = __builtin_amdgcn_exp2f(sacc[i] * c_log2);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        bool dead = rc_of(i) < thr;
                        if constexpr (FEAT & kFeatMask) dead = dead || !((vis >> i) & 1u);
                        sacc[i] = dead ? 0.f : __builtin_amdgcn_exp2f(sacc[i] * c_log2);
                    }
                }
                if constexpr (FEAT & kFeatDrop) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j) pu[s][j] = pack2<Tag>(sacc[8 * s + 2 * j], sacc[8 * s + 2 * j + 1]);
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[i] = ((kp >> i) & 1u) ? sacc[i] * p.keep_scale : 0.f;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[s][j] = pack2<Tag>(sacc[8 * s + 2 * j], sacc[8 * s + 2 * j + 1]);
            }
            {
                f32x16 pacc;
                f32x4 ndv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) ndv[g] = *reinterpret_cast<const f32x4*>(Lt + 64 + 32 * qb + 8 * g + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pacc[4 * g + j] = (FEAT & kFeatDrop) ? 0.f : ndv[g][j];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const s16x8 oa = *reinterpret_cast<const s16x8*>(Ot + 32 * qb * 2 * D + TileSwz<D>::off(r, 2 * ks + h));
                    pacc = mfma32<Tag>(oa, vf[ks], pacc);
                }
                if constexpr (FEAT & kFeatDrop) {   // dP' = keep / (1 - p) * (dO V^T) - delta
#pragma unroll
                    for (int i = 0; i < 16; ++i) pacc[i] = (((kp >> i) & 1u) ? pacc[i] * p.keep_scale : 0.f) + ndv[i >> 2][i & 3];
                }
                if constexpr (std::is_same<Tag, f16_tag>::value) mfma_result_fence(pacc);   // mul_pack<f16> reads pacc from asm
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        sp[s][j] = mul_pack<Tag>((FEAT & kFeatDrop) ? pu[s][j] : pp[s][j], pacc[8 * s + 2 * j], pacc[8 * s + 2 * j + 1]);
                if constexpr (std::is_same<Tag, f16_tag>::value) asm volatile("s_nop 1" : "+v"(sp[0]), "+v"(sp[1]));
            }
            if (D > 64) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x8 pb = *reinterpret_cast<s16x8*>(&pp[s]);
                const s16x8 sb = *reinterpret_cast<s16x8*>(&sp[s]);
                const int qa_ = 32 * qb + 16 * s + 4 * h + tq;
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int ch = 4 * db + 2 * g16 + (tp >> 1);
                    const int o1 = TileSwz<D>::off(qa_, ch) + 8 * (tp & 1);
                    const int o2 = TileSwz<D>::off(qa_ + 8, ch) + 8 * (tp & 1);
                    const s16x8 doT = cat8(lds_tr16(Ot + o1), lds_tr16(Ot + o2));
                    dva[db] = mfma32<Tag>(doT, pb, dva[db]);
                    const s16x8 qT = cat8(lds_tr16(Qt + o1), lds_tr16(Qt + o2));
                    dka[db] = mfma32<Tag>(qT, sb, dka[db]);
                }
                if (D > 64) __builtin_amdgcn_sched_barrier(0);
            }
        }
        dma_wait_all();
        __syncthreads();
        cur ^= 1;
        it = itn;
    }

    if (key < nk) {
        uint16_t* dkrow = dk + kbase + (size_t)key * DR;
        uint16_t* dvrow = dv + kbase + (size_t)key * DR;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 a, b;
                a[0] = pack2_rn<Tag>(dka[db][4 * g + 0] * p.scale, dka[db][4 * g + 1] * p.scale);
                a[1] = pack2_rn<Tag>(dka[db][4 * g + 2] * p.scale, dka[db][4 * g + 3] * p.scale);
                b[0] = pack2_rn<Tag>(dva[db][4 * g + 0], dva[db][4 * g + 1]);
                b[1] = pack2_rn<Tag>(dva[db][4 * g + 2], dva[db][4 * g + 3]);
                if (32 * db + 8 * g + 4 * h >= DR) continue;   // padded columns (DR is a multiple of 8)
                *reinterpret_cast<u32x2*>(dkrow + 32 * db + 8 * g + 4 * h) = a;
                *reinterpret_cast<u32x2*>(dvrow + 32 * db + 8 * g + 4 * h) = b;
            }
    }
}

// ------------------------------------------------------------------------------------------------ dQ
template <typename Tag, int D, int FEAT>
__global__ __launch_bounds__(512, 2) void exm_dq_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                        const uint16_t* __restrict__ v, const uint16_t* __restrict__ dout,
                                                        const float* __restrict__ nlse, const float* __restrict__ ndelta,
                                                        uint16_t* __restrict__ dq, ExParams p, float c_log2) {
    constexpr int NW = 8, BM = 32 * NW, BN = 64, NKS = D / 16, NDB = D / 32, TILE_BYTES = BN * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K tile | V tile]
    const int DR = p.d, nq = p.nq, nk = p.nk;
    const int nqt = (nq + BM - 1) / BM;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nqt;
    const int q0 = (L - bh * nqt) * BM;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t qbase = (size_t)bh * nq * DR, kbase = (size_t)bh * nk * DR;
    const int qrow = q0 + 32 * w + r;
    const bool live_row = qrow < nq;

    const buf_rsrc_t q_rs = make_rsrc(q + qbase, (unsigned)nq * DR * 2);
    const buf_rsrc_t o_rs = make_rsrc(dout + qbase, (unsigned)nq * DR * 2);
    s16x8 qf[NKS], of[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        qf[ks] = buf_load_frag(q_rs, frag_off(qrow, 16 * ks + 8 * h, DR, true));
        of[ks] = buf_load_frag(o_rs, frag_off(qrow, 16 * ks + 8 * h, DR, true));
    }
    const float nl = live_row ? nlse[(size_t)bh * nq + qrow] : 0.f;
    const float nd = live_row ? ndelta[(size_t)bh * nq + qrow] : 0.f;

    const rsrc_s_t k_rs = make_rsrc_s(k + kbase, (unsigned)nk * DR * 2);
    const rsrc_s_t v_rs = make_rsrc_s(v + kbase, (unsigned)nk * DR * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, DR);
    auto stage = [&](int buf, int k0) {
        char* kb_ = smem + buf * 2 * TILE_BYTES;
        dma_stage_tile<D, BN, NW>(k_rs, kb_, k0, dma_voff, w, DR);
        dma_stage_tile<D, BN, NW>(v_rs, kb_ + TILE_BYTES, k0, dma_voff, w, DR);
    };
    const MaskSrc msk = make_mask_src(p, bh);
    const bool use_bm = (FEAT & kFeatMask) && p.bmask != nullptr;
    const bool drop = (FEAT & kFeatDrop) && p.p_drop > 0.f;
    const unsigned hi = (unsigned)bh * p.nqh + ((unsigned)qrow >> 1);
    const int rbw = min(q0 + 32 * w, nq - 1) / p.br;

    f32x16 dqa[NDB];
#pragma unroll
    for (int t = 0; t < NDB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) dqa[t][i] = 0.f;

    const int kend = p.causal ? max(0, min(nk, q0 + BM + p.coff)) : nk;
    const int kend_w = p.causal ? max(0, min(nk, q0 + 32 * w + 32 + p.coff)) : nk;
    const int ntiles = (kend + BN - 1) / BN, ntiles_w = (kend_w + BN - 1) / BN;
    LiveScan<true, BN> scan;
    if (use_bm) scan.init(p, q0, min(q0 + BM, nq), 0, nk, ntiles, lane);
    auto next_live = [&](int t) { return use_bm ? scan.next(t) : t; };
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;

    int t = next_live(0), cur = 0;
    if (t < ntiles) stage(0, t * BN);
    dma_wait_all();
    __syncthreads();
    const bool late_stage = (FEAT & kFeatMask) && msk.on;   // see the forward kernel
    while (t < ntiles_w) {
        const int tn = next_live(t + 1);
        if (!late_stage && tn < ntiles) stage(cur ^ 1, tn * BN);
        const int k0 = t * BN;
        const char* Kt = smem + cur * 2 * TILE_BYTES;
        const char* Vt = Kt + TILE_BYTES;
        u32x4 dsb[2][2];
        unsigned visw[2] = {0xffffu, 0xffffu};
        if constexpr (FEAT & kFeatMask) {
            if (msk.on) {
                visw[0] = dense_bits_q(msk, qrow, nk, k0 + 4 * h, h);
                visw[1] = dense_bits_q(msk, qrow, nk, k0 + 32 + 4 * h, h);
                if (tn < ntiles) stage(cur ^ 1, tn * BN);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                if (use_bm && p.bmask[rbw * p.nbc + min(k0 + 32 * kb, nk - 1) / p.bc] == 0) visw[kb] = 0;
        }
        // a tile of which this wave sees nothing is not computed (wave-uniform; see the forward kernel)
        const bool any_vis = !(FEAT & kFeatMask) || __any((visw[0] | visw[1]) != 0) != 0;
        if (any_vis) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            unsigned vis = visw[kb], kp = 0xffffu;
            if constexpr (FEAT & kFeatDrop) {
                if (drop) kp = keep_bits_q(p, hi, qrow, k0 + 32 * kb + 4 * h);
            }
            f32x16 sacc, pacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sacc[i] = nl; pacc[i] = (FEAT & kFeatDrop) ? 0.f : nd; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int off = TileSwz<D>::off(32 * kb + r, 2 * ks + h);
                const s16x8 ka = *reinterpret_cast<const s16x8*>(Kt + off);
                sacc = mfma32<Tag>(ka, qf[ks], sacc);
                const s16x8 va = *reinterpret_cast<const s16x8*>(Vt + off);
                pacc = mfma32<Tag>(va, of[ks], pacc);
            }
            const bool need_mask = (p.causal && (k0 + 32 * kb + 31 > q0 + 32 * w + p.coff)) || (k0 + 32 * kb + 32 > nk);
            const int lim = p.causal ? min(qrow + p.coff, nk - 1) : nk - 1;
            const int thr = need_mask ? lim - (k0 + 32 * kb + 4 * h) : 64;
            // wave-uniform (not in the dropout build: two copies of its selects cost registers it does not have)
            const bool plain = !(FEAT & kFeatDrop) && !need_mask && (!(FEAT & kFeatMask) || !__any(vis != 0xffffu));
            if (plain) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float dpv = pacc[i];
                    if constexpr (FEAT & kFeatDrop) dpv = (((kp >> i) & 1u) ? dpv * p.keep_scale : 0.f) + nd;
                    pacc[i] = __builtin_amdgcn_exp2f(sacc[i] * c_log2) * dpv;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    bool dead = rc_of(i) > thr;
                    if constexpr (FEAT & kFeatMask) dead = dead || !((vis >> i) & 1u);
                    float dpv = pacc[i];
                    if constexpr (FEAT & kFeatDrop) dpv = (((kp >> i) & 1u) ? dpv * p.keep_scale : 0.f) + nd;
                    pacc[i] = dead ? 0.f : __builtin_amdgcn_exp2f(sacc[i] * c_log2) * dpv;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) dsb[kb][s][j] = pack2<Tag>(pacc[8 * s + 2 * j], pacc[8 * s + 2 * j + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x8 sb = *reinterpret_cast<s16x8*>(&dsb[kb][s]);
                const int key_a = 32 * kb + 16 * s + 4 * h + tq;
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int ch = 4 * db + 2 * g16 + (tp >> 1);
                    const s16x8 a = cat8(lds_tr16(Kt + TileSwz<D>::off(key_a, ch) + 8 * (tp & 1)),
                                         lds_tr16(Kt + TileSwz<D>::off(key_a + 8, ch) + 8 * (tp & 1)));
                    dqa[db] = mfma32<Tag>(a, sb, dqa[db]);
                }
            }
        }   // any_vis
        dma_wait_all();
        __syncthreads();
        cur ^= 1;
        t = tn;
    }
    while (t < ntiles) {
        const int tn = next_live(t + 1);
        if (tn < ntiles) stage(cur ^ 1, tn * BN);
        dma_wait_all();
        __syncthreads();
        cur ^= 1;
        t = tn;
    }
    if (live_row) {
        uint16_t* drow = dq + qbase + (size_t)qrow * DR;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * db + 8 * g + 4 * h >= DR) continue;
                u32x2 val;
                val[0] = pack2_rn<Tag>(dqa[db][4 * g + 0] * p.scale, dqa[db][4 * g + 1] * p.scale);
                val[1] = pack2_rn<Tag>(dqa[db][4 * g + 2] * p.scale, dqa[db][4 * g + 3] * p.scale);
                *reinterpret_cast<u32x2*>(drow + 32 * db + 8 * g + 4 * h) = val;
            }
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool ex_mfma_supported(const ExArgs& a) {
    if (a.dtype != 1 && a.dtype != 2) return false;
    if (a.d < 8 || a.d > 128 || a.d % 8 != 0) return false;
    if (!(a.scale > 0.f)) return false;                       // the running max is taken on the raw scores
    if (a.block_mask && (a.br % 32 != 0 || a.bc % 32 != 0)) return false;
    if (a.mask && a.nq * a.nk >= ((int64_t)1 << 31)) return false;   // 32-bit byte offsets into one (b,h)'s mask
    if (a.bh * ((a.nq + 1) / 2) >= ((int64_t)1 << 32)) return false;  // the generator's row-pair counter
    return true;
}

template <typename Tag, int D, int FEAT>
static hipError_t exm_fwd_t(const ExArgs& a, hipStream_t st) {
    const size_t smem = 2 * 2 * 128 * D * 2;
    auto kern = exm_fwd_kernel<Tag, D, FEAT>;
    hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(((a.nq + 255) / 256) * a.bh));
    ProfScope ps(K_EX_FWD, st);
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k, (const uint16_t*)a.v,
                       (uint16_t*)a.o, a.lse, make_ex_params(a), a.scale * 1.4426950408889634f);
    return hipGetLastError();
}

template <typename Tag, int D, int FEAT>
static hipError_t exm_bwd_t(const ExArgs& a, hipStream_t st) {
    const long long rows = (long long)a.bh * a.nq;
    float* nlse = reinterpret_cast<float*>(a.workspace);
    float* ndelta = nlse + ((rows + 63) & ~63ll);
    const ExParams p = make_ex_params(a);
    const float c = a.scale * 1.4426950408889634f;
    ProfScope ps(K_EX_BWD, st);
    hipLaunchKernelGGL(exm_prep_kernel<Tag>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, (const uint16_t*)a.o,
                       (const uint16_t*)a.dout, (const float*)a.lse, nlse, ndelta, rows, (int)a.d, 1.f / a.scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.nk > 0) {
        const bool m16 = (FEAT & kFeatMask) && p.mask != nullptr && (p.nk & 15) == 0 && (p.mask_bh & 15) == 0 && (((uintptr_t)p.mask) & 15) == 0;
        const size_t smem = (size_t)256 * D * 2 + 4 * 64 * D * 2 + 2 * 128 * sizeof(float) + (m16 ? 8 * 1024 : 0);   // + the waves' mask images
        auto kern = m16 ? exm_dkdv_kernel<Tag, D, FEAT, (FEAT & kFeatMask) != 0> : exm_dkdv_kernel<Tag, D, FEAT, false>;
        e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        dim3 grid((unsigned)(((a.nk + 255) / 256) * a.bh));
        hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k, (const uint16_t*)a.v,
                           (const uint16_t*)a.dout, (const float*)nlse, (const float*)ndelta, (uint16_t*)a.dk, (uint16_t*)a.dv, p, c);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (a.nq > 0) {
        const size_t smem = 2 * 2 * 64 * D * 2;
        auto kern = exm_dq_kernel<Tag, D, FEAT>;
        e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        dim3 grid((unsigned)(((a.nq + 255) / 256) * a.bh));
        hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k, (const uint16_t*)a.v,
                           (const uint16_t*)a.dout, (const float*)nlse, (const float*)ndelta, (uint16_t*)a.dq, p, c);
        e = hipGetLastError();
    }
    return e;
}

template <typename Tag, int D>
static hipError_t exm_by_feat(const ExArgs& a, bool backward, hipStream_t st) {
    const bool masks = a.mask || a.block_mask, drop = a.dropout_p > 0.0;
    if (drop) return backward ? exm_bwd_t<Tag, D, 3>(a, st) : exm_fwd_t<Tag, D, 3>(a, st);
    if (masks) return backward ? exm_bwd_t<Tag, D, 1>(a, st) : exm_fwd_t<Tag, D, 1>(a, st);
    return backward ? exm_bwd_t<Tag, D, 0>(a, st) : exm_fwd_t<Tag, D, 0>(a, st);
}

hipError_t launch_ex_mfma(const ExArgs& a, bool backward, hipStream_t st) {
    if (a.dtype == 2) return a.d > 64 ? exm_by_feat<bf16_tag, 128>(a, backward, st) : exm_by_feat<bf16_tag, 64>(a, backward, st);
    return a.d > 64 ? exm_by_feat<f16_tag, 128>(a, backward, st) : exm_by_feat<f16_tag, 64>(a, backward, st);
}

}  // namespace fa
