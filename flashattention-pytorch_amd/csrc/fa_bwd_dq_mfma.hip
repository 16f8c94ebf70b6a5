// FlashAttention backward, dQ pass, for gfx950 (bf16 / f16; 64-, 128- and 256-wide tiles, narrower head dims padded).
// Launched FIRST at d > 64: it also makes the row constants -lse/scale and -rowsum(dO*O) for the dK/dV pass (PREP).
//
//   dQ[q] = scale * sum_key dS[q][key] K[key],   dS = P * (dP - delta),  P = exp(S - lse),  dP = dO V^T
//   (csrc/fa2/fa2_bwd.cu:91-103 restricted to the dQ accumulation; S and dP are recomputed here so that the
//   backward needs no cross-workgroup sum: no atomics, bitwise reproducible.)
//
// Same decomposition as the forward kernel (fa_fwd_mfma.hip): workgroup = 8 waves = 256 query rows, each wave 32
// rows; K and V tiles of 64 keys double-buffered in LDS.  Products are "swapped" so the QUERY sits on the lane:
//   S^T  = K  Q^T    (A: K rows from LDS,  B: Q fragments in registers)   initial accumulator = -lse/scale
//   dP^T = V dO^T    (A: V rows from LDS,  B: dO fragments in registers)  initial accumulator = -delta
//   dS^T = exp2(c S'^T) * dP'^T          (row constants are per-lane scalars here)
//   dQ^T += K^T dS^T (A: K^T by ds_read_b64_tr_b16, B: dS^T straight from the accumulator registers)
#include "fa_common.h"
#include "fa_kernels.h"
#include <type_traits>

namespace fa {

// PREP: make the row constants here (and store them for the dK/dV kernel) instead of reading them.  On for the 128-
// and 256-wide tiles; at d <= 64 a workgroup's main loop is too short to hide the extra prologue read of O and the
// separate bwd_prep_kernel launch is cheaper (1.50 vs 1.62 ms at N=4096, profiles/r01_tile_sweep.md).
template <typename Tag, int D, bool CAUSAL, int KT, int TPW, bool PAD, bool NLF, bool W4, bool PREP = (D > 64)>
__global__ __launch_bounds__((D == 256 || W4) ? 256 : 512, D == 256 ? 1 : 2) void bwd_dq_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                             const uint16_t* __restrict__ v,
                                                             const uint16_t* __restrict__ dout,
                                                             const uint16_t* __restrict__ o, const float* __restrict__ lse,
                                                             float* __restrict__ nlse, float* __restrict__ ndelta,
                                                             uint16_t* __restrict__ dq, int n, int nqt, float c_log2,
                                                             float scale, int dr) {
    const int DR = PAD ? dr : D;   // elements per tensor row (PAD: head dims below the tile width, fa_common.h)
    // D = 256: 4 waves, one per SIMD, with the whole 512-register file each (Q, dO fragments 128 + dQ^T 128 registers)
    // W4 (d <= 128): the same 4-wave shape but TWO workgroups per CU — the two waves of a SIMD then belong to different
    // workgroups, share no barrier and drift out of phase (one in its MFMA phase, the other in its exp2 / pack phase)
    constexpr int NW = (D == 256 || W4) ? 4 : 8, BM = 32 * NW, BN = 64 * KT, NKS = D / 16, NDB = D / 32;   // KT 64-key sub-tiles per LDS tile / barrier
    constexpr int TILE_BYTES = BN * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K tile | V tile]

    // TPW query tiles per workgroup (see fa_fwd_mfma.hip): non-causal consecutive tiles, causal the heavy + light pair
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
    const size_t base = (size_t)bh * n * DR;

    const buf_rsrc_t q_rs = make_rsrc(q + base, (unsigned)n * DR * 2);
    const buf_rsrc_t o_rs = make_rsrc(dout + base, (unsigned)n * DR * 2);
    const buf_rsrc_t y_rs = make_rsrc(o + base, (unsigned)n * DR * 2);   // the forward's output
    s16x8 qf[NKS], of[NKS];
    float nl, nd;   // row constants of this lane's query; a padded row gets S' = -1e30 -> P = 0
    // The row constants are made here, where a query row's dO is in registers anyway: -delta = -rowsum(dO * O)
    // (csrc/fa2/fa2_bwd.cu:57) from this lane's half of the row plus the other half's in lane ^ 32, and -lse / scale.
    // They are also stored for the dK/dV kernel, which is launched after this one (no separate preparation launch).
    auto load_rows = [&](int qt_) {
        const int row = qt_ * BM + 32 * w + r;
        const bool live = row < n;
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = buf_load_frag(q_rs, frag_off(row, 16 * ks + 8 * h, DR, PAD));
            of[ks] = buf_load_frag(o_rs, frag_off(row, 16 * ks + 8 * h, DR, PAD));
            if (PREP) {
                const s16x8 yf = buf_load_frag(y_rs, frag_off(row, 16 * ks + 8 * h, DR, PAD));
                const u32x4 a = *reinterpret_cast<const u32x4*>(&of[ks]), b = *reinterpret_cast<const u32x4*>(&yf);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    part += unpack_lo<Tag>(a[j]) * unpack_lo<Tag>(b[j]) + unpack_hi<Tag>(a[j]) * unpack_hi<Tag>(b[j]);
            }
        }
        if (PREP) {
            part += wave_half_swap(part);
            nl = live ? -lse[(size_t)bh * n + row] / scale : -1e30f;
            nd = live ? -part : 0.f;
            if (live && h == 0) {
                nlse[(size_t)bh * n + row] = nl;
                ndelta[(size_t)bh * n + row] = nd;
            }
        } else {
            nl = live ? nlse[(size_t)bh * n + row] : -1e30f;
            nd = live ? ndelta[(size_t)bh * n + row] : 0.f;
        }
    };
    load_rows(tile_of(0));

    // K / V tiles arrive by LDS-DMA (no staging registers); rows >= n read as zero
    const rsrc_s_t k_rs = make_rsrc_s(k + base, (unsigned)n * DR * 2);
    const rsrc_s_t v_rs = make_rsrc_s(v + base, (unsigned)n * DR * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, DR);
    const int dma_voff_b = D == 256 ? dma_lane_voff<D>(lane, w + NW, DR) : 0;
    auto stage = [&](int buf, int k0) {
        char* kb_ = smem + buf * 2 * TILE_BYTES;
        dma_stage_tile<D, BN, NW>(k_rs, kb_, k0, dma_voff, w, DR, dma_voff_b);
        dma_stage_tile<D, BN, NW>(v_rs, kb_ + TILE_BYTES, k0, dma_voff, w, DR, dma_voff_b);
    };

    f32x16 dqa[NDB];

    stage(0, 0);
    dma_wait_all();
    __syncthreads();

    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;

    int gbase = 0;   // K/V tiles consumed so far: tile t of the current query tile sits in buffer (gbase + t) & 1
    for (int it = 0; it < ntile_wg; ++it) {
    const int q0 = tile_of(it) * BM;
    const int qrow = q0 + 32 * w + r;
    const bool has_next = it + 1 < ntile_wg;
    const int kend = CAUSAL ? min(n, q0 + BM) : n;
    const int ntiles = (kend + BN - 1) / BN;
    const float nl2 = nl * c_log2;   // -lse * log2(e): P = exp2(c S + nl2)
#pragma unroll
    for (int t = 0; t < NDB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) dqa[t][i] = 0.f;
    auto stage_next = [&](int t) {   // next K/V tile of this query tile, or the first one of the next query tile
        const int nb = (gbase + t + 1) & 1;
        if (t + 1 < ntiles) stage(nb, (t + 1) * BN);
        else if (has_next) stage(nb, 0);
    };

    // tiles this wave computes: under the causal mask a tile whose first key lies past the wave's last row is
    // skipped.  Two loops instead of an `if` inside one: a conditional accumulate makes hipcc carry the
    // accumulators through copies.
    const int ntiles_w = CAUSAL ? min(ntiles, (q0 + 32 * w + 31) / BN + 1) : ntiles;
    for (int t = 0; t < ntiles_w; ++t) {
        const int k0 = t * BN;
        const int cur = (gbase + t) & 1;
        stage_next(t);   // nobody reads that buffer: all waves passed the last barrier
#pragma unroll 1
        for (int sub = 0; sub < KT; ++sub) {   // not unrolled: the body already sits at the register limit
            const int k0s = k0 + 64 * sub;                       // first key of this 64-key sub-tile
            const char* Kt = smem + cur * 2 * TILE_BYTES + sub * 64 * D * 2;
            const char* Vt = Kt + TILE_BYTES;
            u32x4 dsb[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 sacc, pacc;
#pragma unroll
                for (int i = 0; i < 16; ++i) { sacc[i] = NLF ? 0.f : nl; pacc[i] = nd; }   // NLF: S starts at 0 (free), -lse rides the exp2 fma
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const int off = TileSwz<D>::off(32 * kb + r, 2 * ks + h);
                    const s16x8 ka = *reinterpret_cast<const s16x8*>(Kt + off);
                    sacc = mfma32<Tag>(ka, qf[ks], sacc);
                    const s16x8 va = *reinterpret_cast<const s16x8*>(Vt + off);
                    pacc = mfma32<Tag>(va, of[ks], pacc);
                }
                const bool need_mask = (CAUSAL && (k0s + 32 * kb + 31 > q0 + 32 * w)) || (k0s + 32 * kb + 32 > n);
                // register i holds key k0 + 32 kb + 4 h + rc(i): one per-lane threshold, no branch
                const int lim = CAUSAL ? min(qrow, n - 1) : n - 1;
                const int thr = need_mask ? lim - (k0s + 32 * kb + 4 * h) : 64;
                if (need_mask) {   // wave-uniform: only diagonal / ragged blocks pay for the compare + select
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = __builtin_amdgcn_exp2f(NLF ? fmaf(sacc[i], c_log2, nl2) : sacc[i] * c_log2);
                        pacc[i] = ((i & 3) + 8 * (i >> 2) > thr) ? 0.f : p * pacc[i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) pacc[i] = __builtin_amdgcn_exp2f(NLF ? fmaf(sacc[i], c_log2, nl2) : sacc[i] * c_log2) * pacc[i];
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dsb[kb][s][j] = pack2<Tag>(pacc[8 * s + 2 * j], pacc[8 * s + 2 * j + 1]);
                __builtin_amdgcn_sched_barrier(0);   // keep the two key blocks' operand prefetch apart (register budget)
            }
            // ---- dQ^T += K^T dS^T
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
        }
        dma_wait_all();
        __syncthreads();
    }
    // causal: this wave's rows end before the workgroup's last tiles; keep feeding the other waves' tiles
    for (int t = ntiles_w; t < ntiles; ++t) {
        stage_next(t);
        dma_wait_all();
        __syncthreads();
    }
    gbase += ntiles;

    {
        u32x2 vals[NDB * 4];
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * db + g][0] = pack2_rn<Tag>(dqa[db][4 * g + 0] * scale, dqa[db][4 * g + 1] * scale);
                vals[4 * db + g][1] = pack2_rn<Tag>(dqa[db][4 * g + 2] * scale, dqa[db][4 * g + 3] * scale);
            }
        if (has_next) load_rows(tile_of(it + 1));   // the next tile's Q, dO, row constants fly while dQ goes out
        if (qrow < n) {
            uint16_t* drow = dq + base + (size_t)qrow * DR;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (PAD && 32 * db + 8 * g + 4 * h >= DR) continue;   // padded columns (DR is a multiple of 8)
                    *reinterpret_cast<u32x2*>(drow + 32 * db + 8 * g + 4 * h) = vals[4 * db + g];
                }
        }
    }
    }   // query tiles of this workgroup
}

template <typename Tag, int D, int KT, bool PAD = false, bool W4 = false>
static hipError_t launch_dq_kt(const BwdArgs& a, float* nlse, float* ndelta, hipStream_t st) {
    constexpr int NW = (D == 256 || W4) ? 4 : 8, BM = 32 * NW;
    const int nqt = (int)((a.n + BM - 1) / BM);
    const size_t smem = 2 * 2 * (64 * KT) * D * 2;
    const float c = a.scale * 1.4426950408889634f;
    // query tiles per workgroup: 2 under the causal mask (heavy + light pair: -9 ... -11 %) and at d = 128 (-1.7 %), else 1;
    // option dq_tpw overrides (1 | 2)
    int tpw = option(OPT_DQ_TPW);
    if (tpw == 0) tpw = (KT == 1 && (a.causal || D == 128)) ? 2 : 1;   // the fused prologue (PREP) is worth hiding at d = 128
    if (KT != 1 || D == 256) tpw = 1;
    dim3 grid((unsigned)(((nqt + tpw - 1) / tpw) * a.bh));
    ProfScope ps(K_BWD_DQ_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k,
                           (const uint16_t*)a.v, (const uint16_t*)a.dout, (const uint16_t*)a.o, a.lse, nlse, ndelta, (uint16_t*)a.dq, (int)a.n, nqt, c,
                           a.scale, (int)a.d);
        return hipGetLastError();
    };
    auto pick = [&](auto nlf) -> hipError_t {
        constexpr bool NLF = decltype(nlf)::value;
    if constexpr (KT == 1 && D != 256) {
            if (tpw == 2)
                return a.causal ? launch(bwd_dq_mfma_kernel<Tag, D, true, KT, 2, PAD, NLF, W4>) : launch(bwd_dq_mfma_kernel<Tag, D, false, KT, 2, PAD, NLF, W4>);
        }
        return a.causal ? launch(bwd_dq_mfma_kernel<Tag, D, true, KT, 1, PAD, NLF, W4>) : launch(bwd_dq_mfma_kernel<Tag, D, false, KT, 1, PAD, NLF, W4>);
    };
    // option dq_nlf = 1: -lse enters through the exp2 fma and the S accumulators start at 0, instead of -lse/scale as
    // the initial accumulator (16 v_mov per chain).  Measured a null at d = 128 and +1..3 % time at d = 64
    // (profiles/r01_tile_sweep.md), so it is the default only for the 256-wide tiles, whose accumulators live in
    // AGPRs (there the initial values are v_accvgpr_write's).
    if constexpr (D == 256) return pick(std::true_type{});
    if constexpr (KT == 1 && !PAD) {
        if (option(OPT_DQ_NLF) == 1) return pick(std::true_type{});
    }
    return pick(std::false_type{});
}

// K/V tile of the dQ pass: 64 keys per barrier is the measured winner (2.98 vs 3.43 ms, profiles/r01_tile_sweep.md);
// option dq_kt=2 selects two 64-key sub-tiles per barrier (sweep)
template <typename Tag, int D>
static hipError_t launch_dq_t(const BwdArgs& a, float* nlse, float* ndelta, hipStream_t st) {
    if (option(OPT_DQ_W4) == 1 || (option(OPT_DQ_W4) == 0 && option(OPT_DQ_KT) == 0 && option(OPT_DQ_TPW) == 0 && small_grid(a.bh, a.n, true)))
        return launch_dq_kt<Tag, D, 1, false, true>(a, nlse, ndelta, st);
    return option(OPT_DQ_KT) == 2 ? launch_dq_kt<Tag, D, 2>(a, nlse, ndelta, st) : launch_dq_kt<Tag, D, 1>(a, nlse, ndelta, st);
}

hipError_t launch_bwd_dq_mfma(const BwdArgs& a, float* nlse, float* ndelta, hipStream_t st) {
    // d = 128: the one-wave-per-SIMD stream kernel (fa_bwd_dq_w4.hip) is the default (-2.5 % against the 8-wave kernel
    // below at half the LDS operand traffic, profiles/r02_stream_kernels.md); small launches keep the 128-row tiles.
    // Option dq: 5 = always the stream kernel, 8 = the 8-wave kernel.
    const int dq_opt = option(OPT_DQ);
    const bool sweeping = option(OPT_DQ_KT) || option(OPT_DQ_TPW) || option(OPT_DQ_NLF) || option(OPT_DQ_W4);
    if (bwd_dq_w4_supported(a.dtype, a.d) && (dq_opt == 5 || (dq_opt == 0 && !sweeping && (!small_grid(a.bh, a.n, true) || !a.causal))))
        return launch_bwd_dq_w4(a, nlse, ndelta, st);
    if (a.d > 128) {   // 256-wide tiles, 4 waves (one per SIMD)
        if (a.dtype == 2) return a.d == 256 ? launch_dq_kt<bf16_tag, 256, 1, false>(a, nlse, ndelta, st) : launch_dq_kt<bf16_tag, 256, 1, true>(a, nlse, ndelta, st);
        return a.d == 256 ? launch_dq_kt<f16_tag, 256, 1, false>(a, nlse, ndelta, st) : launch_dq_kt<f16_tag, 256, 1, true>(a, nlse, ndelta, st);
    }
    if (a.d != 64 && a.d != 128) {   // head dims below the tile width: zero-padded inside the kernel
        if (a.dtype == 2) return a.d > 64 ? launch_dq_kt<bf16_tag, 128, 1, true>(a, nlse, ndelta, st) : launch_dq_kt<bf16_tag, 64, 1, true>(a, nlse, ndelta, st);
        return a.d > 64 ? launch_dq_kt<f16_tag, 128, 1, true>(a, nlse, ndelta, st) : launch_dq_kt<f16_tag, 64, 1, true>(a, nlse, ndelta, st);
    }
    if (a.dtype == 2) return a.d == 128 ? launch_dq_t<bf16_tag, 128>(a, nlse, ndelta, st) : launch_dq_t<bf16_tag, 64>(a, nlse, ndelta, st);
    return a.d == 128 ? launch_dq_t<f16_tag, 128>(a, nlse, ndelta, st) : launch_dq_t<f16_tag, 64>(a, nlse, ndelta, st);
}

}  // namespace fa
