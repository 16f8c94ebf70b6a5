// FlashAttention backward, dK/dV pass, for gfx950 (bf16 / f16; 64- and 128-wide tiles with two waves per SIMD, 256-wide
// tiles and small launches with 4-wave workgroups; narrower head dims padded).
//
//   dV[key] = sum_q P[q][key] dO[q],   dK[key] = scale * sum_q dS[q][key] Q[q]
//   P = exp(S - lse), dS = P * (dO V^T - delta)            (csrc/fa2/fa2_bwd.cu:91-104, the dK/dV half)
//
// workgroup = 8 waves = 256 keys of one (b,h); wave w owns keys key0 + 32 w .. + 31 and keeps their dK^T and dV^T
// (2 x D/32 accumulator tiles = 128 registers at D = 128) plus its V rows (B operand of dP) in registers; its K
// rows (B operand of S) live in the workgroup's K tile in LDS.  Q and dO arrive in tiles of 64 query rows by
// LDS-DMA, double buffered, one barrier per tile; every tile is read by rows (A operands of S and dP) and by
// columns through ds_read_b64_tr_b16 (A operands of dV^T += dO^T P and dK^T += Q^T dS).
// "Key on the lane": S[q][key] and dP[q][key] have the key as accumulator column, so P and dS feed the dV^T / dK^T
// products straight from the accumulator registers, and -lse/scale, -delta enter as the initial accumulators.
#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

template <typename Tag, int D, bool CAUSAL, bool KREG, int TPW, bool PAD, bool W4 = false, bool STG = false>
__global__ __launch_bounds__((D == 256 || W4) ? 256 : 512, D == 256 ? 1 : 2) void bwd_dkdv_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                               const uint16_t* __restrict__ v,
                                                               const uint16_t* __restrict__ dout,
                                                               const float* __restrict__ nlse,
                                                               const float* __restrict__ ndelta, uint16_t* __restrict__ dk,
                                                               uint16_t* __restrict__ dv, int n, int nkt, float c_log2,
                                                               float scale, int dr) {
    // D = 256: 4 waves (one per SIMD, 512 registers: dK^T and dV^T alone are 256), 128 keys, query tiles of 32 rows
    // W4 (d <= 128): 4 waves = 128 keys per workgroup for launches too small to fill the CUs with 256-key tiles
    constexpr int NW = (D == 256 || W4) ? 4 : 8, BK = 32 * NW, BQ = D == 256 ? 32 : 64, NKS = D / 16, NDB = D / 32;
    const int DR = PAD ? dr : D;   // elements per tensor row (PAD: head dims below the tile width, fa_common.h)
    constexpr int K_BYTES = BK * D * 2, Q_BYTES = BQ * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                      // [256][D]
    char* Qs = Ks + K_BYTES;              // [2][64][D]
    char* Os = Qs + 2 * Q_BYTES;          // [2][64][D]   (dO)
    float* Ls = reinterpret_cast<float*>(Os + 2 * Q_BYTES);  // [2][ 64 x -lse/scale | 64 x -delta ]

    // workgroup L works through TPW key tiles of one (b,h) (persistent: the next tile's K / V / first Q,dO tile are
    // requested before this tile's dK, dV stores, so prologue and epilogue overlap).  Under the causal mask the pair is
    // heavy + light (kt, nkt-1-kt): every workgroup then carries the same number of query tiles.
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
    const size_t base = (size_t)bh * n * DR;
    const size_t rbase = (size_t)bh * n;

    const rsrc_s_t k_rs = make_rsrc_s(k + base, (unsigned)n * DR * 2);
    const rsrc_s_t q_rs = make_rsrc_s(q + base, (unsigned)n * DR * 2);
    const rsrc_s_t o_rs = make_rsrc_s(dout + base, (unsigned)n * DR * 2);
    const rsrc_s_t l_rs = make_rsrc_s(nlse + rbase, (unsigned)n * 4);
    const rsrc_s_t d_rs = make_rsrc_s(ndelta + rbase, (unsigned)n * 4);
    const buf_rsrc_t v_rs = make_rsrc(v + base, (unsigned)n * DR * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, DR);
    const int dma_voff_b = D == 256 ? dma_lane_voff<D>(lane, w + NW, DR) : 0;

    auto stage = [&](int buf, int qs) {
        dma_stage_tile<D, BQ, NW>(q_rs, Qs + buf * Q_BYTES, qs, dma_voff, w, DR, dma_voff_b);
        dma_stage_tile<D, BQ, NW>(o_rs, Os + buf * Q_BYTES, qs, dma_voff, w, DR, dma_voff_b);
        // row constants: 64 floats each, one 4-byte LDS-DMA per lane (rows >= n read as 0: harmless, their dO is 0)
        if (w == 0) dma4_issue(l_rs, lds_addr_of(Ls + buf * 128), lane * 4, __builtin_amdgcn_readfirstlane(qs * 4));
        if (w == 1) dma4_issue(d_rs, lds_addr_of(Ls + buf * 128 + 64), lane * 4, __builtin_amdgcn_readfirstlane(qs * 4));
    };

    s16x8 vf[NKS];
    // KREG (d = 64 only: at d = 128 there is no room): the wave's K rows also stay in registers instead of being
    // re-read from the LDS tile for every 32-query block
    s16x8 kfr[KREG ? NKS : 1];
    // requests of one key tile: K tile by LDS-DMA, V fragments to registers, first Q/dO tile (buffer 0)
    auto begin_tile = [&](int kt_) {
        const int key0_ = kt_ * BK;
        const int key_ = key0_ + 32 * w + r;
        dma_stage_tile<D, BK, NW>(k_rs, Ks, key0_, dma_voff, w, DR, dma_voff_b);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) vf[ks] = buf_load_frag(v_rs, frag_off(key_, 16 * ks + 8 * h, DR, PAD));
        if (KREG) {
            const buf_rsrc_t kk_rs = make_rsrc(k + base, (unsigned)n * DR * 2);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) kfr[ks] = buf_load_frag(kk_rs, frag_off(key_, 16 * ks + 8 * h, DR, PAD));
        }
        stage(0, CAUSAL ? (key0_ / BQ) * BQ : 0);
    };

    begin_tile(tile_of(0));
    dma_wait_all();
    __syncthreads();

    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;

#pragma unroll 1
    for (int ip = 0; ip < TPW; ++ip) {
    const int kt = tile_of(ip);
    if (TPW > 1 && kt < 0) break;
    const int key0 = kt * BK;
    const int kw0 = key0 + 32 * w;        // first key of this wave
    const int key = kw0 + r;              // this lane's key
    f32x16 dka[NDB], dva[NDB];
#pragma unroll
    for (int t = 0; t < NDB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dka[t][i] = 0.f; dva[t][i] = 0.f; }
    const int qs_first = CAUSAL ? (key0 / BQ) * BQ : 0;   // earlier query tiles see none of this workgroup's keys
    const int ntile = (n - qs_first + BQ - 1) / BQ;
    // first tile this wave computes: earlier tiles hold only queries before the wave's first key (causal)
    const int it_first = CAUSAL ? (kw0 / BQ) - (qs_first / BQ) : 0;

    // feed-only iterations (see fa_fwd_mfma.hip: two loops instead of a conditional accumulate)
    for (int it = 0; it < min(it_first, ntile); ++it) {
        if (it + 1 < ntile) stage((it & 1) ^ 1, qs_first + (it + 1) * BQ);
        dma_wait_all();
        __syncthreads();
    }
    for (int it = it_first; it < ntile; ++it) {
        const int qs = qs_first + it * BQ;
        const int cur = it & 1;
        if (it + 1 < ntile) stage(cur ^ 1, qs + BQ);
        const char* Qt = Qs + cur * Q_BYTES;
        const char* Ot = Os + cur * Q_BYTES;
        const float* Lt = Ls + cur * 128;

#pragma unroll
        for (int qb = 0; qb < BQ / 32; ++qb) {
            // ---- S' = Q K^T - lse/scale (row constants as the initial accumulator), P = exp2(c S') packed to 16 bit,
            // then dP' = dO V^T - delta and dS = P dP'.  One chain at a time: at D = 128 the wave has ~96 registers
            // beside the resident dK^T / dV^T / V, and two live f32 tiles plus their operand prefetch do not fit.
            // Register i holds query qs + 32 qb + 4 h + rc(i), rc(i) = (i & 3) + 8 (i >> 2); it is masked when it
            // precedes this lane's key (causal) or the key lies past n: rc(i) < thr, one per-lane threshold.
            const bool need_mask = (CAUSAL && (kw0 + 31 > qs + 32 * qb)) || (kw0 + 32 > n);
            const int thr = !need_mask ? -1 : (key >= n ? 64 : (CAUSAL ? key - (qs + 32 * qb) - 4 * h : -1));
            // this wave's K rows sit 32 w rows into the K tile; the swizzle only depends on the row modulo 16, so
            // the address is the Q row-read pattern plus a per-wave offset (recomputed per tile: saves 8 registers)
            int kofs = 32 * w * 2 * D;
            asm volatile("" : "+v"(kofs));
            u32x4 pp[2], sp[2];
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
                    const s16x8 kf = KREG ? kfr[KREG ? ks : 0] : *reinterpret_cast<const s16x8*>(Ks + ro + kofs);
                    sacc = mfma32<Tag>(qa, kf, sacc);
                }
                if (need_mask) {   // wave-uniform: only diagonal / ragged blocks pay for the compare + select
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        sacc[i] = ((i & 3) + 8 * (i >> 2) < thr) ? 0.f : __builtin_amdgcn_exp2f(sacc[i] * c_log2);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[i] = __builtin_amdgcn_exp2f(sacc[i] * c_log2);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[s][j] = pack2<Tag>(sacc[8 * s + 2 * j], sacc[8 * s + 2 * j + 1]);
            }
            {
                f32x16 pacc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(Lt + 64 + 32 * qb + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pacc[4 * g + j] = b[j];
                }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const s16x8 oa = *reinterpret_cast<const s16x8*>(Ot + 32 * qb * 2 * D + TileSwz<D>::off(r, 2 * ks + h));
                    pacc = mfma32<Tag>(oa, vf[ks], pacc);
                }
                // dS = P dP' with the 16-bit P that also feeds dV (one rounding of P, shared by both products)
                if constexpr (std::is_same<Tag, f16_tag>::value) mfma_result_fence(pacc);   // mul_pack<f16> reads pacc from asm
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        sp[s][j] = mul_pack<Tag>(pp[s][j], pacc[8 * s + 2 * j], pacc[8 * s + 2 * j + 1]);
                // mul_pack<f16> writes dS from asm: two wait states before an MFMA may read it (hipcc sees no VALU there)
                if constexpr (std::is_same<Tag, f16_tag>::value) asm volatile("s_nop 1" : "+v"(sp[0]), "+v"(sp[1]));
            }
            if (D > 64) __builtin_amdgcn_sched_barrier(0);
            // ---- dV^T += dO^T P ,  dK^T += Q^T dS   (A operands: transposed reads of the dO / Q tiles)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x8 pb = *reinterpret_cast<s16x8*>(&pp[s]);
                const s16x8 sb = *reinterpret_cast<s16x8*>(&sp[s]);
                const int qa_ = 32 * qb + 16 * s + 4 * h + tq;   // rows (queries) of the first 4-row block; second is +8
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
        dma_wait_all();   // this wave's share of the next tile has landed
        __syncthreads();
    }

    // ---- epilogue: dK = scale * dK^T (transposed back on the store), dV.  Every wave is past the last barrier, so
    // the K tile and both Q/dO buffers are free: the next key tile's loads go out ahead of the stores.
    const int kt_next = (TPW > 1 && ip + 1 < TPW) ? tile_of(ip + 1) : -1;
    if (TPW > 1 && kt_next >= 0) begin_tile(kt_next);
    if (STG && TPW == 1 && D != 256) {
        // (one key tile per workgroup; the paired form keeps the direct stores: a second epilogue path costs it spills)
        // the K tile in LDS is dead, so each wave stages its 32 x D results in its own
        // 32-row slice of it and stores whole rows (1 KiB per instruction instead of 64 scattered 8-byte pieces)
        char* stg = Ks + w * 32 * D * 2;
        u32x2 vals[NDB * 4];
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * db + g][0] = pack2_rn<Tag>(dka[db][4 * g + 0] * scale, dka[db][4 * g + 1] * scale);
                vals[4 * db + g][1] = pack2_rn<Tag>(dka[db][4 * g + 2] * scale, dka[db][4 * g + 3] * scale);
            }
        store_rows_via_lds<D>(stg, vals, dk + base, kw0, n, lane, DR);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * db + g][0] = pack2_rn<Tag>(dva[db][4 * g + 0], dva[db][4 * g + 1]);
                vals[4 * db + g][1] = pack2_rn<Tag>(dva[db][4 * g + 2], dva[db][4 * g + 3]);
            }
        store_rows_via_lds<D>(stg, vals, dv + base, kw0, n, lane, DR);
    } else if (key < n) {
        uint16_t* dkrow = dk + base + (size_t)key * DR;
        uint16_t* dvrow = dv + base + (size_t)key * DR;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 a, b;
                a[0] = pack2_rn<Tag>(dka[db][4 * g + 0] * scale, dka[db][4 * g + 1] * scale);
                a[1] = pack2_rn<Tag>(dka[db][4 * g + 2] * scale, dka[db][4 * g + 3] * scale);
                b[0] = pack2_rn<Tag>(dva[db][4 * g + 0], dva[db][4 * g + 1]);
                b[1] = pack2_rn<Tag>(dva[db][4 * g + 2], dva[db][4 * g + 3]);
                if (PAD && 32 * db + 8 * g + 4 * h >= DR) continue;   // padded columns (DR is a multiple of 8)
                *reinterpret_cast<u32x2*>(dkrow + 32 * db + 8 * g + 4 * h) = a;
                *reinterpret_cast<u32x2*>(dvrow + 32 * db + 8 * g + 4 * h) = b;
            }
    }
    if (TPW > 1 && kt_next >= 0) {
        dma_wait_all();
        __syncthreads();
    }
    }
}

template <typename Tag, int D, bool PAD = false, bool W4 = false>
static hipError_t launch_dkdv_t(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st) {
    constexpr int NW = (D == 256 || W4) ? 4 : 8, BK = 32 * NW, BQ = D == 256 ? 32 : 64;
    const int nkt = (int)((a.n + BK - 1) / BK);
    const size_t smem = (size_t)BK * D * 2 + 4 * BQ * D * 2 + 2 * 128 * sizeof(float);
    const float c = a.scale * 1.4426950408889634f;
    // key tiles per workgroup: option dkdv_tpw (1 | 2), default see DESIGN.md
    int tpw = option(OPT_DKDV_TPW);
    if (tpw == 0) tpw = (a.causal && !W4) ? 2 : 1;   // small launches: as many workgroups as possible
    dim3 grid((unsigned)(((nkt + tpw - 1) / tpw) * a.bh));
    ProfScope ps(K_BWD_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k,
                           (const uint16_t*)a.v, (const uint16_t*)a.dout, nlse, ndelta, (uint16_t*)a.dk, (uint16_t*)a.dv,
                           (int)a.n, nkt, c, a.scale, (int)a.d);
        return hipGetLastError();
    };
    if constexpr (D == 64 && !PAD && !W4) {
        if (option(OPT_DKDV_KREG) == 1) {
            grid = dim3((unsigned)(nkt * a.bh));   // (one tile per workgroup: the variant has no pairing; until round 2 the causal sweep ran it on half a grid)
            return a.causal ? launch(bwd_dkdv_mfma_kernel<Tag, D, true, true, 1, PAD, W4>) : launch(bwd_dkdv_mfma_kernel<Tag, D, false, true, 1, PAD, W4>);
        }
    }
    if (tpw == 2)
        return a.causal ? launch(bwd_dkdv_mfma_kernel<Tag, D, true, false, 2, PAD, W4>) : launch(bwd_dkdv_mfma_kernel<Tag, D, false, false, 2, PAD, W4>);
    // staged row stores in the epilogue (option dkdv_stg: 1 on, 2 off; default by measurement, see the sweep notes)
    const int so = option(OPT_DKDV_STG);
    const bool stg = so == 1 || (so == 0 && D == 128);
    if constexpr (!PAD && !W4) {
        if (stg) return a.causal ? launch(bwd_dkdv_mfma_kernel<Tag, D, true, false, 1, PAD, W4, true>) : launch(bwd_dkdv_mfma_kernel<Tag, D, false, false, 1, PAD, W4, true>);
    }
    return a.causal ? launch(bwd_dkdv_mfma_kernel<Tag, D, true, false, 1, PAD, W4>) : launch(bwd_dkdv_mfma_kernel<Tag, D, false, false, 1, PAD, W4>);
}

hipError_t launch_bwd_dkdv_mfma(const BwdArgs& a, const float* nlse, const float* ndelta, hipStream_t st) {
    // d = 128: the one-wave-per-SIMD stream kernel (fa_bwd_dkdv_w4.hip) is the default (-13 % non-causal, -8 ... -10 %
    // causal against the 8-wave kernel below, profiles/r02_dkdv_stream.md); small CAUSAL launches keep the 128-key tiles (without
    // the mask the stream kernel wins at every size: profiles/r02_small_launches.md).
    // Option dkdv: 5 = always the stream kernel, 8 = the 8-wave kernel.
    const int dk_opt = option(OPT_DKDV);
    const bool sweeping = option(OPT_DKDV_KREG) == 1 || option(OPT_DKDV_TPW) || option(OPT_DKDV_STG);   // (dkdv_kreg >= 2: forms of the stream kernel)
    if (bwd_dkdv_w4_supported(a.dtype, a.d) && (dk_opt == 5 || (dk_opt == 0 && !sweeping && (a.causal ? !small_grid(a.bh, a.n, true) : true))))   // (round 3: causal rows of <= 1024 too — 8 - 10 % ahead of the 8-wave kernel since the stream kernel's wave starts were fixed)
        return launch_bwd_dkdv_w4(a, nlse, ndelta, st);
    if (a.d > 128) {   // 256-wide tiles, 4 waves (one per SIMD)
        if (a.dtype == 2) return a.d == 256 ? launch_dkdv_t<bf16_tag, 256, false>(a, nlse, ndelta, st) : launch_dkdv_t<bf16_tag, 256, true>(a, nlse, ndelta, st);
        return a.d == 256 ? launch_dkdv_t<f16_tag, 256, false>(a, nlse, ndelta, st) : launch_dkdv_t<f16_tag, 256, true>(a, nlse, ndelta, st);
    }
    if (a.d != 64 && a.d != 128) {   // head dims below the tile width: zero-padded inside the kernel
        if (a.dtype == 2) return a.d > 64 ? launch_dkdv_t<bf16_tag, 128, true>(a, nlse, ndelta, st) : launch_dkdv_t<bf16_tag, 64, true>(a, nlse, ndelta, st);
        return a.d > 64 ? launch_dkdv_t<f16_tag, 128, true>(a, nlse, ndelta, st) : launch_dkdv_t<f16_tag, 64, true>(a, nlse, ndelta, st);
    }
    if (option(OPT_DKDV_KREG) != 1 && option(OPT_DKDV_TPW) == 0 && small_grid(a.bh, a.n, true)) {   // 128-key tiles on 4 waves
        if (a.dtype == 2) return a.d == 128 ? launch_dkdv_t<bf16_tag, 128, false, true>(a, nlse, ndelta, st) : launch_dkdv_t<bf16_tag, 64, false, true>(a, nlse, ndelta, st);
        return a.d == 128 ? launch_dkdv_t<f16_tag, 128, false, true>(a, nlse, ndelta, st) : launch_dkdv_t<f16_tag, 64, false, true>(a, nlse, ndelta, st);
    }
    if (a.dtype == 2) return a.d == 128 ? launch_dkdv_t<bf16_tag, 128>(a, nlse, ndelta, st) : launch_dkdv_t<bf16_tag, 64>(a, nlse, ndelta, st);
    return a.d == 128 ? launch_dkdv_t<f16_tag, 128>(a, nlse, ndelta, st) : launch_dkdv_t<f16_tag, 64>(a, nlse, ndelta, st);
}

}  // namespace fa
