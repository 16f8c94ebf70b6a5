// FlashAttention backward, dQ as ONE product over stored dS (gfx950, bf16 / f16, head_dim 128).
//
//   dQ[q] = scale * sum_key dS[q][key] K[key]                      (csrc/fa2/fa2_bwd.cu:100-103, the dQ accumulation)
//
// The dK/dV stream kernel (fa_bwd_dkdv_w4.hip, DS variant) has every dS element in registers, packed, as the B operand of
// its dK^T products; it stores those 16-byte pieces as they are (8.6 GB at 256 x 4096 x 4096: 3 % of the 288 GB) and this
// kernel makes dQ from them.  Against the recomputing dQ pass (fa_bwd_dq_w4.hip: S^T, dP^T, exp2 and dQ^T = 48 MFMAs per
// 64 x 32 block) that is 16 MFMAs per block and no vector work: the pass is bound by reading dS once from HBM
// (N^2 * 2 bytes per (b,h)), not by the matrix pipe, and no product is computed twice any more (5 products, not 7).
//
// Mapping:
//   * workgroup = 4 waves = 256 query rows, a wave owns 64 (two 32-row blocks): dQ^T of both in 128 accumulation
//     registers.  Stage = 32 keys: the K rows (8 KiB, shared, TileSwz image) and per wave its two dS tiles (2 x 2 KiB,
//     private), all by LDS-DMA, six buffers, five stages ahead (80 KiB of dS in flight per CU), counted vmcnt waits, one
//     barrier per stage.  Measured at 256 x 4096 x 4096: 1.70 ms; the same kernel without its LDS reads and MFMAs 1.63 —
//     it is the transfer that takes the time (5.1 TB/s of dS + the K rows from L2); three or four buffers instead of
//     six, nt on the dS loads: no difference (profiles/r02_ds_handover.md).
//   * dQ^T[d][q] += K^T[d][key] dS^T[key][q]: both operands are read TRANSPOSED from LDS (ds_read_b64_tr_b16): K^T from
//     the row-major K tile as the dK/dV kernels read Q^T, dS^T from the tile as stored — [16-query half s][key r][h][8
//     queries] in memory, re-blocked by the DMA's per-lane source address into [r >> 2][s][r & 3][h][8] so that the two
//     16-lane groups of a half wave read different banks.  The lane that ends up with column c of a 32-query block
//     holds query 16 (c >> 4) + 8 ((c & 7) >> 2) + 4 ((c >> 3) & 1) + (c & 3): the dK/dV kernel's register order.
//   * causal: a (32-query, 64-key) block the dK/dV kernel never wrote (every query before every key) is not loaded: its
//     DMA pieces are sent out of range, which fills the tile in LDS with zeros, and the products of the stage add nothing
//     (the stage body has one form and no branch).
#include "fa_common.h"
#include "fa_kernels.h"
#include <type_traits>
#include <utility>

namespace fa {

template <typename Tag, bool CAUSAL, int NW>
__global__ __launch_bounds__(64 * NW, 1) void bwd_dq_ds_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ ds,
                                                           uint16_t* __restrict__ dq, int n, int nk, int nqt, int nqb, int nkb32,
                                                           float scale) {
    constexpr int D = 128, NDB = 4, BM = 64 * NW, SK = 32, NBUF = NW == 8 ? 3 : 6, AHEAD = NBUF - 1;
    constexpr int K_BYTES = SK * D * 2;             // 8 KiB: the stage's K rows
    constexpr int DS_W = 2 * 2048;                  // 4 KiB: a wave's two dS tiles of the stage
    constexpr int BUF = K_BYTES + NW * DS_W;        // 24 / 40 KiB
    constexpr int KP = 8 / NW;                      // K pieces per wave and stage
    constexpr int PIECES = KP + 4;                  // per wave and stage: K, 2 x 2 of dS
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // Under the causal mask a workgroup takes a PAIR of query tiles, qt and nqt - 1 - qt (the tile with the most keys and the one
    // with the fewest): every workgroup then reads the same number of stages.  One tile per workgroup, most keys first, left the
    // launch a tail of 12 - 15 % (list scheduling of jobs of 1 .. nqt units in (b,h)-major order).
    constexpr int TPW = CAUSAL ? 2 : 1;
    const int npair = (nqt + TPW - 1) / TPW;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / npair;
    const int jp = L - bh * npair;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int coff = nk - n;
    const size_t kvbase = (size_t)bh * nk * D, base = (size_t)bh * n * D;
    const rsrc_s_t k_rs = make_rsrc_s(k + kvbase, (unsigned)nk * D * 2);
    const int k_voff = dma_lane_voff<D>(lane, w, D);
    // dS tile, LDS chunk p = 64 i + lane of piece i holds [r >> 2][s][r & 3][h]: r = 16 i + 4 (lane >> 4) + ((lane >> 1) & 3),
    // s = (lane >> 3) & 1, h = lane & 1; in memory it sits at 1024 s + 32 r + 16 h
    const int s_voff = 1024 * ((lane >> 3) & 1) + 32 * (4 * (lane >> 4) + ((lane >> 1) & 3)) + 16 * (lane & 1);
    const unsigned bbase = lds_addr_of(smem);
    // lane-constant operand addresses inside a buffer (transposed reads: 4-row blocks at rows 4 h + tq (+ 8); the
    // k-step and the query block are immediate offsets)
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    unsigned klo[NDB], khi[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const int ch = 4 * db + 2 * g16 + (tp >> 1);
        klo[db] = TileSwz<D>::off(4 * h + tq, ch) + 8 * (tp & 1);
        khi[db] = TileSwz<D>::off(4 * h + tq + 8, ch) + 8 * (tp & 1);
    }
    const unsigned slo = K_BYTES + w * DS_W + 256 * h + 128 * g16 + 32 * tq + 8 * tp;   // rows 4 h + tq; + 512: rows + 8

#pragma unroll 1
    for (int ip = 0; ip < TPW; ++ip) {
    const int qt = CAUSAL ? (ip == 0 ? nqt - 1 - jp : jp) : jp;   // the long tile first
    if (ip > 0 && qt >= nqt - 1 - jp) break;                      // odd tile count: the middle tile is its own pair
    const int q0 = qt * BM, q0w = q0 + 64 * w;

    // stages of this workgroup; per wave and query block: is block (qb, stage t) one the dK/dV kernel wrote (and non-zero)?
    const int nst_all = (nk + SK - 1) / SK;
    const int last_row = min(q0 + BM, n) - 1;
    const int nst = CAUSAL ? min(nst_all, 2 * ((last_row + coff) / 64 + 1)) : nst_all;
    const int qbi0 = (q0w >> 5);                    // the wave's first 32-query block
    auto live = [&](int qb, int t) -> bool {        // wave-uniform
        if (qbi0 + qb >= nqb || t >= nst) return false;
        return !CAUSAL || 64 * (t >> 1) <= q0w + 32 * qb + 31 + coff;
    };

    // the wave's two rows of dS tiles are neighbours in memory: one descriptor, rows that do not exist are never requested
    const int rows_here = max(0, min(2, nqb - qbi0));
    const rsrc_s_t s_rs = make_rsrc_s(ds + ((size_t)bh * nqb + min(qbi0, nqb - 1)) * nkb32 * 1024, (unsigned)(rows_here * nkb32) * 2048u);
    auto stage = [&](int t) {
        const unsigned b = bbase + (unsigned)(t % NBUF) * BUF;
        const bool kl = t < nst;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int pc = w + NW * j;
            dma16_issue(k_rs, b + pc * 1024, kl ? k_voff : kOobOff, __builtin_amdgcn_readfirstlane((SK * t + 4 * pc) * 2 * D));
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const bool sl = live(qb, t);
            const int soff = __builtin_amdgcn_readfirstlane((qb * nkb32 + t) * 2048);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                dma16_issue(s_rs, b + K_BYTES + w * DS_W + qb * 2048 + i * 1024, sl ? s_voff + 512 * i : kOobOff, soff);
        }
    };

    f32x16 dqa[2][NDB];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) dqa[qb][db][i] = 0.f;

#pragma unroll
    for (int t = 0; t < AHEAD; ++t) stage(t);

#pragma unroll 1
    for (int t = 0; t < nst; ++t) {
        // this wave's pieces of stage t have landed (all but the newest AHEAD - 1 stages), then everybody's; the buffer
        // of stage t - 1 is free again
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AHEAD - 1) * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
        stage(t + AHEAD);
        const unsigned b = bbase + (unsigned)(t % NBUF) * BUF;
        // One form, no branch: a block that is not live (the causal diagonal's far side, a ragged end) was requested out of
        // range, so its tile in LDS is zeros and its products add nothing.
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // 16 keys: rows 16 ks + 4 h + (0 .. 3) and + 8 (the swizzle depends on the row modulo 16 only)
            s16x8 sb[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                sb[qb] = cat8(lds_tr16_at(b + slo + qb * 2048 + ks * 1024), lds_tr16_at(b + slo + qb * 2048 + ks * 1024 + 512));
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const s16x8 ka = cat8(lds_tr16_at(b + klo[db] + ks * 16 * 2 * D), lds_tr16_at(b + khi[db] + ks * 16 * 2 * D));
                dqa[0][db] = mfma32<Tag>(ka, sb[0], dqa[0][db]);
                dqa[1][db] = mfma32<Tag>(ka, sb[1], dqa[1][db]);
            }
        }
    }
    dma_wait_all();   // the stages past the end were requested too (out of range: zeros): nothing may still be writing LDS

    // ---- epilogue: dQ = scale * dQ^T; lane (c, h) holds elements 32 db + 8 g + 4 h .. + 3 of its query row `prow`.  Whole-row
    // stores through LDS (1 KiB contiguous per instruction instead of sixteen 8-byte pieces per lane: the store tail of a
    // workgroup is issue bound, guide T21): every DMA of every wave has landed behind the barrier, the stage buffers are dead,
    // and a wave stages one 32-row block at a time in 8 KiB of its own.
    __syncthreads();
    const int prow = 16 * (r >> 4) + 8 * ((r & 7) >> 2) + 4 * ((r >> 3) & 1) + (r & 3);
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
        store_rows_via_lds<D>(smem + w * 32 * D * 2, vals, dq + base, q0w + 32 * qb, n, lane, D, prow);
    }
    if (TPW > 1) __syncthreads();   // the staging rows are stage buffers again
    }   // query tiles of this workgroup
}

template <typename Tag>
static hipError_t launch_dq_ds_t(const BwdArgs& a, const void* ds, hipStream_t st) {
    // 512 query rows per workgroup (8 waves) while that makes at least 160 workgroups — pairs of tiles under the causal mask —
    // else 256 rows (4 waves): 32 x 4096 causal 0.390 against 0.417 ms, 16 x 8192 0.725 against 0.761; 48 x 4096 the other
    // way round, 0.637 against 0.675 (tools/bwd_variant_sweep.py).  Option dq_w4 = 3 forces the 4-wave form.
    const int64_t nk = a.nk > 0 ? a.nk : a.n;
    const int64_t nqt8 = (a.n + 511) / 512;
    const int nw = (option(OPT_DQ_W4) == 3 || (a.causal ? (nqt8 + 1) / 2 : nqt8) * a.bh < 160) ? 4 : 8;
    const int BM = 64 * nw;
    const int nqt = (int)((a.n + BM - 1) / BM);
    const size_t smem = nw == 8 ? 3 * (32 * 128 * 2 + 8 * 4096) : 6 * (32 * 128 * 2 + 4 * 4096);
    dim3 grid((unsigned)((a.causal ? (nqt + 1) / 2 : nqt) * a.bh));   // causal: pairs of query tiles
    ProfScope ps(K_BWD_DQ_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(64 * nw), smem, st, (const uint16_t*)a.k, (const uint16_t*)ds, (uint16_t*)a.dq, (int)a.n, (int)nk,
                           nqt, ds_tile_rows(a.n), ds_tile_cols(nk), a.scale);
        return hipGetLastError();
    };
    if (nw == 4) return a.causal ? launch(bwd_dq_ds_kernel<Tag, true, 4>) : launch(bwd_dq_ds_kernel<Tag, false, 4>);
    return a.causal ? launch(bwd_dq_ds_kernel<Tag, true, 8>) : launch(bwd_dq_ds_kernel<Tag, false, 8>);
}

hipError_t launch_bwd_dq_ds(const BwdArgs& a, const void* ds, hipStream_t st) {
    return a.dtype == 2 ? launch_dq_ds_t<bf16_tag>(a, ds, st) : launch_dq_ds_t<f16_tag>(a, ds, st);
}

}  // namespace fa
