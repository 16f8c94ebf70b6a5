// FlashAttention forward for gfx950, 16-bit inputs (bf16 / f16), head_dim 64 or 128.
//
// Replaces the host tile loop of csrc/fa2/fa2_fwd.cu:56-103 (and fa1/fa3 siblings) with one fused kernel:
// S = QK^T -> scale -> mask -> online softmax -> O += PV, fp32 accumulation, `o` in the input dtype, `lse` fp32.
//
// Decomposition (wave64, MFMA 32x32x16):
//   workgroup = 8 waves = 256 query rows of one (b,h); each wave owns 32 query rows (4 waves = 128 rows for the
//   256-wide tiles and for small launches).  K/V tiles of 128 keys go HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds; the
//   XOR swizzle is applied to the source address), double buffered, one barrier per tile; Q stays in registers as the
//   B operand.  A workgroup may work through two query tiles (persistent; causal heavy + light pairing).
//   S^T = K . Q^T  ("swapped" product): the 32x32 accumulator then has the QUERY on the lane
//   (col = lane & 31) and 16 keys per lane in registers, so the running max / sum of a query row are
//   per-lane scalars: the row reductions are in-register plus one exchange with lane ^ 32.
//   O^T += V^T . P^T : P^T is taken straight from the S^T accumulator registers (packed to 16 bit) as the
//   B operand — an accumulator tile used as the next MFMA's operand — and V^T comes from the row-major V tile
//   through the transposing LDS read ds_read_b64_tr_b16.  O^T again has the query on the lane, so the
//   online-softmax rescale is a per-lane scalar multiply.
//
// Accumulator layout of v_mfma_f32_32x32x16: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#include "fa_common.h"
#include "fa_kernels.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fa {

// ABL != 0: ablation builds for profiling only (wrong results on purpose; option fwd_abl, tools/ab.py):
//   bit 0: no exp2 / max / sum (P = S packed as is)   bit 1: no LDS-DMA, no barrier (every tile re-reads buffer 0)
//   bit 2: no LDS operand reads (K and V^T fragments are register constants)   bit 3: no LDS-DMA, barrier kept
//   bit 4: LDS-DMA issued but never waited for
// W4 (d <= 128): 4 waves = 128 query rows per workgroup, for launches whose 256-row tiles would leave CUs idle
// (bh * ceil(N / 256) below the CU count): twice the workgroups, each with one wave per SIMD.
template <typename Tag, int D, bool CAUSAL, int KB, bool RS_MFMA, bool LAZY, bool HS, int TPW, bool PAD, int ABL = 0, bool W4 = false>
__global__ __launch_bounds__((D == 256 || W4) ? 256 : 512, D == 256 ? 1 : 2) void fwd_mfma_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                          const uint16_t* __restrict__ v, uint16_t* __restrict__ o,
                                                          float* __restrict__ lse, int n, int nqt, float c_log2,
                                                          float scale, int dr) {
    const int DR = PAD ? dr : D;   // elements per tensor row (PAD: head dims below the tile width, fa_common.h)
    // D = 256: 4 waves, one per SIMD, with the whole 512-register file each (Q fragments 64 + O^T 128 registers)
    constexpr int NW = (D == 256 || W4) ? 4 : 8, BM = 32 * NW, BN = 32 * KB, NKS = D / 16, NDV = D / 32;   // KB = 32-key blocks per K/V tile
    constexpr int TILE_BYTES = BN * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K tile | V tile]

    // TPW query tiles per workgroup (persistent over the tiles of one (b,h)): the next tile's Q fragments and first
    // K/V tile are fetched under the tail of the current one, so only the first tile of a workgroup pays the full
    // prologue (all 256 CUs otherwise run their prologue / epilogue HBM bursts at the same time).  Non-causal: TPW
    // consecutive tiles.  Causal (TPW <= 2): the heavy tile nqt-1-g and the light tile g — equal work per workgroup.
    static_assert(TPW == 1 || KB == 4, "the persistent form stages the epilogue in one 128-key K/V buffer");
    static_assert(!CAUSAL || TPW <= 2, "causal pairing is defined for two tiles");
    const int gpb = (nqt + TPW - 1) / TPW;                     // workgroups per (b,h)
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / gpb;
    const int grp = L - bh * gpb;
    auto tile_of = [&](int i) { return CAUSAL ? (i == 0 ? nqt - 1 - grp : grp) : grp * TPW + i; };
    int ntile_wg = 0;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tt = tile_of(i);
        if (CAUSAL ? (i == 0 || grp < nqt - 1 - grp) : (tt < nqt)) ntile_wg = i + 1;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const size_t base = (size_t)bh * n * DR;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[qrow][16 ks + 8 h .. +7]
    const buf_rsrc_t q_rs = make_rsrc(q + base, (unsigned)n * DR * 2);
    s16x8 qf[NKS];
    auto load_q = [&](int qt_) {
        const int row = qt_ * BM + 32 * w + r;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = buf_load_frag(q_rs, frag_off(row, 16 * ks + 8 * h, DR, PAD));
    };
    load_q(tile_of(0));

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

    f32x16 oacc[NDV];
    float m_run, l_run;   // running max of the raw scores of this lane's query row; this half-wave's share of the row sum
    // RS_MFMA: the row sum rides the matrix pipe instead — one extra MFMA per k-step with an all-ones A operand gives
    // sum_key P[q][key] in every register of `lacc` (col = this lane's query, both lane halves included).
    f32x16 lacc;
    s16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (short)(sizeof(Tag) && std::is_same<Tag, bf16_tag>::value ? 0x3F80 : 0x3C00);

    stage(0, 0);
    dma_wait_all();
    __syncthreads();

    // lane-constant pieces of the transposed V read address
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;

    int gbase = 0;   // K/V tiles consumed so far by this workgroup: tile t of the current query tile sits in buffer (gbase + t) & 1
    for (int it = 0; it < ntile_wg; ++it) {
    const int q0 = tile_of(it) * BM;
    const int qrow = q0 + 32 * w + r;
    const bool has_next = it + 1 < ntile_wg;
    const int kend = CAUSAL ? min(n, q0 + BM) : n;
    const int ntiles = (kend + BN - 1) / BN;
#pragma unroll
    for (int t = 0; t < NDV; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
    m_run = -INFINITY;
    l_run = 0.f;
    // what to fetch while tile t is being consumed: the next K/V tile of this query tile, or — on its last tile — the
    // first K/V tile of the workgroup's next query tile (same (b,h), so the same rows 0 .. BN-1)
    auto stage_next = [&](int t) {
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
        const int cur = (ABL & 10) ? 0 : (gbase + t) & 1;
        if (!(ABL & 10)) stage_next(t);   // nobody reads that buffer: all waves passed the last barrier

        const char* Kt = smem + cur * 2 * TILE_BYTES;
        const char* Vt = Kt + TILE_BYTES;
        {
            f32x16 sacc[KB];
            if constexpr (!HS) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const s16x8 a = (ABL & 4) ? qf[(ks + 1) % NKS] : *reinterpret_cast<const s16x8*>(Kt + TileSwz<D>::off(32 * kb + r, 2 * ks + h));
                        sacc[kb] = mfma32<Tag>(a, qf[ks], sacc[kb]);
                    }
                }
            } else {
                // HS ("hand-scheduled" in source): left alone, hipcc sinks every LDS operand read to just before the MFMA
                // that consumes it (ds_read, s_waitcnt, v_mfma per step), so each of the 8*KB steps eats the LDS latency.
                // Here the K fragment of step j+2 is issued before MFMA j and sched_barrier(0) pins that order; the
                // compiler still inserts the (counted) waits.
                constexpr int NSTEP = KB * NKS;
                auto kfrag = [&](int j) {
                    return *reinterpret_cast<const s16x8*>(Kt + TileSwz<D>::off(32 * (j / NKS) + r, 2 * (j % NKS) + h));
                };
                s16x8 ring[3];
                ring[0] = kfrag(0);
                ring[1] = kfrag(1);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
                for (int j = 0; j < NSTEP; ++j) {
                    if (j + 2 < NSTEP) ring[(j + 2) % 3] = kfrag(j + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    sacc[j / NKS] = mfma32<Tag>(ring[j % 3], qf[j % NKS], sacc[j / NKS]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- mask (diagonal tiles / ragged last tile only)
            const bool need_mask = (CAUSAL && (k0 + BN - 1 > q0 + 32 * w)) || (k0 + BN > n);
            if (need_mask) {
                // key index of register i is k0 + 32 kb + 4 h + rc(i), rc(i) = (i & 3) + 8 (i >> 2): compare the
                // compile-time rc(i) with one per-lane threshold instead of materialising every key index
                const int lim = CAUSAL ? min(qrow, n - 1) : n - 1;   // last visible key of this lane's row
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int thr = lim - (k0 + 32 * kb + 4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if ((i & 3) + 8 * (i >> 2) > thr) sacc[kb][i] = -INFINITY;
                }
            }
            // ---- online softmax for query row `qrow` (per lane; the other 32 keys live in lane ^ 32)
            float mx = sacc[0][0];
            if (!(ABL & 1)) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kb][i]);
            mx = fmaxf(mx, wave_half_swap(mx));
            }
            // Online-softmax bookkeeping.  LAZY (default): keep the stale running max while no row of the wave has grown
            // past it by more than 2^8 (P then lies in (0, 256] instead of (0, 1]: the same relative precision in bf16 /
            // f16 / f32, no overflow), and skip the O / l rescale for that tile — it is needed in the first tiles only.
            // When any row does cross the threshold, EVERY lane moves to its exact new max and everything accumulated
            // so far (O and l; P of this tile is not formed yet) is scaled exactly once.
            const float m_new = fmaxf(m_run, mx);
            float mc;
            bool rescale = true;
            if (LAZY) rescale = __any((m_new - m_run) * c_log2 > 8.0f) != 0;   // m_run = -inf on the first tile -> true
            if (rescale) {
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * c_log2);
                mc = m_use * c_log2;
                m_run = m_new;
                if constexpr (D == 256) {   // accumulators live in AGPRs: scale them in place (fa_common.h)
                    acc_scale_begin();
#pragma unroll
                    for (int t2 = 0; t2 < NDV; ++t2) acc_scale(oacc[t2], alpha);
                    acc_scale_end();
                } else {
#pragma unroll
                    for (int t2 = 0; t2 < NDV; ++t2)
#pragma unroll
                        for (int i = 0; i < 16; ++i) oacc[t2][i] *= alpha;
                }
                if (RS_MFMA) lacc[0] *= alpha;   // every register holds the same sum; only register 0 is read back
                l_run *= alpha;
            } else {
                mc = m_run * c_log2;
            }
            // then per 32-key block: exp2 -> pack -> issue that block's P.V MFMAs (they execute asynchronously, so the
            // next block's exp2 / sum / pack runs underneath them)
            float rs = 0.f;
            if constexpr (!HS) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = (ABL & 1) ? sacc[kb][i] : __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], c_log2, -mc));
                    sacc[kb][i] = p;
                    if (!RS_MFMA && !(ABL & 1)) rs += p;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 pk;
                    pk[0] = pack2<Tag>(sacc[kb][8 * s + 0], sacc[kb][8 * s + 1]);
                    pk[1] = pack2<Tag>(sacc[kb][8 * s + 2], sacc[kb][8 * s + 3]);
                    pk[2] = pack2<Tag>(sacc[kb][8 * s + 4], sacc[kb][8 * s + 5]);
                    pk[3] = pack2<Tag>(sacc[kb][8 * s + 6], sacc[kb][8 * s + 7]);
                    const s16x8 pb = *reinterpret_cast<s16x8*>(&pk);
                    const int key_a = 32 * kb + 16 * s + 4 * h + tq;  // rows of the first 4-row block; second is +8
                    if (RS_MFMA) lacc = mfma32<Tag>(ones, pb, lacc);
#pragma unroll
                    for (int dvb = 0; dvb < NDV; ++dvb) {
                        const int ch = 4 * dvb + 2 * g16 + (tp >> 1);
                        const s16x4 lo = lds_tr16(Vt + TileSwz<D>::off(key_a, ch) + 8 * (tp & 1));
                        const s16x4 hi = lds_tr16(Vt + TileSwz<D>::off(key_a + 8, ch) + 8 * (tp & 1));
                        const s16x8 a = (ABL & 4) ? qf[dvb] : cat8(lo, hi);
                        oacc[dvb] = mfma32<Tag>(a, pb, oacc[dvb]);
                    }
                }
            }
            } else {
                // exp2 / pack of block kb+1 is sliced between the P.V MFMAs of block kb (two elements = one packed dword
                // per MFMA step), and the transposed V fragment of step t+1 is issued before MFMA t.
                u32x4 pp[2][2];   // packed P of the block being consumed / being produced (static indices after unrolling)
                auto exp_pair = [&](int kb, int m) {   // elements 2m, 2m+1 of block kb -> dword (m >> 2, m & 3)
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[kb][2 * m], c_log2, -mc));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[kb][2 * m + 1], c_log2, -mc));
                    if (!RS_MFMA) rs += p0 + p1;
                    pp[kb & 1][m >> 2][m & 3] = pack2<Tag>(p0, p1);
                };
                auto vfrag = [&](int kb, int t2) {     // A operand of step (s = t2 >> 2 ... ) of block kb
                    const int s2 = t2 / NDV, dvb = t2 % NDV;
                    const int key_a = 32 * kb + 16 * s2 + 4 * h + tq;
                    const int ch = 4 * dvb + 2 * g16 + (tp >> 1);
                    return cat8(lds_tr16(Vt + TileSwz<D>::off(key_a, ch) + 8 * (tp & 1)),
                                lds_tr16(Vt + TileSwz<D>::off(key_a + 8, ch) + 8 * (tp & 1)));
                };
                constexpr int PSTEP = 2 * NDV;          // MFMAs per 32-key block
#pragma unroll
                for (int m = 0; m < 8; ++m) exp_pair(0, m);
                s16x8 vnext = vfrag(0, 0);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                    for (int t2 = 0; t2 < PSTEP; ++t2) {
                        const s16x8 vcur = vnext;
                        if (t2 + 1 < PSTEP) vnext = vfrag(kb, t2 + 1);
                        else if (kb + 1 < KB) vnext = vfrag(kb + 1, 0);
                        if (kb + 1 < KB) {                               // 8 pairs of block kb+1 over the PSTEP steps
#pragma unroll
                            for (int m = (8 * t2) / PSTEP; m < (8 * (t2 + 1)) / PSTEP; ++m) exp_pair(kb + 1, m);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const s16x8 pb = *reinterpret_cast<s16x8*>(&pp[kb & 1][t2 / NDV]);
                        if (RS_MFMA && (t2 % NDV) == 0) lacc = mfma32<Tag>(ones, pb, lacc);
                        oacc[t2 % NDV] = mfma32<Tag>(vcur, pb, oacc[t2 % NDV]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            l_run += rs;
        }
        if (!(ABL & 2)) {
        if (!(ABL & 16)) dma_wait_all();   // this wave's share of the next tile has landed ...
        __syncthreads();  // ... and so has everyone else's
        }
    }
    // causal: this wave's rows end before the workgroup's last tiles; keep feeding the other waves' tiles
    for (int t = ntiles_w; t < ntiles; ++t) {
        stage_next(t);
        dma_wait_all();
        __syncthreads();
    }
    const int lastbuf = (gbase + ntiles - 1) & 1;   // buffer of the tile just finished: dead now, the epilogue's staging area
    gbase += ntiles;

    // ---- epilogue: normalise, store O (input dtype) and lse (fp32, natural log)
    const float l_tot = RS_MFMA ? lacc[0] : l_run + wave_half_swap(l_run);
    if (has_next) load_q(tile_of(it + 1));   // Q of the next tile is in flight while this tile's O goes out
    {
        const float inv = 1.f / l_tot;
        u32x2 vals[NDV * 4];
#pragma unroll
        for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vals[4 * dvb + g][0] = pack2_rn<Tag>(oacc[dvb][4 * g + 0] * inv, oacc[dvb][4 * g + 1] * inv);
                vals[4 * dvb + g][1] = pack2_rn<Tag>(oacc[dvb][4 * g + 2] * inv, oacc[dvb][4 * g + 3] * inv);
            }
        // every wave is past the last barrier: that tile's buffer is dead, each wave takes 32 x D x 2 bytes of it
        // (with TPW == 1 buffer 0 is used: it always holds at least 8 x 32 x D x 2 bytes, see the launcher)
        char* stg = smem + (TPW == 1 ? 0 : lastbuf * 2 * TILE_BYTES) + w * 32 * D * 2;
        store_rows_via_lds<D>(stg, vals, o + base, q0 + 32 * w, n, lane, DR);
        if (qrow < n && h == 0) lse[(size_t)bh * n + qrow] = m_run * scale + logf(l_tot);
    }
    if (has_next) __syncthreads();   // the staging area is the next tile's first DMA target
    }   // query tiles of this workgroup
}

bool fwd_mfma_supported(int dtype, int64_t d) { return (dtype == 1 || dtype == 2) && d >= 8 && d <= 256 && d % 8 == 0; }

// ------------------------------------------------------------------------------------------------
// Debug trace (tools/trace_stag.py): when a buffer is registered, waves 0 and 4 of workgroup 0 of the staggered kernel
// store the shader clock after every phase body and after every barrier.  [wave half][event] int64; event 0 = count.
__device__ long long* g_trace = nullptr;
hipError_t set_trace_buffer(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)); }

// Staggered ("ping-pong") forward: the DEFAULT at d = 128 with 128-key tiles (1.95 ms against 2.14 for the lock-step kernel
// above at B8 H32 N4096, bitwise the same results; profiles/r01_tile_sweep.md); also built with 64-key tiles (fwd_stag = 1).  The two waves that share a SIMD run the same program; with
// one barrier per tile they stay in lock step, so their MFMA phases collide and their softmax (VALU) phases collide, and
// the tile time is close to the SUM of the two.  Here every 64-key tile is split in two halves separated by barriers,
//     M_t = [ O^T += V^T P^T (tile t-1) ; S^T = K Q^T (tile t) ]      matrix pipe
//     V_t = [ online softmax of tile t, O rescale, pack P ]           vector pipe
// and waves 4..7 run half a tile behind waves 0..3, so at any time one wave of a SIMD is in M and the other in V.
// A wave that is alone on the matrix pipe gets no latency cover from its partner, so the M phase is hand timed: LDS
// operand reads from inline asm three MFMAs ahead (a 4-slot ring, the first slots filled before the barrier), counted
// lgkmcnt waits fused to each MFMA, P.V and S accumulation chains interleaved.  K and V are triple buffered with the
// LDS-DMA issued two tiles ahead at even global half-steps g = 2u (K(u+2), V(u+1)) and counted vmcnt waits, so a
// transfer has four half-steps to land.
// W2: one counted operand wait per TWO MFMAs (an even step also waits for the next step's operand, an odd step issues none), as
// the backward stream kernels do: one instruction less per two steps on an issue port that is 80 % busy (option fwd_w2).
template <typename Tag, int D, bool CAUSAL, int KB, bool PAD = false, int RD = 4, bool W2 = false>
__global__ __launch_bounds__(512, 2) void fwd_mfma_stag_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                               const uint16_t* __restrict__ v, uint16_t* __restrict__ o,
                                                               float* __restrict__ lse, int n, int nqt, float c_log2,
                                                               float scale, int dr_dbg /* row length | debug ablation flags (option fwd_abl) << 16 */,
                                                               int nk /* keys; n = query rows; causal needs nk >= n: the diagonal sits at key = row + nk - n */) {
    const int dbg = dr_dbg >> 16;
    const int DR = PAD ? (dr_dbg & 0xffff) : D;   // elements per tensor row (PAD: head dims below the tile width, fa_common.h)
    constexpr int BM = 256, BN = 32 * KB, NKS = D / 16, NDV = D / 32;
    constexpr int TILE_BYTES = BN * D * 2;
    // 64-key tiles: K and V triple buffered, DMA two tiles ahead.  128-key tiles (160 KB of LDS do not hold six of them):
    // double buffered, DMA one tile ahead, which is as much time because the phases are twice as long.
    constexpr int NBUF = KB == 4 ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [K x NBUF | V x NBUF]: tile t in buffer t % NBUF
    char* Kbuf = smem;
    char* Vbuf = smem + NBUF * TILE_BYTES;

    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = L / nqt;
    int qt = L - bh * nqt;
    if (CAUSAL) qt = nqt - 1 - qt;
    const int q0 = qt * BM;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int qrow = q0 + 32 * w + r;
    const size_t base = (size_t)bh * n * DR, kbase = (size_t)bh * nk * DR;
    const int coff = nk - n;
    // waves 4..7 (the second wave of every SIMD) run one half-step behind (debug flag 64: waves 0..3 lag instead — does the longer
    // matrix phase of the lagging half follow the wave's age or its role?)
    const int stag = ((dr_dbg >> 16) & 64) ? 1 - (w >> 2) : (w >> 2);

    const buf_rsrc_t q_rs = make_rsrc(q + base, (unsigned)n * DR * 2);
    s16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = buf_load_frag(q_rs, frag_off(qrow, 16 * ks + 8 * h, DR, PAD));

    const int kend = CAUSAL ? min(nk, q0 + BM + coff) : nk;
    const int T = (kend + BN - 1) / BN;                                               // tiles of the workgroup
    const int Tw = CAUSAL ? min(T, (q0 + 32 * w + 31 + coff) / BN + 1) : T;                  // tiles this wave computes

    const rsrc_s_t k_rs = make_rsrc_s(k + kbase, (unsigned)nk * DR * 2);
    const rsrc_s_t v_rs = make_rsrc_s(v + kbase, (unsigned)nk * DR * 2);
    const int dma_voff = dma_lane_voff<D>(lane, w, DR);
    // Issued at the start of global half-step 2u, first read in half-step 2u+4: K(u+2) and V(u+1).  Two tiles of
    // flight time (a 64-key tile is consumed in about a microsecond, less than one trip to L2 / HBM under load).
    // Always issued, so every wave has the same number of DMAs per step and the waits can be counted; tiles past the
    // end of the tensor cost nothing (the range check answers with zeros).
    constexpr int DMA_PER_ISSUE = 2 * (BN / (512 / D)) / 8;
    auto issue = [&](int u) {
        if (dbg & 4) return;                                   // ablation: no DMA
        if (NBUF == 3) {
            dma_stage_tile<D, BN, 8>(k_rs, Kbuf + ((u + 2) % 3) * TILE_BYTES, (u + 2) * BN, dma_voff, w, DR);
            dma_stage_tile<D, BN, 8>(v_rs, Vbuf + ((u + 1) % 3) * TILE_BYTES, (u + 1) * BN, dma_voff, w, DR);
        } else {   // K(u+1), V(u): first read in half-step 2u+2
            dma_stage_tile<D, BN, 8>(k_rs, Kbuf + ((u + 1) & 1) * TILE_BYTES, (u + 1) * BN, dma_voff, w, DR);
            dma_stage_tile<D, BN, 8>(v_rs, Vbuf + (u & 1) * TILE_BYTES, u * BN, dma_voff, w, DR);
        }
    };

    f32x16 oacc[NDV];
#pragma unroll
    for (int t = 0; t < NDV; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 sacc[KB];
    u32x4 pp[KB][2];
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;

    auto do_softmax = [&](int t) {
        if (dbg & 8) return;                                   // ablation: no V phase at all
        if (dbg & 32) {                                        // ablation: a plain exp2+fma chain of the same length, then pack
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[kb][i] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], 0.5f, -1.0f));
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[kb][s][j] = pack2<Tag>(sacc[kb][8 * s + 2 * j], sacc[kb][8 * s + 2 * j + 1]);
            return;
        }
        if (dbg & 1) {                                         // ablation: pack only
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[kb][s][j] = pack2<Tag>(sacc[kb][8 * s + 2 * j], sacc[kb][8 * s + 2 * j + 1]);
            return;
        }
        const int k0 = t * BN;
        const bool need_mask = (CAUSAL && (k0 + BN - 1 > q0 + 32 * w + coff)) || (k0 + BN > nk);
        if (need_mask) {
            const int lim = CAUSAL ? min(qrow + coff, nk - 1) : nk - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int thr = lim - (k0 + 32 * kb + 4 * h);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((i & 3) + 8 * (i >> 2) > thr) sacc[kb][i] = -INFINITY;
            }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kb][i]);
        mx = fmaxf(mx, wave_half_swap(mx));
        const float m_new = fmaxf(m_run, mx);
        float mc;
        // lazy rescale, as in the lock-step kernel: O and l move to the new max only when some row grew by > 2^8
        if (__any((m_new - m_run) * c_log2 > 8.0f) != 0) {
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * c_log2);
            mc = m_use * c_log2;
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int t2 = 0; t2 < NDV; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[t2][i] *= alpha;
        } else {
            mc = m_run * c_log2;
        }
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], c_log2, -mc));
                sacc[kb][i] = p;
                rs += p;
            }
        l_run += rs;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) pp[kb][s][j] = pack2<Tag>(sacc[kb][8 * s + 2 * j], sacc[kb][8 * s + 2 * j + 1]);
    };
    // lane-constant operand addresses (LDS bytes, buffer 0 of K / V); rows advance through the immediate offset: the
    // swizzle depends on the row modulo 16 only, so +32 kb (+16 s) rows is a constant number of bytes.  ONE base per operand
    // kind: the k-step / d-block enters the chunk index through disjoint bits, chunk ^ f(row) = (2 ks) ^ (h ^ f) and
    // (4 dvb) ^ (...), and the tile bases are multiples of the row pitch, so the other addresses are the base XOR a constant
    // (made per phase, one v_xor each, where the buffer offset used to be added) — 16 registers less than an address array
    // per kind, which is what pays for the deeper operand ring (RD).
    unsigned ka0, vlo0, vhi0;
    {
        const unsigned k0a = lds_addr_of(Kbuf), v0a = lds_addr_of(Vbuf);
        ka0 = k0a + TileSwz<D>::off(r, h);
        const int ch = 2 * g16 + (tp >> 1);
        vlo0 = v0a + TileSwz<D>::off(4 * h + tq, ch) + 8 * (tp & 1);
        vhi0 = v0a + TileSwz<D>::off(4 * h + tq + 8, ch) + 8 * (tp & 1);
    }
    // The phase comes in two parts: WHICH = 0 (addresses of this phase's buffers + the first RD-1 operand requests) runs
    // at the END of the preceding phase, ahead of the barrier, so the LDS latency of the first operands passes while the
    // wave waits for its partner; WHICH = 1 is the MFMA stream.
    // RD ring slots: operands are requested RD - 1 MFMAs ahead
    s16x8 ring[RD];
    unsigned kq[NKS], vl[NDV], vh[NDV];
    auto do_M = [&](auto has_pv, auto has_s, auto which, int t) {
        constexpr int WHICH = decltype(which)::value;
        constexpr bool PV = decltype(has_pv)::value, SS = decltype(has_s)::value;
        constexpr int NPV = PV ? KB * 2 * NDV : 0, NS = SS ? KB * NKS : 0, NSTEP = NPV + NS;
        constexpr int ROWB = 2 * D;
        // Step order: a lone wave needs several INDEPENDENT accumulation chains in flight — back-to-back MFMAs into one
        // accumulator run at about half rate (tools/ubench/overlap.hip) — so P.V steps (NDV chains) and S steps (KB
        // chains) alternate, and consecutive S steps alternate between the key blocks.
        // step j -> (is_pv, index): P.V index p = (kb, s2, dvb) with dvb fastest; S index i = (ks, kb) with kb fastest.
        struct Map {
            static constexpr bool is_pv(int j) { return NS == 0 || (NPV != 0 && (j < 2 * (NPV < NS ? NPV : NS) ? (j & 1) == 0 : NPV > NS)); }
            static constexpr int idx(int j) {
                constexpr int m = NPV < NS ? NPV : NS;
                return (NPV == 0 || NS == 0) ? j : (j < 2 * m ? j / 2 : j - m);
            }
        };
        // this phase's buffers folded into the lane addresses once; everything else is an immediate offset
        if constexpr (WHICH == 0) {
            const unsigned vsel = (t % NBUF) * TILE_BYTES, ksel = ((t + (PV ? 1 : 0)) % NBUF) * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) kq[ks] = (ka0 + ksel) ^ (32u * ks);
#pragma unroll
            for (int dvb = 0; dvb < NDV; ++dvb) { vl[dvb] = (vlo0 + vsel) ^ (64u * dvb); vh[dvb] = (vhi0 + vsel) ^ (64u * dvb); }
        }
        auto fetch = [&](auto jc) {   // operand fragment of step j into its ring slot
            constexpr int j = decltype(jc)::value;
            if constexpr (j < NSTEP) {
                constexpr int x = Map::idx(j);
                if constexpr (Map::is_pv(j)) {
                    constexpr int kb = x / (2 * NDV), s2 = (x / NDV) & 1, dvb = x % NDV, off = (32 * kb + 16 * s2) * ROWB;
                    ring[j % RD] = cat8(lds_tr16_asm<off>(vl[dvb]), lds_tr16_asm<off>(vh[dvb]));
                } else {
                    constexpr int kb = x % KB, ks = x / KB;
                    ring[j % RD] = lds_b128_asm<32 * kb * ROWB>(kq[ks]);
                }
            }
        };
        auto nops = [](int j) constexpr { return j >= NSTEP ? 0 : (Map::is_pv(j) ? 2 : 1); };   // LDS instructions of step j
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = Map::idx(j);
            fetch(std::integral_constant<int, j + RD - 1>{});
            __builtin_amdgcn_sched_barrier(0);
            // LDS instructions issued after the operand this step waits for: its own, or (W2, even steps) the next step's;
            // W2's odd steps issue no wait: the even step before them has covered their operand
            constexpr bool waits = !W2 || (j % 2) == 0;
            constexpr int nw = [&]() constexpr { int c = 0; for (int q = (W2 ? 2 : 1); q < RD; ++q) c += nops(j + q); return c; }();
            static_assert(!W2 || NSTEP % 2 == 0, "W2 pairs the steps");
            if constexpr (Map::is_pv(j)) {
                constexpr int kb = x / (2 * NDV), s2 = (x / NDV) & 1, dvb = x % NDV;
                if constexpr (waits) MfmaWait<Tag, nw>::acc(ring[j % RD], *reinterpret_cast<s16x8*>(&pp[kb][s2]), oacc[dvb]);
                else MfmaNoWait<Tag>::acc(ring[j % RD], *reinterpret_cast<s16x8*>(&pp[kb][s2]), oacc[dvb]);
            } else {
                constexpr int kb = x % KB, ks = x / KB;
                if constexpr (ks == 0) {
                    if constexpr (waits) MfmaWait<Tag, nw>::first(ring[j % RD], qf[ks], sacc[kb]);
                    else MfmaNoWait<Tag>::first(ring[j % RD], qf[ks], sacc[kb]);
                } else {
                    if constexpr (waits) MfmaWait<Tag, nw>::acc(ring[j % RD], qf[ks], sacc[kb]);
                    else MfmaNoWait<Tag>::acc(ring[j % RD], qf[ks], sacc[kb]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (WHICH == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing older may sit in the LDS queue: the counts are exact
            for_each_const(fetch, std::make_integer_sequence<int, RD - 1>{});
        } else {
            // the matrix-phase wave outranks its partner's vector phase (option fwd_hs = 1 turns this off for the A/B)
            if (!(dbg & (16 | 256))) __builtin_amdgcn_s_setprio(2);
            for_each_const(step, std::make_integer_sequence<int, NSTEP>{});
            mfma_stream_fence(sacc, oacc);   // the vector phase reads sacc / oacc by VALU right after the barrier
            if (!(dbg & (16 | 256))) __builtin_amdgcn_s_setprio(0);
        }
    };

    // K(0), K(1), V(0), then global half-steps g = 0 .. 2T+1, one barrier after each; a wave's local step is g - stag.
    // What half-step 2u+2 reads was issued at 2u-2 or earlier: at the barrier that ends the ODD half-step 2u+1 each
    // wave waits until only its newest issue (that of 2u) is still in flight.
    long long* const trace = g_trace;
    const bool tron = trace != nullptr && blockIdx.x == 0 && (w == 0 || w == 4);
    int tri = 1;
    auto stamp = [&]() {
        if (tron) {
            // inline asm: hipcc must not see a scalar-memory read in flight, or every LDS wait nearby becomes lgkmcnt(0)
            unsigned long long c;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c)::"memory");
            if (lane == 0 && tri < 2040) trace[(w >> 2) * 2048 + tri] = (long long)c;   // (indexed by wave, not by role)
            ++tri;
        }
    };
    auto end_half = [&](int gg) {
        stamp();
        if (gg & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF == 3 ? DMA_PER_ISSUE : 0) : "memory");
        __syncthreads();
        stamp();
    };
    if ((dbg & 256) && stag) __builtin_amdgcn_s_setprio(1);   // A/B: one static priority for the younger half, no per-phase flips (guide, two waves per SIMD, item 4)
    dma_stage_tile<D, BN, 8>(k_rs, Kbuf, 0, dma_voff, w, DR);
    if (NBUF == 3) {
        dma_stage_tile<D, BN, 8>(k_rs, Kbuf + TILE_BYTES, BN, dma_voff, w, DR);
        dma_stage_tile<D, BN, 8>(v_rs, Vbuf, 0, dma_voff, w, DR);
    }
    dma_wait_all();
    __syncthreads();
    int g = 0;
    if (stag) { issue(0); end_half(0); g = 1; }               // second half idles through half-step 0
    if (!(g & 1)) issue(g >> 1);
    do_M(std::false_type{}, std::true_type{}, std::integral_constant<int, 0>{}, 0);
    do_M(std::false_type{}, std::true_type{}, std::integral_constant<int, 1>{}, 0);             // M_0 = S(0)
    end_half(g); ++g;
    for (int t = 0; t < Tw; ++t) {
        if (!(g & 1)) issue(g >> 1);
        do_softmax(t);                                        // V_t
        // first operands of M_{t+1} ahead of the barrier — with three buffers both tiles landed a step ago; with two the
        // tiles are only guaranteed after this barrier, so the requests follow it
        if (NBUF == 3) do_M(std::true_type{}, std::true_type{}, std::integral_constant<int, 0>{}, t);
        end_half(g); ++g;
        if (!(g & 1)) issue(g >> 1);
        if (NBUF != 3) do_M(std::true_type{}, std::true_type{}, std::integral_constant<int, 0>{}, t);
        do_M(std::true_type{}, std::true_type{}, std::integral_constant<int, 1>{}, t);          // M_{t+1} = P.V(t) ; S(t+1)  (after the last tile S is unused:
                                                              // one code path keeps the accumulators in place)
        end_half(g); ++g;
    }
    for (; g < 2 * T + 2; ++g) {                              // feed-only half-steps (causal tail, idle half)
        if (!(g & 1)) issue(g >> 1);
        end_half(g);
    }
    dma_wait_all();   // nothing of this workgroup may still be writing LDS when the next one takes the CU
    if (tron && lane == 0) trace[(w >> 2) * 2048] = tri;

    // ---- epilogue.  Every wave is past the last barrier and nothing writes LDS any more: the K / V buffers are dead and each
    // wave takes 32 x D x 2 bytes of them to turn its row-on-the-lane tile into whole-row stores (1 KiB contiguous per store
    // instruction instead of sixteen 8-byte pieces per lane at a row stride: the per-workgroup store tail is issue bound,
    // guide T21) — the same staging, and so the same bytes, as the lock-step kernel's epilogue.
    const float l_tot = l_run + wave_half_swap(l_run);
    if (dbg & 128) {   // A/B: the direct stores of rounds 1 - 2 (8 bytes per lane and instruction)
        if (qrow < n) {
            const float inv = 1.f / l_tot;
            uint16_t* orow = o + base + (size_t)qrow * DR;
#pragma unroll
            for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    u32x2 pk;
                    pk[0] = pack2_rn<Tag>(oacc[dvb][4 * gq + 0] * inv, oacc[dvb][4 * gq + 1] * inv);
                    pk[1] = pack2_rn<Tag>(oacc[dvb][4 * gq + 2] * inv, oacc[dvb][4 * gq + 3] * inv);
                    if (PAD && 32 * dvb + 8 * gq + 4 * h >= DR) continue;
                    *reinterpret_cast<u32x2*>(orow + 32 * dvb + 8 * gq + 4 * h) = pk;
                }
            if (h == 0) lse[(size_t)bh * n + qrow] = m_run * scale + logf(l_tot);
        }
        return;
    }
    {
        const float inv = 1.f / l_tot;
        u32x2 vals[NDV * 4];
#pragma unroll
        for (int dvb = 0; dvb < NDV; ++dvb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                vals[4 * dvb + gq][0] = pack2_rn<Tag>(oacc[dvb][4 * gq + 0] * inv, oacc[dvb][4 * gq + 1] * inv);
                vals[4 * dvb + gq][1] = pack2_rn<Tag>(oacc[dvb][4 * gq + 2] * inv, oacc[dvb][4 * gq + 3] * inv);
            }
        store_rows_via_lds<D>(smem + w * 32 * D * 2, vals, o + base, q0 + 32 * w, n, lane, DR);
        if (qrow < n && h == 0) lse[(size_t)bh * n + qrow] = m_run * scale + logf(l_tot);
    }
}

// K/V tile size in 32-key blocks: 4 (128 keys) is the measured winner (profiles/r01_tile_sweep.md);
// FA_FWD_KB=1|2|4 overrides it for the sweep (1 only at d = 128).
static int fwd_kb_override() { return option(OPT_FWD_KB); }

template <typename Tag, int D, int KB, bool PAD = false>
static hipError_t launch_fwd_t(const FwdArgs& a, hipStream_t st, bool want_stag = false) {
    constexpr int NW = D == 256 ? 4 : 8, BM = 32 * NW;
    const int nqt = (int)((a.n + BM - 1) / BM);
    size_t smem = 2 * 2 * (32 * KB) * D * 2;
    if (smem < (size_t)NW * 32 * D * 2) smem = (size_t)NW * 32 * D * 2;   // the epilogue stages the BM x D output tile in LDS
    const float c = a.scale * 1.4426950408889634f;
    dim3 grid((unsigned)(nqt * a.bh));
    int last_arg = (int)a.d;   // row length; the staggered kernel takes debug ablation flags in this slot instead
    ProfScope ps(K_FWD_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k,
                           (const uint16_t*)a.v, (uint16_t*)a.o, a.lse, (int)a.n, nqt, c, a.scale, last_arg);
        return hipGetLastError();
    };
    // staggered schedule (fwd_mfma_stag_kernel): d = 128, 64-key tiles (three buffers each) or 128-key tiles (two)
    if constexpr ((D == 128 || D == 64) && (KB == 2 || KB == 4)) {
        if (want_stag) {
            smem = (size_t)2 * (KB == 4 ? 2 : 3) * (32 * KB) * D * 2;
            last_arg = (int)a.d | ((option(OPT_FWD_STAG) != 0 ? option(OPT_FWD_ABL) : 0) << 16);   // debug flags only with an explicit fwd_stag
            auto launch_s = [&](auto kern) -> hipError_t {
                hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, grid, dim3(64 * NW), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k,
                                   (const uint16_t*)a.v, (uint16_t*)a.o, a.lse, (int)a.n, nqt, c, a.scale, last_arg, (int)(a.nk > 0 ? a.nk : a.n));
                return hipGetLastError();
            };
            if constexpr (D == 128 && KB == 4 && !PAD) {   // operand ring depth (option fwd_rd: 4, 6, 8; 0 = default)
                const int rd = option(OPT_FWD_RD);
                if (option(OPT_FWD_W2) == 1) return a.causal ? launch_s(fwd_mfma_stag_kernel<Tag, D, true, KB, PAD, 6, true>) : launch_s(fwd_mfma_stag_kernel<Tag, D, false, KB, PAD, 6, true>);
                if (rd == 6) return a.causal ? launch_s(fwd_mfma_stag_kernel<Tag, D, true, KB, PAD, 6>) : launch_s(fwd_mfma_stag_kernel<Tag, D, false, KB, PAD, 6>);
                if (rd == 8) return a.causal ? launch_s(fwd_mfma_stag_kernel<Tag, D, true, KB, PAD, 8>) : launch_s(fwd_mfma_stag_kernel<Tag, D, false, KB, PAD, 8>);
            }
            return a.causal ? launch_s(fwd_mfma_stag_kernel<Tag, D, true, KB, PAD>) : launch_s(fwd_mfma_stag_kernel<Tag, D, false, KB, PAD>);
        }
    }
    if constexpr (!PAD && D != 256) {   // sweep variants exist for the 64 / 128 tile widths only
    if (option(OPT_FWD_RS) != 0)
        return a.causal ? launch(fwd_mfma_kernel<Tag, D, true, KB, true, true, false, 1, PAD>) : launch(fwd_mfma_kernel<Tag, D, false, KB, true, true, false, 1, PAD>);
    if (option(OPT_FWD_EAGER) != 0)   // rescale every tile (the textbook order), for the A/B
        return a.causal ? launch(fwd_mfma_kernel<Tag, D, true, KB, false, false, false, 1, PAD>) : launch(fwd_mfma_kernel<Tag, D, false, KB, false, false, false, 1, PAD>);
    if (option(OPT_FWD_HS) != 0)
        return a.causal ? launch(fwd_mfma_kernel<Tag, D, true, KB, false, true, true, 1, PAD>) : launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, true, 1, PAD>);
    }
    if constexpr (KB == 4 && D == 128 && !PAD && std::is_same<Tag, bf16_tag>::value) {
        if (!a.causal) switch (option(OPT_FWD_ABL)) {   // profiling ablations: see the kernel's header comment
            case 1: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 1>);
            case 2: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 2>);
            case 3: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 3>);
            case 4: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 4>);
            case 6: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 6>);
            case 7: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 7>);
            case 8: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 8>);
            case 16: return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD, 16>);
            default: break;
        }
    }
    if constexpr (KB == 4) {
        // query tiles per workgroup: measured winners (profiles/r01_tile_sweep.md) are 2 under the causal mask (heavy +
        // light tile: equal work per workgroup) and at d = 64 (-13 %), 1 at d = 128 non-causal; option fwd_tpw overrides
        int tpw = option(OPT_FWD_TPW);
        if (tpw == 0) tpw = (a.causal || D == 64) ? 2 : 1;
        if (D == 256) tpw = 1;
        if (tpw == 2) {
            grid = dim3((unsigned)(((nqt + 1) / 2) * a.bh));
            return a.causal ? launch(fwd_mfma_kernel<Tag, D, true, KB, false, true, false, 2, PAD>) : launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 2, PAD>);
        }
        if constexpr (!PAD) {
            if (tpw == 4 && !a.causal) {
                grid = dim3((unsigned)(((nqt + 3) / 4) * a.bh));
                return launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 4, PAD>);
            }
        }
    }
    return a.causal ? launch(fwd_mfma_kernel<Tag, D, true, KB, false, true, false, 1, PAD>) : launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, PAD>);
}

// Small launches: 128-row tiles on 4 waves (W4)
template <typename Tag, int D>
static hipError_t launch_fwd_w4(const FwdArgs& a, hipStream_t st) {
    constexpr int NW = 4, BM = 32 * NW, KB = 4;
    const int nqt = (int)((a.n + BM - 1) / BM);
    const size_t smem = 2 * 2 * (32 * KB) * D * 2;
    const float c = a.scale * 1.4426950408889634f;
    ProfScope ps(K_FWD_MFMA, st);
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = ensure_dynamic_smem(reinterpret_cast<const void*>(kern), (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nqt * a.bh)), dim3(64 * NW), smem, st, (const uint16_t*)a.q, (const uint16_t*)a.k,
                           (const uint16_t*)a.v, (uint16_t*)a.o, a.lse, (int)a.n, nqt, c, a.scale, (int)a.d);
        return hipGetLastError();
    };
    return a.causal ? launch(fwd_mfma_kernel<Tag, D, true, KB, false, true, false, 1, false, 0, true>)
                    : launch(fwd_mfma_kernel<Tag, D, false, KB, false, true, false, 1, false, 0, true>);
}

// Is the launch better served by the 4-wave / 128-row kernels than by 256-row tiles?  (option small_grid: 1 = never, 2 = always.)
// Measured at d = 128, bh x N from 12 x 2048 to 24 x 8192 (profiles/r02_small_launches.md): the forward while its 128-row
// workgroups still fit the chip in ONE round (<= 128 tiles of 256 rows: 0.051 vs 0.058 ms at 16 x 2048; with 160 tiles the second
// round costs 0.099 vs 0.060); the causal backward up to 256 tiles (0.550 vs 0.612 ms at 8 x 8192; 0.887 vs 0.702 at 12 x 8192).
// The non-causal backward does not ask: the stream kernels win at every size there (0.264 vs 0.388 ms at 8 x 4096).
// 64-wide tiles (d <= 64): the forward's 4-wave kernel stays ahead up to 256 tiles (0.043 vs 0.068 ms at 32 x 2048).
bool small_grid(int64_t bh, int64_t n, bool backward, int64_t d) {
    const int o = option(OPT_SMALL_GRID);
    if (o == 1) return false;
    if (o == 2) return true;
    return bh * ((n + 255) / 256) <= ((backward || d <= 64) ? 256 : 128);
}

template <typename Tag, int D>
static hipError_t launch_fwd_kb(const FwdArgs& a, hipStream_t st) {
    const int kb = fwd_kb_override();
    // Schedule at d = 128: the staggered kernel with 128-key tiles is the default (-9 % non-causal, -2 ... -10 % causal
    // against lock step, profiles/r01_tile_sweep.md; bitwise the same results, it performs the same operations in the
    // same order).  Option fwd_stag: 1 = staggered with 64-key tiles, 2 = lock step, 3 = staggered with 128-key tiles.
    const int so = option(OPT_FWD_STAG);
    const bool other_sweep = kb || option(OPT_FWD_RS) || option(OPT_FWD_EAGER) || option(OPT_FWD_HS) || option(OPT_FWD_TPW) || option(OPT_FWD_ABL);
    // Under the causal mask the staggered kernel (one tile per workgroup, heaviest first) pays only on long rows: the lock-step
    // kernel with its heavy + light tile pairs is ahead by 15 - 27 % at N = 1024, 5 - 25 % at N = 2048, 3 - 10 % at N = 4096 below
    // about 3000 row tiles, and behind by 3 - 6 % from there on (256 x 4096, N >= 8192: profiles/r02_small_launches.md §5).
    // Without the mask the two are level at N = 1024 and the lock-step kernel is 8 - 10 % ahead at N = 512.
    const bool short_rows = a.causal ? !(a.n >= 8192 || (a.n >= 4096 && a.bh * ((a.n + 255) / 256) >= 3072)) : a.n < 1024;
    if ((D == 128 && (so == 3 || (so == 0 && !other_sweep && !short_rows))) || (D == 64 && so == 3)) return launch_fwd_t<Tag, D, 4>(a, st, true);
    const bool stag = D == 128 && so == 1;
    if (stag) return launch_fwd_t<Tag, D, 2>(a, st, true);
    if (kb == 1 && D == 128) return launch_fwd_t<Tag, D, (D == 128 ? 1 : 2)>(a, st);
    if (kb == 2) return launch_fwd_t<Tag, D, 2>(a, st);
    // 64-wide tiles without the mask: 64-key K/V tiles, one query tile per workgroup (re-measured in round 2: 4 - 9 % ahead of
    // 128-key tiles with two query tiles per workgroup, 128 x 2048 ... 64 x 8192; under the mask the pairing of heavy and light tiles
    // — which needs the 128-key form — stays ahead except on the longest launches)
    if (D == 64 && kb == 0 && !a.causal && !other_sweep) return launch_fwd_t<Tag, D, 2>(a, st);
    return launch_fwd_t<Tag, D, 4>(a, st);   // 128-key tiles: fewest barriers per key (LDS 128 KiB at d = 128)
}

bool nqnk_mfma_supported(int dtype, int64_t d, int64_t bh, int64_t nq, int64_t nk, int causal) {
    return (dtype == 1 || dtype == 2) && d == 128 && nq > 0 && nk > 0 && (!causal || nk >= nq) && !small_grid(bh, nq < nk ? nq : nk, false, d);
}
hipError_t launch_fwd_nqnk(const FwdArgs& a, hipStream_t st) {   // the staggered kernel, 128-key tiles, with a.nk keys
    return a.dtype == 2 ? launch_fwd_t<bf16_tag, 128, 4>(a, st, true) : launch_fwd_t<f16_tag, 128, 4>(a, st, true);
}

hipError_t launch_fwd_mfma(const FwdArgs& a, hipStream_t st) {
    if (a.d > 128) {   // 256-wide tiles of 64 keys, 4 waves (one per SIMD)
        if (a.dtype == 2) return a.d == 256 ? launch_fwd_t<bf16_tag, 256, 2, false>(a, st) : launch_fwd_t<bf16_tag, 256, 2, true>(a, st);
        return a.d == 256 ? launch_fwd_t<f16_tag, 256, 2, false>(a, st) : launch_fwd_t<f16_tag, 256, 2, true>(a, st);
    }
    if (a.d != 64 && a.d != 128) {   // head dims 8, 16, ... below the tile width: zero-padded inside the kernel
        // 128-wide tiles: the staggered kernel, as for d = 128 (fwd_stag = 2: lock step); 64-wide: lock step
        const bool stag = a.d > 64 && option(OPT_FWD_STAG) != 2 && !small_grid(a.bh, a.n, false, a.d);
        if (a.dtype == 2) return a.d > 64 ? launch_fwd_t<bf16_tag, 128, 4, true>(a, st, stag) : launch_fwd_t<bf16_tag, 64, 4, true>(a, st);
        return a.d > 64 ? launch_fwd_t<f16_tag, 128, 4, true>(a, st, stag) : launch_fwd_t<f16_tag, 64, 4, true>(a, st);
    }
    const bool sweeping = option(OPT_FWD_KB) || (option(OPT_FWD_STAG) & 1) || option(OPT_FWD_RS) || option(OPT_FWD_EAGER) || option(OPT_FWD_HS) ||
                          option(OPT_FWD_TPW) || option(OPT_FWD_ABL);
    if (!sweeping && small_grid(a.bh, a.n, false, a.d)) {
        if (a.dtype == 2) return a.d == 128 ? launch_fwd_w4<bf16_tag, 128>(a, st) : launch_fwd_w4<bf16_tag, 64>(a, st);
        return a.d == 128 ? launch_fwd_w4<f16_tag, 128>(a, st) : launch_fwd_w4<f16_tag, 64>(a, st);
    }
    if (a.dtype == 2) return a.d == 128 ? launch_fwd_kb<bf16_tag, 128>(a, st) : launch_fwd_kb<bf16_tag, 64>(a, st);
    return a.d == 128 ? launch_fwd_kb<f16_tag, 128>(a, st) : launch_fwd_kb<f16_tag, 64>(a, st);
}

}  // namespace fa
