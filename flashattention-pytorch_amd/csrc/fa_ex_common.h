// Shared by the extended-attention kernels (fa_ex.hip: exact f32; fa_ex_mfma.hip: 16-bit MFMA): the parameter block, the
// visibility rule and the counter-based dropout generator.
#pragma once
#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

struct ExParams {
    int nq, nk, d;
    int causal;            // 0 | 1 (bottom-right aligned)
    int coff;              // nk - nq
    const uint8_t* mask;   // [nq][nk] bytes or null
    long long mask_bh;     // elements between the masks of consecutive (b,h): 0 = shared
    const uint8_t* bmask;  // [nbr][nbc] bytes or null
    int br, bc, nbc;
    float p_drop, keep_scale;   // keep_scale = 1 / (1 - p)
    unsigned drop_thr;          // keep iff the element's 16 uniform bits are >= drop_thr = floor(65536 p) + 1
    unsigned nqh;               // ceil(nq / 2): row pairs per (b,h)
    unsigned long long seedmix; // seed * G + G
    float scale;
};

// Dropout generator: ONE splitmix64 value per 2 x 2 quad of (query row, key) elements, 16 uniform bits per element — the
// 64-bit mixing (two 64 x 64 multiplies) is the expensive part on a GPU, and in every kernel a lane owns either two
// neighbouring keys of one row or two neighbouring rows of one key, so a lane uses two fields of each value it makes.
//   counter = (bh * ceil(nq / 2) + row / 2) << 32 | key / 2 ;   z = splitmix64(counter + seed * G + G)
//   field   = 2 * (row & 1) + (key & 1) ;   u = (z >> 16 field) & 0xffff ;   keep iff u >= floor(65536 p) + 1
// The oracle (oracle/attention_oracle.py: dropout_keep) runs the same arithmetic on uint64, so the masks agree bit for bit.
__device__ __forceinline__ unsigned long long ex_hash(unsigned hi, unsigned lo, unsigned long long seedmix) {
    unsigned long long z = (((unsigned long long)hi << 32) | lo) + seedmix;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ bool ex_keep(const ExParams& p, int bh, int row, int key) {
    if (p.p_drop <= 0.f) return true;
    const unsigned long long z = ex_hash((unsigned)bh * p.nqh + ((unsigned)row >> 1), (unsigned)key >> 1, p.seedmix);
    const unsigned u = (unsigned)(z >> (16 * (2 * (row & 1) + (key & 1)))) & 0xffffu;
    return u >= p.drop_thr;
}
__device__ __forceinline__ bool ex_visible(const ExParams& p, int bh, int row, int key) {
    if (row >= p.nq || key >= p.nk) return false;
    if (p.causal && key > row + p.coff) return false;
    if (p.mask && p.mask[(size_t)bh * p.mask_bh + (size_t)row * p.nk + key] == 0) return false;
    if (p.bmask && p.bmask[(row / p.br) * p.nbc + key / p.bc] == 0) return false;
    return true;
}
// does the block-sparse mask leave anything of rows [r0, r1) x keys [k0, k1)?  (uniform over the workgroup)
__device__ __forceinline__ bool ex_tile_live(const ExParams& p, int r0, int r1, int k0, int k1) {
    if (!p.bmask) return true;
    for (int rb = r0 / p.br; rb <= (r1 - 1) / p.br; ++rb)
        for (int cb = k0 / p.bc; cb <= (k1 - 1) / p.bc; ++cb)
            if (p.bmask[rb * p.nbc + cb]) return true;
    return false;
}

inline ExParams make_ex_params(const ExArgs& a) {
    ExParams p;
    p.nq = (int)a.nq; p.nk = (int)a.nk; p.d = (int)a.d;
    p.causal = a.causal ? 1 : 0;
    p.coff = (int)(a.nk - a.nq);
    p.mask = a.mask; p.mask_bh = a.mask_bh_stride;
    p.bmask = a.block_mask; p.br = (int)(a.br > 0 ? a.br : 1); p.bc = (int)(a.bc > 0 ? a.bc : 1);
    p.nbc = (int)((a.nk + p.bc - 1) / p.bc);
    p.p_drop = (float)a.dropout_p;
    p.keep_scale = a.dropout_p > 0.0 ? (float)(1.0 / (1.0 - a.dropout_p)) : 1.f;
    const double t = a.dropout_p * 65536.0;
    p.drop_thr = a.dropout_p > 0.0 ? (unsigned)((long long)t + 1) : 0u;   // floor(65536 p) + 1 <= 65536
    p.nqh = (unsigned)((a.nq + 1) / 2);
    p.seedmix = a.seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
    p.scale = a.scale;
    return p;
}

}  // namespace fa
