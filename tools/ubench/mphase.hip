// Micro-benchmark: the staggered forward's matrix phase (fa_fwd_mfma.hip do_M) transplanted with synthetic operands,
// run by ONE wave per SIMD (the situation of a wave whose SIMD partner is in its vector phase), to find what makes it
// take 47 cycles per MFMA in the kernel against 35-37 in tools/ubench/overlap.hip.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I flashattention-pytorch_amd/csrc -I include tools/ubench/mphase.hip -o tools/ubench/mphase
#include "fa_common.h"
#include <cstdio>
using namespace fa;

// VARIANT bits: 8 = waves 4-7 stay resident and meet waves 0-3 at a barrier after every phase (idle partner)
// VARIANT bits: 1 = tiles high in LDS (K at 0, V at 96 KiB like a 3+3 buffer layout) instead of K at 0 / V at 16 KiB
//               2 = steps ordered P.V block then S block (no interleave)     4 = compiler-managed reads (no asm)
template <int VARIANT>
__global__ __launch_bounds__(512, 2) void kreal(float* out, int iters, int zero) {
    constexpr int D = 128, KB = 2, NKS = 8, NDV = 4, ROWB = 256, TILE = 64 * 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 6 * TILE / 4; i += 512) reinterpret_cast<int*>(smem)[i] = 0x3c003c00 + (i & 255);
    __syncthreads();
    const bool idle_half = threadIdx.x >= 256;
    if (!(VARIANT & 8) && idle_half) return;   // one wave per SIMD
    if ((VARIANT & 8) && idle_half) {
        for (int it = 0; it < iters; ++it) __syncthreads();
        return;
    }
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    char* Kbuf = smem;
    char* Vbuf = smem + ((VARIANT & 1) ? 3 * TILE : TILE);
    unsigned ka[NKS], vlo[NDV], vhi[NDV];
    const unsigned k0a = lds_addr_of(Kbuf), v0a = lds_addr_of(Vbuf);
    for (int ks = 0; ks < NKS; ++ks) ka[ks] = k0a + TileSwz<D>::off(r, 2 * ks + h);
    for (int dvb = 0; dvb < NDV; ++dvb) {
        const int ch = 4 * dvb + 2 * g16 + (tp >> 1);
        vlo[dvb] = v0a + TileSwz<D>::off(4 * h + tq, ch) + 8 * (tp & 1);
        vhi[dvb] = v0a + TileSwz<D>::off(4 * h + tq + 8, ch) + 8 * (tp & 1);
    }
    s16x8 qf[NKS];
    u32x4 pp[KB][2];
    for (int ks = 0; ks < NKS; ++ks)
        for (int i = 0; i < 8; ++i) qf[ks][i] = (short)(0x3800 + i + ks + (lane & 3));
    for (int kb = 0; kb < KB; ++kb)
        for (int s = 0; s < 2; ++s)
            for (int j = 0; j < 4; ++j) pp[kb][s][j] = 0x38003800u + kb + 2 * s + j + (lane & 1);
    f32x16 oacc[NDV], sacc[KB];
    for (int t = 0; t < NDV; ++t)
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
    for (int t = 0; t < KB; ++t)
        for (int i = 0; i < 16; ++i) sacc[t][i] = 0.f;
    unsigned long long cyc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        constexpr int NPV = KB * 2 * NDV, NS = KB * NKS, NSTEP = NPV + NS;
        struct Map {
            static constexpr bool is_pv(int j) { return (VARIANT & 2) ? j < NPV : (j & 1) == 0; }
            static constexpr int idx(int j) { return (VARIANT & 2) ? (j < NPV ? j : j - NPV) : j / 2; }
        };
        const unsigned vsel = (it % 3) * TILE * (zero + ((VARIANT & 1) ? 1 : 0)), ksel = ((it + 1) % 3) * TILE * (zero + ((VARIANT & 1) ? 1 : 0));
        unsigned kq[NKS], vl[NDV], vh[NDV];
        for (int ks = 0; ks < NKS; ++ks) kq[ks] = ka[ks] + ksel;
        for (int dvb = 0; dvb < NDV; ++dvb) { vl[dvb] = vlo[dvb] + vsel; vh[dvb] = vhi[dvb] + vsel; }
        constexpr int RD = 4;
        s16x8 ring[RD];
        auto fetch = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < NSTEP) {
                constexpr int x = Map::idx(j);
                if constexpr (Map::is_pv(j)) {
                    constexpr int kb = x / (2 * NDV), s2 = (x / NDV) & 1, dvb = x % NDV, off = (32 * kb + 16 * s2) * ROWB;
                    if constexpr (VARIANT & 4) ring[j % RD] = cat8(lds_tr16_at(vl[dvb] + off), lds_tr16_at(vh[dvb] + off));
                    else ring[j % RD] = cat8(lds_tr16_asm<off>(vl[dvb]), lds_tr16_asm<off>(vh[dvb]));
                } else {
                    constexpr int kb = x % KB, ks = x / KB;
                    if constexpr (VARIANT & 4) ring[j % RD] = lds_b128_at(kq[ks] + 32 * kb * ROWB);
                    else ring[j % RD] = lds_b128_asm<32 * kb * ROWB>(kq[ks]);
                }
            }
        };
        auto nops = [](int j) constexpr { return j >= NSTEP ? 0 : (Map::is_pv(j) ? 2 : 1); };
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value, x = Map::idx(j);
            fetch(std::integral_constant<int, j + RD - 1>{});
            __builtin_amdgcn_sched_barrier(0);
            constexpr int newer = [&]() constexpr { int c = 0; for (int q = 1; q < RD; ++q) c += nops(j + q); return c; }();
            if constexpr (Map::is_pv(j)) {
                constexpr int kb = x / (2 * NDV), s2 = (x / NDV) & 1, dvb = x % NDV;
                if constexpr (VARIANT & 4) oacc[dvb] = mfma32<bf16_tag>(ring[j % RD], *reinterpret_cast<s16x8*>(&pp[kb][s2]), oacc[dvb]);
                else MfmaWait<bf16_tag, newer>::acc(ring[j % RD], *reinterpret_cast<s16x8*>(&pp[kb][s2]), oacc[dvb]);
            } else {
                constexpr int kb = x % KB, ks = x / KB;
                if constexpr (VARIANT & 4) sacc[kb] = mfma32<bf16_tag>(ring[j % RD], qf[ks], sacc[kb]);
                else if constexpr (ks == 0) MfmaWait<bf16_tag, newer>::first(ring[j % RD], qf[ks], sacc[kb]);
                else MfmaWait<bf16_tag, newer>::acc(ring[j % RD], qf[ks], sacc[kb]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for_each_const(fetch, std::make_integer_sequence<int, RD - 1>{});
        for_each_const(step, std::make_integer_sequence<int, NSTEP>{});
        asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        cyc += t1 - t0;
        if (VARIANT & 8) __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < NDV; ++t)
        for (int i = 0; i < 16; ++i) s += oacc[t][i];
    for (int t = 0; t < KB; ++t)
        for (int i = 0; i < 16; ++i) s += sacc[t][i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[16] = (float)((double)cyc / iters);
}

template <int VARIANT>
static void run(const char* what, float* d) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kreal<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 64 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kreal<VARIANT>), dim3(256), dim3(512), 6 * 64 * 256, 0, d, 10, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kreal<VARIANT>), dim3(256), dim3(512), 6 * 64 * 256, 0, d, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0, cyc = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&cyc, d + 16, 4, hipMemcpyDeviceToHost);
    printf("variant %d  %-64s %7.3f ms  %.1f ns per MFMA  %.0f shader clocks per 32-MFMA phase (%.1f per MFMA)\n", VARIANT, what, ms,
           ms * 1e6 / iters / 32, cyc, cyc / 32);
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    run<0>("kernel's matrix phase, K at 0, V at 16 KiB", d);
    run<1>("tiles spread over 96 KiB (3 + 3 buffers), rotating", d);
    run<2>("P.V block then S block (no chain interleave)", d);
    run<4>("compiler-managed reads and waits", d);
    run<5>("compiler-managed, tiles spread over 96 KiB", d);
    run<8>("kernel's matrix phase, partner wave resident and waiting at the barrier", d);
    return 0;
}
