// Micro-benchmark for VERDICT r2 item 4: would v_mfma_f32_16x16x32_bf16 pay in the stream kernels?
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_shape.hip -o tools/ubench/mfma_shape
// One wave per SIMD (256-thread workgroups, one per CU, the mapping of fa_bwd_dkdv_w4.hip), random bf16 operands in registers,
// one 32 x 32 output tile per wave.  A "unit" is one K = 32 step over that tile = 65 536 FLOP = 64 cycles of the matrix pipe in
// either shape:   SHAPE 0: 2 x v_mfma_f32_32x32x16_bf16 (2 A and 2 B fragments)
//                 SHAPE 1: 4 x v_mfma_f32_16x16x32_bf16 (2 A and 2 B fragments, four 16 x 16 accumulators)
// with F independent single-issue vector instructions (v_fma_f32; every fourth a v_exp_f32) spread evenly over the unit's MFMA
// gaps — the dK/dV stream carries about 5 per 32x32x16 gap = F 10 per unit — and optionally L ds_read_b128 per
// unit (the stream reads 0.75 KB of LDS per 32x32x16 MFMA = 1.5 per unit).  Prints ns per unit per SIMD and the TFLOP/s the
// chip would deliver at that rate, after 2 s of warm-up launches (the clock under load is what is being compared: the guide
// measures 1.12 - 1.15 x for the 16x16x32 shape on bare loops, MI355X_MICROARCH.md 'DVFS give-back' item 7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// a bf16 value in (-2, 2) with random mantissa and sign
__device__ __forceinline__ short rnd_bf16(unsigned seed) {
    const unsigned h = hash(seed);
    return (short)(((h & 0x807f) | ((126u + ((h >> 8) & 1u)) << 7)) & 0xffff);
}

template <int F, int SLOT, int NSLOT, int U>
__device__ __forceinline__ void fillers(float (&x)[16]) {   // this slot's share of the unit's F vector instructions
    constexpr int lo = F * SLOT / NSLOT, hi = F * (SLOT + 1) / NSLOT;
#pragma unroll
    for (int j = lo; j < hi; ++j) {   // independent instructions on rotating registers; the values stay in (-0.7, 2)
        if ((j & 3) == 3) x[(j + 3 * U) & 15] = __builtin_amdgcn_exp2f(x[(j + 3 * U) & 15]);
        else x[(j + 3 * U) & 15] = fmaf(x[(j + 3 * U) & 15], 0.25f, -0.5f);
    }
}

template <int SHAPE, int F, int L>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char tile[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<unsigned*>(tile)[i] = hash(i) & 0xbf7fbf7fu;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char* base = tile + lane * 16;
    s16x8 a[2], b[2];
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 8; ++i) {
            a[t][i] = rnd_bf16(threadIdx.x * 64 + blockIdx.x * 131 + t * 8 + i);
            b[t][i] = rnd_bf16(threadIdx.x * 64 + blockIdx.x * 131 + 16 + t * 8 + i);
        }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = -0.5f - 0.01f * i - 1e-4f * threadIdx.x;
    f32x16 c32 = {};
    f32x4 c16[4] = {};
    s16x8 ld[2][2] = {{a[0], a[1]}, {a[0], a[1]}};   // operands of the unit in flight / of the next one (read one unit ahead)
    for (int it = 0; it < iters; ++it) {
        auto unit = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (L >= 1) ld[(u + 1) & 1][0] = *reinterpret_cast<const s16x8*>(base + 1024 * ((2 * u) & 31));
            if (L >= 2) ld[(u + 1) & 1][1] = *reinterpret_cast<const s16x8*>(base + 1024 * ((2 * u + 1) & 31));
            const s16x8 a0 = L >= 1 ? ld[u & 1][0] : a[0], a1 = L >= 2 ? ld[u & 1][1] : a[1];
            __builtin_amdgcn_sched_barrier(0);
            if (SHAPE == 0) {
                c32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b[0]), c32, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                fillers<F, 0, 2, u>(x);
                __builtin_amdgcn_sched_barrier(0);
                c32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b[1]), c32, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                fillers<F, 1, 2, u>(x);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    c16[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (q & 1) ? a1 : a0),
                                                                     __builtin_bit_cast(bf16x8, b[q >> 1]), c16[q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q == 0) fillers<F, 0, 4, u>(x);
                    if (q == 1) fillers<F, 1, 4, u>(x);
                    if (q == 2) fillers<F, 2, 4, u>(x);
                    if (q == 3) fillers<F, 3, 4, u>(x);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{}); unit(std::integral_constant<int, 2>{}); unit(std::integral_constant<int, 3>{});
        unit(std::integral_constant<int, 4>{}); unit(std::integral_constant<int, 5>{}); unit(std::integral_constant<int, 6>{}); unit(std::integral_constant<int, 7>{});
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + c32[i];
    for (int q = 0; q < 4; ++q) s += c16[q][0] + c16[q][1] + c16[q][2] + c16[q][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int SHAPE, int F, int L>
static double run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, F, L>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

template <int F, int L>
static void row(float* d) {
    const int iters = 60000;                    // 480 000 units per wave: 15 - 25 ms per launch
    double best[2] = {1e30, 1e30}, med[2];
    for (int rep = 0; rep < 7; ++rep) {         // interleaved: both shapes see the same chip state
        const double t0 = run<0, F, L>(d, iters), t1 = run<1, F, L>(d, iters);
        if (t0 < best[0]) best[0] = t0;
        if (t1 < best[1]) best[1] = t1;
        med[0] = t0; med[1] = t1;
    }
    const double units = 8.0 * iters;
    for (int s = 0; s < 2; ++s) {
        const double ns = best[s] * 1e6 / units;
        printf("| %-9s | %2d | %d | %7.2f | %6.0f |\n", s ? "16x16x32" : "32x32x16", F, L, ns, 65536.0 / ns * 1024.0 / 1e3);
    }
    printf("|   time 16x16x32 / 32x32x16: %.3f (last pair %.3f) | | | | |\n", best[1] / best[0], med[1] / med[0]);
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    for (int i = 0; i < 60; ++i) run<0, 10, 2>(d, 60000);   // ~2 s of load before anything is timed
    printf("| shape | F vector instructions per unit | L ds_read_b128 per unit | ns per unit per SIMD (the matrix pipe's floor: 64 cycles) | chip TFLOP/s at that rate |\n|---|---|---|---|---|\n");
    row<0, 0>(d);
    row<4, 0>(d);
    row<8, 0>(d);
    row<10, 0>(d);
    row<12, 0>(d);
    row<0, 2>(d);
    row<8, 2>(d);
    row<10, 2>(d);
    row<12, 2>(d);
    hipFree(d);
    return 0;
}
