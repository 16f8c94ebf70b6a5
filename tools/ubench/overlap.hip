// Micro-benchmark: how much of one wave's VALU work hides under the other wave's MFMAs on the same SIMD (gfx950)?
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench/overlap.hip -o tools/ubench/overlap
// Modes (512 threads = 2 waves per SIMD, 256 workgroups = one per CU):
//   0  waves 0-3: MFMA chain loop,    waves 4-7: idle
//   1  waves 0-3: idle,               waves 4-7: VALU loop (exp2 + fma)
//   2  waves 0-3: MFMA chain loop,    waves 4-7: VALU loop                 (overlap across waves?)
//   3  every wave: [NM MFMAs][NV VALU] alternating phases, no barrier       (what the attention kernels look like)
//   4  every wave: MFMAs and VALU finely interleaved in one stream (1 MFMA : NV/NM VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NM, int NV>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float seed) {
    const int w = threadIdx.x >> 6;
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + threadIdx.x + i); b[i] = (short)(0x3800 + i); }
    f32x16 c0 = {}, c1 = {};
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i + threadIdx.x * 1e-3f;
    const bool do_m = MODE == 0 ? w < 4 : MODE == 1 ? false : MODE == 2 ? w < 4 : true;
    const bool do_v = MODE == 0 ? false : MODE == 1 ? w >= 4 : MODE == 2 ? w >= 4 : true;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
#pragma unroll
            for (int m = 0; m < NM; m += 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV / NM; ++v) { const int j = (m * NV / NM + v) & 15; x[j] = __builtin_amdgcn_exp2f(fmaf(x[j], 0.5f, -1.0f)); }
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV / NM; ++v) { const int j = ((m + 1) * NV / NM + v) & 15; x[j] = __builtin_amdgcn_exp2f(fmaf(x[j], 0.5f, -1.0f)); }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int m = 0; m < NM; m += 2) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_v) {
#pragma unroll
                for (int v = 0; v < NV; ++v) x[v & 15] = __builtin_amdgcn_exp2f(fmaf(x[v & 15], 0.5f, -1.0f));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + c0[i] + c1[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// LDS-fed MFMA chain, the shape of the attention kernels' S = K Q^T phase: every MFMA takes its A operand from a
// ds_read_b128.  DEPTH = reads kept in flight ahead of the MFMA that consumes them (1 = read, wait, mfma).
template <int DEPTH, int NM, bool ONE = false>
__global__ __launch_bounds__(512, 2) void kl(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char tile[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += 512) reinterpret_cast<int*>(tile)[i] = 0x3c003c00 + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char* base = tile + lane * 16;   // lane-linear 16-byte reads: conflict-free
    s16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (short)(0x3800 + i);
    f32x16 c0 = {}, c1 = {};
    if (ONE && threadIdx.x >= 256) return;   // one wave per SIMD
    for (int it = 0; it < iters; ++it) {
        s16x8 ring[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH - 1; ++j) ring[j] = *reinterpret_cast<const s16x8*>(base + 1024 * (j & 31));
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int j = m + DEPTH - 1;
            if (j < NM) ring[j % DEPTH] = *reinterpret_cast<const s16x8*>(base + 1024 * (j & 31));
            __builtin_amdgcn_sched_barrier(0);
            if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[m % DEPTH], b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[m % DEPTH], b, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int DEPTH, int NM, bool ONE = false>
static void runl(const char* what, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kl<DEPTH, NM, ONE>), dim3(256), dim3(512), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kl<DEPTH, NM, ONE>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("lds-fed depth %d NM=%d  %-50s %8.3f ms  %7.1f ns/iter  (%.1f ns per MFMA per SIMD)\n", DEPTH, NM, what, ms,
           ms * 1e6 / iters, ms * 1e6 / iters / ((ONE ? 1 : 2) * NM));
}

// The attention forward's two operand-read patterns on a swizzled [128 keys][128] bf16 tile (fa_common.h TileSwz<128>):
// PAT 0: K rows, ds_read_b128 (A operand of S^T = K Q^T);  PAT 1: V^T, two ds_read_b64_tr_b16 per MFMA;  PAT 2: both phases.
__device__ __forceinline__ int swz128(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}
template <int PAT>
__global__ __launch_bounds__(512, 2) void kp(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char tile[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<int*>(tile)[i] = 0x3c003c00 + (i & 1023);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    const char* Kt = tile;
    const char* Vt = tile + 32768;
    s16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (short)(0x3800 + i);
    f32x16 c[4] = {};
    for (int it = 0; it < iters; ++it) {
        if (PAT == 0 || PAT == 2) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const s16x8 a = *reinterpret_cast<const s16x8*>(Kt + swz128(32 * kb + r, 2 * ks + h));
                    c[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[kb], 0, 0, 0);
                }
        }
        if (PAT == 1 || PAT == 2) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dvb = 0; dvb < 4; ++dvb) {
                        const int key_a = 32 * kb + 16 * s2 + 4 * h + tq, ch = 4 * dvb + 2 * g16 + (tp >> 1);
                        const s16x4 lo = tr16(Vt + swz128(key_a, ch) + 8 * (tp & 1));
                        const s16x4 hi = tr16(Vt + swz128(key_a + 8, ch) + 8 * (tp & 1));
                        const s16x8 a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        c[dvb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[dvb], 0, 0, 0);
                    }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i] + c[2][i] + c[3][i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
// Ping-pong: waves 0-3 and 4-7 alternate a matrix phase (64 LDS-fed MFMAs = one forward tile: S then P.V patterns) and
// a vector phase (NV exp2+fma pairs = the tile's softmax), one s_barrier per phase when SYNC, free running otherwise.
// LOCK: both halves run the same phase at the same time (what a lock-step kernel does).
template <int NV, bool SYNC, bool LOCK, bool PF = false, int NSTEP = 64, bool VARB = false, int PRIO = 0>
__global__ __launch_bounds__(512, 2) void kpp(float* out, int iters, int roff = 0) {
    __shared__ __attribute__((aligned(16))) char tile[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<int*>(tile)[i] = 0x3c003c00 + (i & 1023);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const int li = lane & 15, g16 = (lane >> 4) & 1, tq = li >> 2, tp = li & 3;
    const char* Kt = tile;
    const char* Vt = tile + 32768;
    s16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (short)(0x3800 + i);
    s16x8 bq[8];   // VARB: a different B operand per step, like the Q fragments / packed P of the real kernel
    for (int q = 0; q < 8; ++q)
        for (int i = 0; i < 8; ++i) bq[q][i] = (short)(0x3800 + i + q + (threadIdx.x & 3));
    f32x16 c[4] = {};
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 1.0f + i + threadIdx.x * 1e-3f;
    const int half = LOCK ? 0 : (w >= 4);
    if (VARB) { Kt += roff; Vt += roff; }   // runtime tile offset: addresses need a v_add per read
    unsigned long long mcyc = 0, vcyc = 0, t0 = 0, t1 = 0;
    for (int ph = 0; ph < 2 * iters; ++ph) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if (((ph + half) & 1) == 0) {
          if (PF) {   // operand fragments two MFMAs ahead, order pinned
            auto kfrag = [&](int j) { return *reinterpret_cast<const s16x8*>(Kt + swz128(32 * (j >> 3) + r, 2 * (j & 7) + h)); };
            auto vfrag = [&](int j) {
                const int kb = j >> 3, s2 = (j >> 2) & 1, dvb = j & 3;
                const int key_a = 32 * kb + 16 * s2 + 4 * h + tq, ch = 4 * dvb + 2 * g16 + (tp >> 1);
                const s16x4 lo = tr16(Vt + swz128(key_a, ch) + 8 * (tp & 1));
                const s16x4 hi = tr16(Vt + swz128(key_a + 8, ch) + 8 * (tp & 1));
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            s16x8 ring[3];
            ring[0] = kfrag(0); ring[1] = kfrag(1);
#pragma unroll
            for (int j = 0; j < NSTEP; ++j) {
                if (j + 2 < NSTEP / 2) ring[(j + 2) % 3] = kfrag(j + 2);
                else if (j + 2 < NSTEP) ring[(j + 2) % 3] = vfrag(j + 2 - NSTEP / 2);
                __builtin_amdgcn_sched_barrier(0);
                const int ci = j < NSTEP / 2 ? (j >> 3) : (j & 3);
                c[ci] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[j % 3], VARB ? bq[j & 7] : b, c[ci], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
          } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const s16x8 a = *reinterpret_cast<const s16x8*>(Kt + swz128(32 * kb + r, 2 * ks + h));
                    c[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[kb], 0, 0, 0);
                }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dvb = 0; dvb < 4; ++dvb) {
                        const int key_a = 32 * kb + 16 * s2 + 4 * h + tq, ch = 4 * dvb + 2 * g16 + (tp >> 1);
                        const s16x4 lo = tr16(Vt + swz128(key_a, ch) + 8 * (tp & 1));
                        const s16x4 hi = tr16(Vt + swz128(key_a + 8, ch) + 8 * (tp & 1));
                        const s16x8 a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        c[dvb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[dvb], 0, 0, 0);
                    }
          }
        } else {
            if (PRIO == 1) __builtin_amdgcn_s_setprio(3);   // vector phase outranks the partner's matrix phase
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) x[v & 15] = __builtin_amdgcn_exp2f(fmaf(x[v & 15], 0.5f, -1.0f));
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(3);   // reverse: matrix phase outranks
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (((ph + half) & 1) == 0) mcyc += t1 - t0; else vcyc += t1 - t0;
        if (SYNC) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + c[0][i] + c[1][i] + c[2][i] + c[3][i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) {
        out[16 + 4 * (threadIdx.x >> 8)] = (float)((double)mcyc / iters);
        out[17 + 4 * (threadIdx.x >> 8)] = (float)((double)vcyc / iters);
    }
}
template <int NV, bool SYNC, bool LOCK, bool PF = false, int NSTEP = 64, bool VARB = false, int PRIO = 0>
static void runpp(const char* what, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kpp<NV, SYNC, LOCK, PF, NSTEP, VARB, PRIO>), dim3(256), dim3(512), 0, 0, d, 10, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kpp<NV, SYNC, LOCK, PF, NSTEP, VARB, PRIO>), dim3(256), dim3(512), 0, 0, d, iters, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    float hc[8];
    hipMemcpy(hc, d + 16, sizeof(hc), hipMemcpyDeviceToHost);
    printf("ping-pong NV=%d sync=%d lock=%d  %-44s %8.3f ms  (%.1f ns per MFMA per SIMD; floor 14.8)  shader clocks per phase: wave0 M %.0f V %.0f | wave4 M %.0f V %.0f\n",
           NV, SYNC, LOCK, what, ms, ms * 1e6 / iters / (2 * NSTEP), hc[0], hc[1], hc[4], hc[5]);
}

template <int PAT>
static void runp(const char* what, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kp<PAT>), dim3(256), dim3(512), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kp<PAT>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int nm = PAT == 2 ? 64 : 32;
    printf("pattern %d  %-60s %8.3f ms  (%.1f ns per MFMA per SIMD)\n", PAT, what, ms, ms * 1e6 / iters / (2 * nm));
}

template <int MODE, int NM, int NV>
static void run(const char* what, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(256), dim3(512), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(256), dim3(512), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per iteration per SIMD at 2.4 GHz nominal
    printf("mode %d NM=%d NV=%d  %-58s %8.3f ms  %7.1f ns/iter\n", MODE, NM, NV, what, ms, ms * 1e6 / iters);
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int iters = 20000;
    run<0, 16, 32>("MFMA only (waves 0-3): 16 MFMA / iter", d, iters);
    run<1, 16, 32>("VALU only (waves 4-7): 32 exp2+fma / iter", d, iters);
    run<2, 16, 32>("MFMA waves + VALU waves together", d, iters);
    run<1, 16, 64>("VALU only: 64 exp2+fma / iter", d, iters);
    run<2, 16, 64>("MFMA waves + VALU(64) waves together", d, iters);
    run<3, 16, 32>("all 8 waves: [16 MFMA][32 VALU] phases", d, iters);
    run<4, 16, 32>("all 8 waves: 1 MFMA : 2 VALU interleaved", d, iters);
    run<3, 16, 64>("all 8 waves: [16 MFMA][64 VALU] phases", d, iters);
    run<4, 16, 64>("all 8 waves: 1 MFMA : 4 VALU interleaved", d, iters);
    run<3, 32, 64>("all 8 waves: [32 MFMA][64 VALU] phases", d, iters);
    run<3, 64, 128>("all 8 waves: [64 MFMA][128 VALU] phases", d, iters);
    runl<1, 16>("8 waves: read, wait, mfma", d, iters);
    runl<2, 16>("8 waves: 1 read ahead", d, iters);
    runl<3, 16>("8 waves: 2 reads ahead", d, iters);
    runl<5, 16>("8 waves: 4 reads ahead", d, iters);
    runl<1, 16, true>("ONE wave per SIMD: read, wait, mfma", d, iters);
    runl<2, 16, true>("ONE wave per SIMD: 1 read ahead", d, iters);
    runl<3, 16, true>("ONE wave per SIMD: 2 reads ahead", d, iters);
    runl<5, 16, true>("ONE wave per SIMD: 4 reads ahead", d, iters);
    runp<0>("K rows: ds_read_b128 per MFMA (swizzled tile)", d, iters / 2);
    runp<1>("V^T: 2 x ds_read_b64_tr_b16 per MFMA", d, iters / 2);
    runp<2>("both phases (one forward tile without softmax)", d, iters / 2);
    runpp<64, true, false>("tile = 64 MFMA | 64 exp2+fma, alternating halves, barrier", d, iters / 4);
    runpp<64, false, false>("same, free running", d, iters / 4);
    runpp<64, true, true>("same, both halves in the same phase (lock-step)", d, iters / 4);
    runpp<64, true, false, true>("alternating + barrier + operands 2 ahead", d, iters / 4);
    runpp<64, true, true, true>("lock-step + operands 2 ahead", d, iters / 4);
    runpp<96, true, false, true>("96/tile alternating + barrier + operands 2 ahead", d, iters / 4);
    runpp<128, true, false, true>("128/tile alternating + barrier + operands 2 ahead", d, iters / 4);
    runpp<128, true, true, false>("128/tile lock-step", d, iters / 4);
    runpp<32, true, false, true, 32>("32 MFMA | 32 pairs per phase, alternating + prefetch", d, iters / 4);
    runpp<32, true, false, true, 32, true>("same + rotating B operands + runtime tile offset", d, iters / 4);
    runpp<0, true, false, true, 32, true>("same, empty V phase", d, iters / 4);
    runpp<64, true, false, true, 32>("32 MFMA | 64 pairs per phase, alternating + prefetch", d, iters / 4);
    runpp<64, true, false, true, 32, true, 0>("32 MFMA | 64 pairs, alternating + prefetch, equal priority", d, iters / 4);
    runpp<64, true, false, true, 32, true, 1>("same, vector phase at s_setprio 3", d, iters / 4);
    runpp<64, true, false, true, 32, true, 2>("same, matrix phase at s_setprio 3", d, iters / 4);
    runpp<96, true, false, true, 32, true, 0>("32 MFMA | 96 pairs, equal priority", d, iters / 4);
    runpp<96, true, false, true, 32, true, 1>("32 MFMA | 96 pairs, vector phase at s_setprio 3", d, iters / 4);
    runpp<96, true, false>("96 exp2+fma per tile, alternating, barrier", d, iters / 4);
    runpp<96, true, true>("96 exp2+fma per tile, lock-step", d, iters / 4);
    return 0;
}
