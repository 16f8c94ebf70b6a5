#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A: one-hot probes to find the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 (fmt 0) operands.
// a_bytes[lane][32], b_bytes[lane][32]: raw e4m3 bytes per lane; out[lane][16]
__global__ void k(const uint8_t* a_bytes, const uint8_t* b_bytes, float* out, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    memcpy(&a, a_bytes + lane * 32, 32);
    memcpy(&b, b_bytes + lane * 32, 32);
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
int main() {
    uint8_t *da, *db; float* dout;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dout, 64 * 16 * 4);
    static uint8_t ha[64 * 32], hb[64 * 32]; static float ho[64 * 16];
    const uint8_t ONE = 0x38, TWO = 0x40;   // e4m3: 1.0 = 0x38, 2.0 = 0x40
    // probe 1: A all ones, B all ones, scales 127 -> every C = 64 if scale 127 means 1.0
    for (int sa : {127, 128}) {
        memset(ha, ONE, sizeof(ha)); memset(hb, ONE, sizeof(hb));
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, sa, 127);
        hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        printf("all ones, scale_a=%d scale_b=127: c[0][0]=%g c[63][15]=%g\n", sa, ho[0], ho[63 * 16 + 15]);
    }
    // probe 2: A one-hot at (lane la, byte ba) = 2.0, B all ones: which C rows light up (value 2)?  C[row][col]: col = lane&31,
    // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  A[i][k]*B[k][j] summed over k: a single A element (i,k) gives C[i][j] += 2 for all j.
    for (int la : {0, 1, 31, 32, 33, 63}) for (int ba : {0, 1, 8, 16, 31}) {
        memset(ha, 0, sizeof(ha)); memset(hb, ONE, sizeof(hb));
        ha[la * 32 + ba] = TWO;
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, 127, 127);
        hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        // find which (lane,reg) are nonzero: report set of rows and whether all cols
        int rows[32] = {0}; int cnt = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (ho[l * 16 + r] != 0.f) { rows[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]++; cnt++; }
        printf("A one-hot lane %2d byte %2d: nonzeros %d rows:", la, ba, cnt);
        for (int i = 0; i < 32; ++i) if (rows[i]) printf(" %d(x%d)", i, rows[i]);
        printf("\n");
    }
    // probe 3: which k index does (lane, byte) of A pair with in B?  A one-hot (lane la, byte ba) = 2, B one-hot (lane lb, byte bb) = 2 -> C nonzero iff same k.
    for (int la : {0, 32}) for (int ba : {0, 5, 16, 31}) {
        printf("A(lane %d, byte %d) pairs with B (lane>>5, byte):", la, ba);
        for (int lbh = 0; lbh < 2; ++lbh) for (int bb = 0; bb < 32; ++bb) {
            memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
            ha[la * 32 + ba] = TWO; hb[(lbh * 32 + 3) * 32 + bb] = TWO;
            hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, 127, 127);
            hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
            bool nz = false; for (int i = 0; i < 64 * 16; ++i) nz |= ho[i] != 0.f;
            if (nz) printf(" (%d,%d)", lbh, bb);
        }
        printf("\n");
    }
    return 0;
}
