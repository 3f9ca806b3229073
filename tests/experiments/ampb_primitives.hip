// Hardware semantics the whole-AMPBlock kernel (amphion_amd/csrc/ampb_f16x3.hip) is built on, checked on the GPU itself:
//   1. v_permlane32_swap: lanes 32-63 of the first operand swap with lanes 0-31 of the second
//   2. DPP row_shl:4 / row_shr:4 / quad_perm [1,0,3,2] / [2,3,0,1] source lanes
//   3. v_mfma_f32_32x32x16_f16 with its operands exchanged gives the TRANSPOSED tile with the same bits
//   4. the 8 x 8 lane transpose (three DPP butterfly stages) used to turn "lane = channel" into "lane = column"
// hipcc --offload-arch=gfx950 -O2 ampb_primitives.hip -o ampb_primitives && ./ampb_primitives
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ void transpose8(float (&R)[8], bool b4, bool b2, bool b1) {
    // (the DPP moves are evaluated by EVERY lane, then selected: inside a conditional they would run under a partial EXEC mask and
    //  read disabled lanes)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float lo = R[j], hi = R[j + 4];
        const float from_below = dpp_mov<0x114>(hi), from_above = dpp_mov<0x104>(lo);
        R[j] = b4 ? from_below : lo;
        R[j + 4] = b4 ? hi : from_above;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = (jj & 1) + 4 * (jj >> 1);
        const float lo = R[j], hi = R[j + 2];
        const float phi = dpp_mov<0x4E>(hi), plo = dpp_mov<0x4E>(lo);
        R[j] = b2 ? phi : lo;
        R[j + 2] = b2 ? hi : plo;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 2 * jj;
        const float lo = R[j], hi = R[j + 1];
        const float phi = dpp_mov<0xB1>(hi), plo = dpp_mov<0xB1>(lo);
        R[j] = b1 ? phi : lo;
        R[j + 1] = b1 ? hi : plo;
    }
}

__global__ void prim_kernel(const _Float16* A, const _Float16* B, const float* Cin, float* out) {
    const int lane = threadIdx.x;
    // 1. permlane32_swap
    {
        const unsigned a = 1000u + lane, b = 2000u + lane;
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        out[lane] = (float)r[0];
        out[64 + lane] = (float)r[1];
    }
    // 2. DPP
    {
        const float v = (float)lane;
        out[128 + lane] = dpp_mov<0x104>(v);   // row_shl:4
        out[192 + lane] = dpp_mov<0x114>(v);   // row_shr:4
        out[256 + lane] = dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
        out[320 + lane] = dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    }
    // 3. MFMA: D1 = A[32x16] * B[16x32] + C ; D2 = B^T * A^T + C^T
    {
        f16x8 af, bf;
        for (int e = 0; e < 8; ++e) {
            af[e] = A[(lane & 31) * 16 + 8 * (lane >> 5) + e];      // A[m][k]
            bf[e] = B[(8 * (lane >> 5) + e) * 32 + (lane & 31)];    // B[k][n]
        }
        f32x16 c1, c2;
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), j = lane & 31;
            c1[r] = Cin[i * 32 + j];     // D1[i][j]
            c2[r] = Cin[j * 32 + i];     // D2[i][j] = D1[j][i]
        }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf, af, c2, 0, 0, 0);
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), j = lane & 31;
            out[384 + i * 32 + j] = c1[r];            // D1[i][j]
            out[384 + 1024 + j * 32 + i] = c2[r];     // D2[i][j] stored at [j][i]: must equal D1
        }
    }
    // 4. transpose8: lane e (of each group of 8) holds M[e][j] = 100 * lane + j  ->  afterwards M[j][e]
    {
        float R[8];
        for (int j = 0; j < 8; ++j) R[j] = 100.f * lane + j;
        transpose8(R, (lane & 4) != 0, (lane & 2) != 0, (lane & 1) != 0);
        for (int j = 0; j < 8; ++j) out[384 + 2048 + lane * 8 + j] = R[j];
    }
}

int main() {
    const int NOUT = 384 + 2048 + 512;
    _Float16 hA[32 * 16], hB[16 * 32];
    float hC[1024], ho[NOUT];
    srand(7);
    for (auto& v : hA) v = (_Float16)((rand() % 4001 - 2000) / 16.0f);
    for (auto& v : hB) v = (_Float16)((rand() % 4001 - 2000) / 64.0f);
    for (auto& v : hC) v = (rand() % 20001 - 10000) / 3.0f;
    _Float16 *dA, *dB;
    float *dC, *dO;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dO, sizeof(ho));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(prim_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
    if (hipMemcpy(ho, dO, sizeof(ho), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: hip error\n"); return 2; }
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const float e0 = l < 32 ? 1000.f + l : 2000.f + (l - 32);   // first operand: lanes 32-63 <- second's lanes 0-31
        const float e1 = l < 32 ? 1000.f + (l + 32) : 2000.f + l;   // second operand: lanes 0-31 <- first's lanes 32-63
        if (ho[l] != e0 || ho[64 + l] != e1) { if (!bad) printf("permlane32_swap lane %d: %g %g (want %g %g)\n", l, ho[l], ho[64 + l], e0, e1); ++bad; }
    }
    printf("1 permlane32_swap: %s\n", bad ? "FAIL" : "ok");
    int bad2 = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15, i = l & 15;
        const float shl = (i + 4 < 16) ? (float)(row + i + 4) : 0.f, shr = (i >= 4) ? (float)(row + i - 4) : 0.f;
        if (ho[128 + l] != shl || ho[192 + l] != shr || ho[256 + l] != (float)(l ^ 1) || ho[320 + l] != (float)(l ^ 2)) {
            if (!bad2) printf("dpp lane %d: shl4 %g shr4 %g qp1 %g qp2 %g\n", l, ho[128 + l], ho[192 + l], ho[256 + l], ho[320 + l]);
            ++bad2;
        }
    }
    printf("2 dpp: %s\n", bad2 ? "FAIL" : "ok");
    int bad3 = 0;
    for (int i = 0; i < 1024; ++i)
        if (memcmp(&ho[384 + i], &ho[384 + 1024 + i], 4) != 0) { if (!bad3) printf("mfma swap [%d]: %.9g vs %.9g\n", i, ho[384 + i], ho[384 + 1024 + i]); ++bad3; }
    printf("3 mfma operand swap bitwise: %s (%d mismatches)\n", bad3 ? "FAIL" : "ok", bad3);
    int bad4 = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int g = l & ~7, e = l & 7;
            const float want = 100.f * (g + j) + e;
            if (ho[384 + 2048 + l * 8 + j] != want) { if (!bad4) printf("transpose8 lane %d reg %d: %g want %g\n", l, j, ho[384 + 2048 + l * 8 + j], want); ++bad4; }
        }
    printf("4 transpose8: %s\n", bad4 ? "FAIL" : "ok");
    printf("%s\n", (bad || bad2 || bad3 || bad4) ? "PRIMITIVES FAIL" : "PRIMITIVES PASS");
    return (bad || bad2 || bad3 || bad4) ? 1 : 0;
}
