// Do MFMA and VALU instructions overlap on a gfx950 SIMD?  (round 5: the FIR-on-MFMA probe and the ampb knock-outs both came out
// ADDITIVE.)  One loop body = NM independent v_mfma_f32_32x32x16_f16 (4 accumulators round-robin) + NV independent v_fma_f32
// (8 chains), interleaved by the source order (mode 0), or as two blocks (mode 1: all MFMAs, then all FMAs); run with 1 and 2 waves
// per SIMD; compared with the loop that has only the MFMAs and the loop that has only the FMAs.
// hipcc --offload-arch=gfx950 -O3 tests/experiments/mfma_valu_overlap.hip -o tests/experiments/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f); }
    f32x16 acc[4] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1) + threadIdx.x;
    const float m = 1.0001f, c = seed;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            constexpr int G = NM > 0 ? NM : 1;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (NM > 0) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g & 3], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / G; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], m, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < NM; ++g) acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], m, c);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int MODE>
static float run(int wgs, float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        k<NM, NV, MODE><<<wgs, 256>>>(d, iters, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int wgs = 256 * wps;    // 256 CUs x (1 | 2) workgroups of 4 waves = 1 | 2 waves per SIMD
        const float tm = run<8, 0, 0>(wgs, d, iters), tv = run<0, 64, 0>(wgs, d, iters);
        const float ti = run<8, 64, 0>(wgs, d, iters), tb = run<8, 64, 1>(wgs, d, iters);
        printf("%d wave(s) / SIMD, per loop body of 8 MFMA (32x32x16 f16) + 64 v_fma_f32:  MFMA only %.3f ms   VALU only %.3f ms   interleaved %.3f ms   "
               "two blocks %.3f ms   (sum %.3f, max %.3f)\n", wps, tm, tv, ti, tb, tm + tv, tm > tv ? tm : tv);
        // cycles per body at the clock implied by the MFMA-only loop (8 x 32 cycles per wave and body)
        printf("   MFMA-only: %.1f ns per body per wave-slot -> %.2f GHz if 8 passes each\n", tm * 1e6 / iters, 8 * 32 * wps / (tm * 1e6 / iters));
    }
    // how much VALU work fits under the MFMAs: v : m from 1 to 8 (8 MFMAs per body; interleaved)
    for (int wps = 1; wps <= 2; ++wps) {
        const int wgs = 256 * wps;
        const float t0 = run<8, 0, 0>(wgs, d, iters);
        const float t1 = run<8, 8, 0>(wgs, d, iters), t2 = run<8, 16, 0>(wgs, d, iters), t4 = run<8, 32, 0>(wgs, d, iters), t8 = run<8, 64, 0>(wgs, d, iters);
        const float v1 = run<0, 8, 0>(wgs, d, iters), v2 = run<0, 16, 0>(wgs, d, iters), v4 = run<0, 32, 0>(wgs, d, iters), v8 = run<0, 64, 0>(wgs, d, iters);
        printf("%d wave(s) / SIMD  v:m      1      2      4      8\n   MFMA + VALU    %6.3f %6.3f %6.3f %6.3f ms  (MFMA alone %.3f)\n   VALU alone     %6.3f %6.3f %6.3f %6.3f ms\n"
               "   hidden share of the VALU time: %4.0f%% %4.0f%% %4.0f%% %4.0f%%\n", wps, t1, t2, t4, t8, t0, v1, v2, v4, v8,
               100 * (1 - (t1 - t0) / v1), 100 * (1 - (t2 - t0) / v2), 100 * (1 - (t4 - t0) / v4), 100 * (1 - (t8 - t0) / v8));
    }
    return 0;
}
