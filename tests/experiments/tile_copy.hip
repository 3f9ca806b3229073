// Design-time experiment (round 3): how fast can one workgroup-tile of a [B, C = 32, T] fp32 tensor be read and written back in
// (A) the MFMA C layout the whole-tile kernels use (lane = column, 16 rows per lane: dword accesses, 128 B per row per half-wave),
// (B) float4 rows (lane = 4 consecutive columns: 1 KB per wave instruction), with the same occupancy (72 KB of LDS per workgroup:
// two workgroups per CU)?   hipcc --offload-arch=gfx950 -O3 tests/experiments/tile_copy.hip -o tests/experiments/tile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int C = 32, W = 512;

template <int MODE>
__global__ __launch_bounds__(256, 2) void copy_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int tiles_per_item) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int item = blockIdx.x / tiles_per_item, tile = blockIdx.x % tiles_per_item;
    const float* xb = x + (size_t)item * C * T + (size_t)tile * W;
    float* yb = y + (size_t)item * C * T + (size_t)tile * W;
    if (MODE == 0) {          // C layout: wave = 128 columns, lane l31 = column, hi selects rows +4; 16 rows (r&3)+8(r>>2)+4hi
        const int hi = lane >> 5, l31 = lane & 31;
        float v[4][16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[t][r] = xb[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * T + wave * 128 + 32 * t + l31];
        if (lds[tid] == 12345.f) v[0][0] += 1.f;       // keep the LDS allocation alive
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) yb[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hi) * T + wave * 128 + 32 * t + l31] = v[t][r] * 1.0001f;
    } else {                  // float4 rows: wave w takes rows 8w .. 8w+7; a row of 512 columns = 2 wave instructions
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float4*>(xb + (size_t)(8 * wave + (i >> 1)) * T + 256 * (i & 1) + 4 * lane);
        if (lds[tid] == 12345.f) v[0].x += 1.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float4 o = v[i]; o.x *= 1.0001f; o.y *= 1.0001f; o.z *= 1.0001f; o.w *= 1.0001f;
            *reinterpret_cast<float4*>(yb + (size_t)(8 * wave + (i >> 1)) * T + 256 * (i & 1) + 4 * lane) = o;
        }
    }
}

int main() {
    const int B = 64, T = 65536, tiles = T / W;
    const size_t n = (size_t)B * C * T;
    float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
    hipMemset(x, 0, n * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&copy_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&copy_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int lds_kb : {72, 16}) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(copy_kernel<0>, dim3(B * tiles), dim3(256), lds_kb * 1024, 0, x, y, T, tiles);
                else hipLaunchKernelGGL(copy_kernel<1>, dim3(B * tiles), dim3(256), lds_kb * 1024, 0, x, y, T, tiles);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("mode %s, %2d KB LDS per workgroup: %.1f us = %.2f TB/s (read + write of %.0f MB each)\n", mode == 0 ? "C-layout dword" : "float4 rows   ",
                   lds_kb, best * 1e3, 2.0 * n * 4 / best / 1e9, n * 4 / 1e6);
        }
    }
    return 0;
}
