// Sustained v_mfma_f32_32x32x16_f16 rate on MI355X with random vs zero operands (power/clock ceiling
// that bounds the f16x3 conv kernels).   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_loop(const f16x8* __restrict__ ab, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f16x8 a0 = ab[tid], a1 = ab[tid + gridDim.x * 256], b0 = ab[tid + 2 * gridDim.x * 256], b1 = ab[tid + 3 * gridDim.x * 256];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
}

// `mfma_peak <seconds>`: the random-operand loop (mode 2) launched back to back for that long -- a steady load for tools/power_per_kernel.py to sample
// power and clock under; prints the sustained rate of the second half.
static int sustained(double seconds);

int main(int argc, char** argv) {
    if (argc > 1) return sustained(atof(argv[1]));
    const int blocks = 256 * 8, iters = 4096;
    const size_t n = (size_t)blocks * 256 * 4;
    std::vector<_Float16> h(n * 8);
    f16x8* d; float* o;
    hipMalloc(&d, n * sizeof(f16x8)); hipMalloc(&o, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {   // 0: zeros, 1: random small (products ~ N(0, 1e-2)), 2: random O(1..1000) like the conv operands
        srand(1);
        for (auto& v : h) { float r = (float)rand() / RAND_MAX * 2.f - 1.f; v = (_Float16)(mode == 0 ? 0.f : mode == 1 ? r * 0.1f : r * 1000.f); }
        hipMemcpy(d, h.data(), n * sizeof(f16x8), hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
            if (rep == 2) printf("mode %d (%s): %.3f ms  %.1f TFLOP/s  (%.1f %% of 2516.6)\n", mode, mode == 0 ? "zeros" : mode == 1 ? "random small" : "random large", ms, flop / ms / 1e9, 100.0 * flop / ms / 1e9 / 2516.6);
        }
    }
    return 0;
}

#include <chrono>
static int sustained(double seconds) {
    const int blocks = 256 * 8, iters = 4096;
    const size_t n = (size_t)blocks * 256 * 4;
    std::vector<_Float16> h(n * 8);
    f16x8* d; float* o;
    hipMalloc(&d, n * sizeof(f16x8)); hipMalloc(&o, (size_t)blocks * 256 * 4);
    srand(1);
    for (auto& v : h) { float r = (float)rand() / RAND_MAX * 2.f - 1.f; v = (_Float16)(r * 1000.f); }
    hipMemcpy(d, h.data(), n * sizeof(f16x8), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    double tf_last = 0;
    for (;;) {
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        tf_last = 20.0 * (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el >= seconds) break;
    }
    printf("sustained %.1f s: %.1f TFLOP/s (random operands, last 20 launches)\n", seconds, tf_last);
    return 0;
}
