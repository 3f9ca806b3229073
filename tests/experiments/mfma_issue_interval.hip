// How often can a v_mfma_f32_32x32x16_f16 issue?  Zero operands (no power limit), one launch per variant: accumulators in VGPRs (what hipcc picks
// for a small kernel) or pinned to AGPRs, 4 or 8 independent accumulators, 1 / 2 / 4 / 8 waves per SIMD.  Prints cycles per MFMA from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 mfma_issue_interval.hip -o mfma_issue_interval      (round 5)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool AGPR>
__global__ __launch_bounds__(1024) void loop(const f16x8* __restrict__ ab, float* __restrict__ out, long long* cyc, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a0 = ab[tid & 1023], b0 = ab[(tid + 7) & 1023];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a0), "v"(b0));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a0), "v"(b0));
            }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = w1 - w0; }
}

template <int NACC, bool AGPR>
static void run(const char* name, int threads, f16x8* d, float* o, long long* c) {
    const int blocks = 256, iters = 16384;              // ONE workgroup per CU: threads / 256 waves per SIMD, all resident for the whole run
    hipLaunchKernelGGL((loop<NACC, AGPR>), dim3(blocks), dim3(threads), 0, 0, d, o, c, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop<NACC, AGPR>), dim3(blocks), dim3(threads), 0, 0, d, o, c, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), c, 2 * blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0, wsum = 0; for (int i = 0; i < blocks; ++i) { sum += (double)h[i]; wsum += (double)h[blocks + i]; }
    const double waves_per_simd = threads / 256.0;
    const double per_wave = sum / blocks / (iters * 16.0);                 // cycles between a wave's own MFMAs (wave 0 of each workgroup)
    const double tf = (double)blocks * (threads / 64) * iters * 16 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    const double mhz = sum / blocks / (ms * 1e3);                           // the loop's cycles over the launch's microseconds
    const double wall_us = wsum / blocks / 100.0;                           // s_memrealtime: 100 MHz
    printf("%-26s %4.2f waves / SIMD: %6.1f clock64 ticks per MFMA per wave;  loop %8.1f us by s_memrealtime, launch %8.1f us by events => clock64 runs at %6.0f MHz;  %7.1f TFLOP/s (%.1f %% of 2516.6)\n",
           name, waves_per_simd, per_wave, wall_us, ms * 1e3, sum / blocks / wall_us, tf, 100.0 * tf / 2516.6);
    (void)mhz;
}

int main() {
    f16x8* d; float* o; long long* c;
    hipMalloc(&d, 1024 * sizeof(f16x8)); hipMemset(d, 0, 1024 * sizeof(f16x8));
    hipMalloc(&o, (size_t)256 * 1024 * 4); hipMalloc(&c, 512 * sizeof(long long));
    for (int threads = 256; threads <= 1024; threads *= 2) {
        run<2, false>("2 accumulators in VGPRs", threads, d, o, c);
        run<2, true>("2 accumulators in AGPRs", threads, d, o, c);
        run<4, false>("4 accumulators in VGPRs", threads, d, o, c);
        run<4, true>("4 accumulators in AGPRs", threads, d, o, c);
    }
    return 0;
}
