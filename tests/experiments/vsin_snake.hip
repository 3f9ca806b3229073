// Round 4: Snake's sin(alpha * u)^2 through the hardware sine with an ACCURATE reduction -- the error of round 2's probe
// (vsin_accuracy.hip) was the fp32 rounding of x / 2pi, not v_sin_f32 itself.  With c = alpha / (2 pi) split on the host into
// c_hi + c_lo:   w = u * c_hi;  k = rint(w);  f = fma(u, c_hi, -k);  f = fma(u, c_lo, f)   (|f| <= 0.5 revolutions, exact to ~1e-8)
//                sin(alpha u) = v_sin_f32(f)
// 4 ordinary instructions + 1 transcendental against 13 FMAs + rint + 2 multiplies, no large-argument path.  Reported: max abs error
// of sin^2 against double precision per argument range and alpha, for this form, for the polynomial form in use (act1d_math.h) and the
// throughput of both (a register-resident loop).
//   hipcc --offload-arch=gfx950 -O3 -I../../amphion_amd/csrc -I../../include vsin_snake.hip -o vsin_snake && ./vsin_snake
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

#include "act1d_math.h"

__device__ __forceinline__ float sin2_hw(float u, float c_hi, float c_lo) {
    const float w = u * c_hi;
    const float k = rintf(w);
    float f = fmaf(u, c_hi, -k);
    f = fmaf(u, c_lo, f);
    const float s = __builtin_amdgcn_sinf(f);
    return s * s;
}

__global__ void acc_kernel(const float* u, float alpha, float c_hi, float c_lo, float* hw, float* poly, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    hw[i] = sin2_hw(u[i], c_hi, c_lo);
    poly[i] = amp::snake_sin2(u[i] * alpha);
}

template <int MODE>
__global__ void rate_kernel(float* out, float alpha, float c_hi, float c_lo, int iters) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * 0.5f + (MODE == 0 ? amp::snake_sin2(v[j] * alpha) : sin2_hw(v[j], c_hi, c_lo));
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> h(n), a(n), b(n);
    float *du, *d1, *d2;
    hipMalloc(&du, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    const float alphas[4] = {0.37f, 1.0f, 2.718f, 15.9f};
    const float ranges[5] = {3.0f, 30.f, 300.f, 4094.f, 1e6f};
    for (int ai = 0; ai < 4; ++ai) {
        const double c = (double)alphas[ai] / (2.0 * M_PI);
        const float c_hi = (float)c, c_lo = (float)(c - (double)c_hi);
        for (int ri = 0; ri < 5; ++ri) {
            srand(1 + ri);
            for (int i = 0; i < n; ++i) h[i] = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * ranges[ri];
            hipMemcpy(du, h.data(), n * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(acc_kernel, dim3(n / 256), dim3(256), 0, 0, du, alphas[ai], c_hi, c_lo, d1, d2, n);
            hipMemcpy(a.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, n * 4, hipMemcpyDeviceToHost);
            double e1 = 0, e2 = 0;
            for (int i = 0; i < n; ++i) {
                const double s = sin((double)alphas[ai] * (double)h[i]), ref = s * s;
                e1 = fmax(e1, fabs(a[i] - ref));
                e2 = fmax(e2, fabs(b[i] - ref));
            }
            printf("alpha %-6g |u| <= %-7g (|alpha u| <= %-9g): hw-sine form %.3e   polynomial form %.3e\n", alphas[ai], ranges[ri], alphas[ai] * ranges[ri], e1, e2);
        }
    }
    // throughput: 8 independent chains per lane, 256 CUs x 8 workgroups x 256 threads
    const double c = 1.0 / (2.0 * M_PI);
    const float c_hi = (float)c, c_lo = (float)(c - (double)c_hi);
    float* dout; hipMalloc(&dout, 2048 * 256 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(2048), dim3(256), 0, 0, dout, 1.0f, c_hi, c_lo, 2000);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(2048), dim3(256), 0, 0, dout, 1.0f, c_hi, c_lo, 2000);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms for %.1f G evaluations = %.2f G/s\n", mode == 0 ? "polynomial" : "hw sine   ", ms, 2048.0 * 256 * 8 * 2000 / 1e9, 2048.0 * 256 * 8 * 2000 / ms / 1e6);
    }
    return 0;
}
