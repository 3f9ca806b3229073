// fp32 vector FMA rate on MI355X: scalar v_fma_f32 vs packed v_pk_fma_f32 (two fp32 per lane per instruction),
// independent chains, VGPR and SGPR-broadcast operands.  Bounds act1d_kernel's arithmetic.
//   hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: v_fma_f32 (16 chains), 1: v_pk_fma_f32 all-VGPR (8 chains x 2), 2: v_pk_fma_f32 with a uniform (SGPR) multiplier
__global__ __launch_bounds__(256) void fma_loop(const float* __restrict__ in, float* __restrict__ out, int iters, float m, float c) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[(tid + i * 97) & 65535];
    const float mv = in[tid & 1023] * 1e-3f + m, cv = in[(tid + 5) & 1023] * 1e-3f + c;   // per-lane multiplier / addend
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(mv), "v"(cv));   // the SLP vectoriser would pack plain C
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x2 p = {v[2 * i], v[2 * i + 1]};
                    p = MODE == 1 ? __builtin_elementwise_fma(p, (f32x2){mv, mv}, (f32x2){cv, cv})
                                  : __builtin_elementwise_fma(p, (f32x2){m, m}, (f32x2){c, c});
                    v[2 * i] = p.x; v[2 * i + 1] = p.y;
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[tid] = s;
}

int main() {
    const int blocks = 256 * 16, iters = 4096;
    float *d, *o;
    hipMalloc(&d, 65536 * 4); hipMalloc(&o, (size_t)blocks * 256 * 4);
    float h[65536];
    for (int i = 0; i < 65536; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"v_fma_f32 (VGPR operands)", "v_pk_fma_f32 (VGPR operands)", "v_pk_fma_f32 (SGPR multiplier/addend)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(fma_loop<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters, 0.999f, 0.001f);
            if (mode == 1) hipLaunchKernelGGL(fma_loop<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters, 0.999f, 0.001f);
            if (mode == 2) hipLaunchKernelGGL(fma_loop<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double)blocks * 256 * iters * 4 * 16;
            if (rep == 2) printf("%-42s %.3f ms  %.1f TFLOP/s fp32 (%.1f %% of 157.3)\n", names[mode], ms, 2 * fma / ms / 1e9, 100.0 * 2 * fma / ms / 1e9 / 157.3);
        }
    }
    return 0;
}
