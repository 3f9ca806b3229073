// Arithmetic shared by act1d_kernel (small_kernels.hip) and the conv kernel's fused activation epilogue
// (conv_f16x3.hip): Activation1d = 2x up-sampling FIR -> Snake -> 2x down-sampling FIR
// (modules/anti_aliasing/act.py:31-36, resample.py:36-65, filter.py:92-99, modules/activation_functions/snake.py:51-61).
// Both kernels run exactly these operation sequences, so they agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace amp {

// sin(x)^2 for Snake (snake.py:56-61).  The activation evaluates two sines per output sample and was
// sin-bound with the full-range libm sinf (Payne-Hanek branch + ~40 instructions): a 3-term Cody-Waite
// reduction by pi (exact k * PI_A for |k| < 2^16) and the odd Taylor polynomial through r^13 on
// [-pi/2, pi/2] costs 13 FMAs and stays within 1.1e-7 of the exact sine for |x| <= 1e5 (libm: 0.7e-7);
// the sign lost by reducing modulo pi does not matter under the square.  Larger arguments are reduced in
// fp64 (snake_reduce_slow: a dozen instructions, so that the whole-AMPBlock kernel can carry the rare path
// inline) and then take the same polynomial: every kernel of the library evaluates exactly this function.
__device__ __forceinline__ float snake_reduce_slow(float x) {
    const double xd = (double)x;
    const double k = __builtin_rint(xd * 0.31830988618379067154);
    double r = __builtin_fma(-k, 3.14159265358979311600, xd);     // pi rounded to fp64 ...
    r = __builtin_fma(-k, 1.22464679914735317723e-16, r);         // ... and what that rounding dropped
    const float rf = (float)r;
    // |x| beyond ~1e15 has no phase left in fp64 either: 0 for finite arguments, NaN for inf / NaN (as sin does)
    return __builtin_fabsf(rf) <= 1.6f ? rf : (x - x);
}

__device__ __forceinline__ float snake_sin2(float x) {
    const float k = rintf(x * 0.31830988618379067f);
    float r = fmaf(-k, 3.140625f, x);
    r = fmaf(-k, 9.67502593994140625e-4f, r);
    r = fmaf(-k, 1.509957990978376432e-07f, r);
    if (__builtin_expect(fabsf(x) > 1.0e5f, 0)) r = snake_reduce_slow(x);
    const float r2 = r * r;
    float p = 1.0f / 6227020800.0f;
    p = fmaf(p, r2, -1.0f / 39916800.0f);
    p = fmaf(p, r2, 1.0f / 362880.0f);
    p = fmaf(p, r2, -1.0f / 5040.0f);
    p = fmaf(p, r2, 1.0f / 120.0f);
    p = fmaf(p, r2, -1.0f / 6.0f);
    const float sn = fmaf(r * r2, p, r);
    return sn * sn;
}

// Two fp32 lanes per instruction (v_pk_fma_f32 / v_pk_mul_f32: the 157 TFLOP/s vector rate needs them).  Every
// half goes through exactly the scalar operation sequence, so packed and scalar paths agree bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return (f32x2){v, v}; }

// sin(x)^2 of snake_sin2's fast path on four pairs (x = a*u already formed by the caller).  Written step by step
// across the four independent chains so that dependent packed ops are never back to back.
__device__ __forceinline__ void snake_sin2_pk4(const f32x2 (&x)[4], f32x2 (&out)[4]) {
    f32x2 k[4], r[4], r2[4], p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) k[q] = __builtin_elementwise_rint(x[q] * 0.31830988618379067f);
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = pk_fma(-k[q], pk_splat(3.140625f), x[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = pk_fma(-k[q], pk_splat(9.67502593994140625e-4f), r[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = pk_fma(-k[q], pk_splat(1.509957990978376432e-07f), r[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) r2[q] = r[q] * r[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = pk_fma(pk_splat(1.0f / 6227020800.0f), r2[q], pk_splat(-1.0f / 39916800.0f));
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = pk_fma(p[q], r2[q], pk_splat(1.0f / 362880.0f));
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = pk_fma(p[q], r2[q], pk_splat(-1.0f / 5040.0f));
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = pk_fma(p[q], r2[q], pk_splat(1.0f / 120.0f));
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = pk_fma(p[q], r2[q], pk_splat(-1.0f / 6.0f));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 sn = pk_fma(r[q] * r2[q], p[q], r[q]);
        out[q] = sn * sn;
    }
}

// LDS position of Snake value i.  Lanes write / read `sl` as float4 at a stride of 8 floats (32 B): every such access
// is a 2-way bank conflict (ds_read_b128 serves 16-lane groups {0-3, 12-15, 20-27}, ... over 64 banks, ds_write_b128
// 8 contiguous lanes over 32; rocprofv3, visit U: 56 % of the kernel's LDS cycles were conflict cycles).  Flipping the
// low bit of the float4 index with the parity of its bits 3 and 4 makes all seven accesses of a tile conflict-free
// (enumerated over the hardware's lane groups, tests/experiments/act1d_swizzle.py).
__device__ __forceinline__ int sl_pos4(int j4) { return j4 ^ (((j4 >> 3) ^ (j4 >> 4)) & 1); }     // float4 index
__device__ __forceinline__ int sl_pos(int i) { return (sl_pos4(i >> 2) << 2) | (i & 3); }         // float index


}  // namespace amp
