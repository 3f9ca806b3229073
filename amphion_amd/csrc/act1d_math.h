// Arithmetic shared by act1d_kernel (small_kernels.hip) and the conv kernel's fused activation epilogue
// (conv_f16x3.hip): Activation1d = 2x up-sampling FIR -> Snake -> 2x down-sampling FIR
// (modules/anti_aliasing/act.py:31-36, resample.py:36-65, filter.py:92-99, modules/activation_functions/snake.py:51-61).
// Both kernels run exactly these operation sequences, so they agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace amp {

// sin(x)^2 for Snake (snake.py:56-61), x = alpha * u already rounded to fp32 as the reference rounds it.
// Round 4: the hardware sine behind an exact range reduction.  v_sin_f32 takes revolutions and is accurate to ~2e-7 on a reduced
// argument; what made it useless in round 2's probe was the fp32 rounding of x / 2pi.  With 1 / 2pi split into hi + lo:
//     k = rint(x * hi);   f = fma(x, hi, -k)  (the exact product minus k: one rounding, |f| <= 0.5);   f = fma(x, lo, f)
// f is the phase in revolutions to ~1e-8 for every finite x up to ~1e8 (beyond, and for inf / NaN, the result is a bounded
// garbage value / NaN, as in the reference), and sin(x)^2 = v_sin(f)^2 stays within 3.3e-7 of the exact value over all of that
// range (tests/experiments/vsin_snake.hip; the reduction by pi + Taylor polynomial it replaces: 1.1e-7 up to |x| = 1e5 and an fp64
// fallback beyond).  5 ordinary instructions + 1 quarter-rate transcendental instead of 14, no large-argument branch: the evaluation
// is 1.5x faster, and the activation -- VALU-bound everywhere it runs -- about 15 %.  Every kernel of the library evaluates exactly
// this function, so they agree bit for bit.
constexpr float kInv2PiHi = 0.15915493667125702f;      // fp32(1 / 2pi)
constexpr float kInv2PiLo = 6.4206382432985265e-09f;   // fp32(1 / 2pi - kInv2PiHi)

__device__ __forceinline__ float snake_sin2(float x) {
    const float k = rintf(x * kInv2PiHi);
    float f = fmaf(x, kInv2PiHi, -k);
    f = fmaf(x, kInv2PiLo, f);
    const float s = __builtin_amdgcn_sinf(f);
    return s * s;
}

// Two fp32 lanes per instruction where the ISA has them (v_pk_fma_f32 / v_pk_mul_f32).  Every half goes through exactly the scalar
// operation sequence, so packed and scalar paths agree bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return (f32x2){v, v}; }

// snake_sin2 on four pairs (x = a*u already formed by the caller), step by step across the four independent chains.
__device__ __forceinline__ void snake_sin2_pk4(const f32x2 (&x)[4], f32x2 (&out)[4]) {
    f32x2 k[4], f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) k[q] = __builtin_elementwise_rint(x[q] * kInv2PiHi);
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = pk_fma(x[q], pk_splat(kInv2PiHi), -k[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = pk_fma(x[q], pk_splat(kInv2PiLo), f[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 sn = {__builtin_amdgcn_sinf(f[q].x), __builtin_amdgcn_sinf(f[q].y)};
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
