// Bandwidth-bound kernels of the generator that are not GEMM-shaped (gfx950).
//   conv_post   : Conv1d(C -> 1, k7) + tanh            hifigan.py:215-217, bigvgan.py:328-329
//   act1d       : Activation1d(Snake|SnakeBeta)         modules/anti_aliasing/act.py:31-36
//   add_channel_bias : x + cond(g) for a length-1 g      hifigan.py:426-427
#include "amp_internal.h"

namespace amp {

// ---------------------------------------------------------------------------------------------
// conv_post: one output channel.  256 threads x 4 consecutive outputs; 8 input channels staged
// per step in LDS ([8][1024 + 16], 16-B aligned rows so every lane reads 3 x ds_read_b128).
// ---------------------------------------------------------------------------------------------
constexpr int CP_TT = 1024;
constexpr int CP_S = CP_TT + 16;

__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         int Cin, int T, int K, float slope_in, int apply_tanh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xl = smem;                 // [8][CP_S]
    float* wl = smem + 8 * CP_S;      // [Cin*K]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * CP_TT;
    const int pad = (K - 1) / 2;
    for (int i = tid; i < Cin * K; i += 256) wl[i] = w[i];
    float acc[4];
    const float b0 = bias ? bias[0] : 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = b0;
    const float* xb = x + (size_t)b * Cin * T;
    for (int c0 = 0; c0 < Cin; c0 += 8) {
        __syncthreads();
        for (int idx = tid; idx < 8 * CP_S; idx += 256) {
            const int row = idx / CP_S, col = idx - row * CP_S;
            const int t = t0 - pad + col;
            const int ch = c0 + row;
            float v = 0.f;
            if (ch < Cin && t >= 0 && t < T) v = xb[(size_t)ch * T + t];
            xl[idx] = v > 0.f ? v : v * slope_in;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (c0 + r < Cin) {
                const float4* rowp = reinterpret_cast<const float4*>(xl + r * CP_S + 4 * tid);
                const float4 v0 = rowp[0], v1 = rowp[1], v2 = rowp[2];
                const float v[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
                const float* wr = wl + (c0 + r) * K;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    if (j < K) {
                        const float wj = wr[j];
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[o] = fmaf(wj, v[o + j], acc[o]);
                    }
                }
            }
        }
    }
    float* yb = y + (size_t)b * T;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int t = t0 + 4 * tid + o;
        if (t < T) yb[t] = apply_tanh ? tanhf(acc[o]) : acc[o];
    }
}

hipError_t launch_conv_post(const float* x, const float* w_dev, const float* bias_dev, float* y, int B, int Cin,
                            int T, int K, float slope_in, int apply_tanh, hipStream_t stream) {
    if (K > 9 || K < 1 || (K & 1) == 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)(8 * CP_S + Cin * K) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    dim3 grid((unsigned)((T + CP_TT - 1) / CP_TT), (unsigned)B);
    hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), lds, stream, x, w_dev, bias_dev, y, Cin, T, K, slope_in,
                       apply_tanh);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// act1d: y = down2( snake( up2(x) ) ) with replicate padding, one pass, nothing leaves the CU between
// the three steps (the reference runs them as three separate ops: 5 tensor round trips).
//   u[n]  = 2 * sum_m xp[m] * f[n + 15 - 2m],  xp[m] = x[clamp(m - 5)]         (resample.py:36-45)
//   s[n]  = u + invb * sin(a*u)^2                                               (snake.py:56-61)
//   y[t]  = sum_j f[j] * s[clamp(2t + j - 5, 0, 2T-1)]                          (filter.py:92-99)
// ---------------------------------------------------------------------------------------------
constexpr int A1_TT = 1024;

__global__ __launch_bounds__(256) void act1d_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T,
                                                    const float* __restrict__ a_dev,
                                                    const float* __restrict__ invb_dev, const float* __restrict__ fu,
                                                    const float* __restrict__ fd) {
    __shared__ float xl[A1_TT + 12];
    __shared__ float sl[2 * A1_TT + 12];
    __shared__ float ful[12], fdl[12];
    const int tid = threadIdx.x;
    const int ntiles = (T + A1_TT - 1) / A1_TT;
    const int bc = blockIdx.x / ntiles;
    const int c = bc % C;
    const int t0 = (blockIdx.x - bc * ntiles) * A1_TT;
    const float a = a_dev[c], invb = invb_dev[c];
    const float* xr = x + (size_t)bc * T;
    if (tid < 12) { ful[tid] = fu[tid]; fdl[tid] = fd[tid]; }
    for (int i = tid; i < A1_TT + 12; i += 256) {
        int t = t0 - 6 + i;
        t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
        xl[i] = xr[t];
    }
    __syncthreads();
    const int twoT = 2 * T;
    for (int i = tid; i < 2 * A1_TT + 11; i += 256) {
        int n = 2 * t0 + i - 5;
        n = n < 0 ? 0 : (n > twoT - 1 ? twoT - 1 : n);
        const int np = n + 15;
        const int mmax = np >> 1;
        const int par = np & 1;
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int xi = mmax - k - 5;  // index into x before clamping
            xi = xi < 0 ? 0 : (xi > T - 1 ? T - 1 : xi);
            u = fmaf(xl[xi - (t0 - 6)], ful[par + 2 * k], u);
        }
        u *= 2.f;
        const float sn = sinf(u * a);
        sl[i] = u + invb * (sn * sn);
    }
    __syncthreads();
    float* yr = y + (size_t)bc * T;
    for (int k = tid; k < A1_TT; k += 256) {
        const int t = t0 + k;
        if (t < T) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) acc = fmaf(fdl[j], sl[2 * k + j], acc);
            yr[t] = acc;
        }
    }
}

hipError_t launch_act1d(const float* x, float* y, int B, int C, int T, const float* a_dev, const float* invb_dev,
                        const float* filt_up12, const float* filt_dn12, hipStream_t stream) {
    dim3 grid((unsigned)(((T + A1_TT - 1) / A1_TT) * (size_t)(B * C)));
    hipLaunchKernelGGL(act1d_kernel, grid, dim3(256), 0, stream, x, y, C, T, a_dev, invb_dev, filt_up12, filt_dn12);
    return hipGetLastError();
}

__global__ void add_channel_bias_kernel(float* __restrict__ y, const float* __restrict__ cb, int T) {
    const int bc = blockIdx.x;
    const float v = cb[bc];
    float* yr = y + (size_t)bc * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) yr[t] += v;
}

hipError_t launch_add_channel_bias(float* y, const float* cb, int B, int C, int T, hipStream_t stream) {
    dim3 grid((unsigned)(B * C));
    hipLaunchKernelGGL(add_channel_bias_kernel, grid, dim3(256), 0, stream, y, cb, T);
    return hipGetLastError();
}

}  // namespace amp
