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
                                                         int Cin, int T, int K, float slope_in, int apply_tanh,
                                                         const int* __restrict__ lens, int len_mul) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xl = smem;                 // [8][CP_S]
    float* wl = smem + 8 * CP_S;      // [Cin*K]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * CP_TT;
    const int pad = (K - 1) / 2;
    int Tv = T;  // valid columns of this item (ragged batch)
    if (lens) { const int l = lens[b] * len_mul; Tv = l < Tv ? l : Tv; }
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
            if (ch < Cin && t >= 0 && t < Tv) v = xb[(size_t)ch * T + t];
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
                            int T, int K, float slope_in, int apply_tanh, const int* lens, int len_mul,
                            hipStream_t stream) {
    if (K > 9 || K < 1 || (K & 1) == 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)(8 * CP_S + Cin * K) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    dim3 grid((unsigned)((T + CP_TT - 1) / CP_TT), (unsigned)B);
    hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), lds, stream, x, w_dev, bias_dev, y, Cin, T, K, slope_in,
                       apply_tanh, lens, len_mul);
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

// sin(x)^2 for Snake (snake.py:56-61).  The activation evaluates two sines per output sample and was
// sin-bound with the full-range libm sinf (Payne-Hanek branch + ~40 instructions): a 3-term Cody-Waite
// reduction by pi (exact k * PI_A for |k| < 2^16) and the odd Taylor polynomial through r^13 on
// [-pi/2, pi/2] costs 13 FMAs and stays within 1.1e-7 of the exact sine for |x| <= 1e5 (libm: 0.7e-7);
// the sign lost by reducing modulo pi does not matter under the square.  Larger arguments take sinf.
__device__ __forceinline__ float snake_sin2(float x) {
    if (fabsf(x) > 1.0e5f) { const float s = sinf(x); return s * s; }
    const float k = rintf(x * 0.31830988618379067f);
    float r = fmaf(-k, 3.140625f, x);
    r = fmaf(-k, 9.67502593994140625e-4f, r);
    r = fmaf(-k, 1.509957990978376432e-07f, r);
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

constexpr int A1_NTILE = 4;   // consecutive tiles per workgroup (next tile's inputs prefetched into registers)

__global__ __launch_bounds__(256) void act1d_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T,
                                                    const float* __restrict__ a_dev,
                                                    const float* __restrict__ invb_dev, const float* __restrict__ fu,
                                                    const float* __restrict__ fd, const int* __restrict__ lens,
                                                    int len_mul) {
    // staged window xl[i] = x[clamp(t0 - 8 + i)], i < A1_TT + 16 (starts 8 before the tile: 16-B aligned rows)
    __shared__ __attribute__((aligned(16))) float xl[A1_TT + 16];
    __shared__ __attribute__((aligned(16))) float sl[2 * A1_TT + 12];
    __shared__ float ful[12], fdl[12];
    const int tid = threadIdx.x;
    const int ntiles = (T + A1_TT - 1) / A1_TT;   // T = row stride (padded length)
    const int ngroups = (ntiles + A1_NTILE - 1) / A1_NTILE;
    const int bc = blockIdx.x / ngroups;
    const int c = bc % C;
    const int tile_first = (blockIdx.x - bc * ngroups) * A1_NTILE;
    // Tv = the utterance's own length: the replicate padding (resample.py:36-45, filter.py:92-99) clamps to
    // ITS last sample, so a padded batch equals the per-utterance results
    int Tv = T;
    if (lens) { const int l = lens[bc / C] * len_mul; Tv = l < Tv ? l : Tv; }
    const float a = a_dev[c], invb = invb_dev[c];
    const float* xr = x + (size_t)bc * T;
    float* yr = y + (size_t)bc * T;
    if (tid < 12) { ful[tid] = fu[tid]; fdl[tid] = fd[tid]; }
    const int twoT = 2 * Tv;
    const bool vec_rows = (T & 3) == 0;            // rows start 16-B aligned (x and y come from the workspace)
    // interior tile: no index is clamped anywhere in it, and its window can be loaded as aligned float4
    auto is_interior = [&](int t0) { return vec_rows && (t0 >= 8) && (t0 + A1_TT + 8 <= Tv); };

    float4 pre0, pre1;                             // prefetched window of the next interior tile
    bool have_pre = false;
    for (int tl = 0; tl < A1_NTILE; ++tl) {
        const int t0 = (tile_first + tl) * A1_TT;
        if (t0 >= Tv || tile_first + tl >= ntiles) break;      // nothing valid from here on (block-uniform)
        if (is_interior(t0)) {
            // ---- vector path: float4 global loads (next tile prefetched under this tile's arithmetic),
            //      3 + 5 ds_read_b128 per thread instead of 96 scalar LDS reads ----
            float4 v0, v1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (have_pre) { v0 = pre0; v1 = pre1; }
            else {
                v0 = *reinterpret_cast<const float4*>(xr + t0 - 8 + 4 * tid);
                if (tid < 4) v1 = *reinterpret_cast<const float4*>(xr + t0 - 8 + A1_TT + 4 * tid);
            }
            *reinterpret_cast<float4*>(&xl[4 * tid]) = v0;
            if (tid < 4) *reinterpret_cast<float4*>(&xl[A1_TT + 4 * tid]) = v1;
            const int tn = t0 + A1_TT;
            have_pre = (tl + 1 < A1_NTILE) && (tile_first + tl + 1 < ntiles) && is_interior(tn);
            if (have_pre) {
                pre0 = *reinterpret_cast<const float4*>(xr + tn - 8 + 4 * tid);
                if (tid < 4) pre1 = *reinterpret_cast<const float4*>(xr + tn - 8 + A1_TT + 4 * tid);
            }
            __syncthreads();
            for (int i0 = 8 * tid; i0 < 2 * A1_TT + 11; i0 += 8 * 256) {
                // Snake value i = i0 + e (n = 2*t0 + i - 5) needs x[((n + 15) >> 1) - k - 5], k = 0..5
                //   = xl[(i0 >> 1) + ((e + 10) >> 1) + 3 - k]: a 12-float window at xl[i0 >> 1]
                const int xb = i0 >> 1;                           // multiple of 4
                const float4 w0 = *reinterpret_cast<const float4*>(&xl[xb]);
                const float4 w1 = *reinterpret_cast<const float4*>(&xl[xb + 4]);
                const float4 w2 = *reinterpret_cast<const float4*>(&xl[xb + 8]);
                const float xw[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
                float sv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int par = e & 1;                        // (n + 15) & 1, i0 even
                    const int top = ((e + 10) >> 1) + 3;          // window index of k = 0
                    float u = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; ++k) u = fmaf(xw[top - k], ful[par + 2 * k], u);
                    u *= 2.f;
                    sv[e] = u + invb * snake_sin2(u * a);
                }
                if (i0 + 8 <= 2 * A1_TT + 12) {
                    *reinterpret_cast<float4*>(&sl[i0]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
                    *reinterpret_cast<float4*>(&sl[i0 + 4]) = make_float4(sv[4], sv[5], sv[6], sv[7]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (i0 + e < 2 * A1_TT + 12) sl[i0 + e] = sv[e];
                }
            }
            __syncthreads();
            {
                const int k0 = 4 * tid;                           // outputs k0..k0+3 need sl[2*k0 .. 2*k0 + 17]
                float sw[20];
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const float4 q = *reinterpret_cast<const float4*>(&sl[2 * k0 + 4 * v]);
                    sw[4 * v] = q.x; sw[4 * v + 1] = q.y; sw[4 * v + 2] = q.z; sw[4 * v + 3] = q.w;
                }
                float4 o;
                float* op = &o.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 12; ++j) acc = fmaf(fdl[j], sw[2 * e + j], acc);
                    op[e] = acc;
                }
                *reinterpret_cast<float4*>(yr + t0 + k0) = o;     // vec_rows: 16-B aligned
            }
        } else {
            // ---- edge tiles (first / last of a row, ragged ends, unaligned rows): clamp every index ----
            have_pre = false;
            for (int i = tid; i < A1_TT + 16; i += 256) {
                int t = t0 - 8 + i;
                t = t < 0 ? 0 : (t > Tv - 1 ? Tv - 1 : t);
                xl[i] = xr[t];
            }
            __syncthreads();
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
                    xi = xi < 0 ? 0 : (xi > Tv - 1 ? Tv - 1 : xi);
                    u = fmaf(xl[xi - (t0 - 8)], ful[par + 2 * k], u);
                }
                u *= 2.f;
                sl[i] = u + invb * snake_sin2(u * a);
            }
            __syncthreads();
            for (int k = tid; k < A1_TT; k += 256) {
                const int t = t0 + k;
                if (t < Tv) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 12; ++j) acc = fmaf(fdl[j], sl[2 * k + j], acc);
                    yr[t] = acc;
                }
            }
        }
        __syncthreads();   // xl / sl are rewritten by the next tile
    }
}

hipError_t launch_act1d(const float* x, float* y, int B, int C, int T, const float* a_dev, const float* invb_dev,
                        const float* filt_up12, const float* filt_dn12, const int* lens, int len_mul,
                        hipStream_t stream) {
    const int ntiles = (T + A1_TT - 1) / A1_TT;
    const int ngroups = (ntiles + A1_NTILE - 1) / A1_NTILE;
    dim3 grid((unsigned)((size_t)ngroups * (size_t)(B * C)));
    hipLaunchKernelGGL(act1d_kernel, grid, dim3(256), 0, stream, x, y, C, T, a_dev, invb_dev, filt_up12, filt_dn12,
                       lens, len_mul);
    return hipGetLastError();
}

// APNet head (apnet.py:379-383): pha = atan2(I, R); rea = exp(logamp) * cos(pha); imag = exp(logamp) * sin(pha)
__global__ __launch_bounds__(256) void apnet_polar_kernel(const float* __restrict__ logamp, const float* __restrict__ R,
                                                          const float* __restrict__ I, size_t n, float* __restrict__ pha,
                                                          float* __restrict__ rea, float* __restrict__ imag) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float p = atan2f(I[i], R[i]);
        const float m = expf(logamp[i]);
        float sn, cs;
        sincosf(p, &sn, &cs);
        pha[i] = p;
        rea[i] = m * cs;
        imag[i] = m * sn;
    }
}

hipError_t launch_apnet_polar(const float* logamp, const float* R, const float* I, size_t n, float* pha, float* rea,
                              float* imag, hipStream_t stream) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 65535u * 16u) blocks = 65535u * 16u;
    hipLaunchKernelGGL(apnet_polar_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, logamp, R, I, n, pha, rea, imag);
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

// ---------------------------------------------------------------------------------------------
// VITS posterior encoder / flow element-wise kernels.  All are one pass over [B, C, T] with
// time-contiguous float4-free coalesced access (T is arbitrary), grid = B*C rows x time blocks.
// ---------------------------------------------------------------------------------------------
#define AMP_ROW_LOOP(rows, T)                                             \
    const int tb = (T + 1023) / 1024;                                     \
    const int row = blockIdx.x / tb;                                      \
    const int tstart = (blockIdx.x - row * tb) * 1024 + threadIdx.x;      \
    const int tend = min(T, (int)((blockIdx.x - row * tb) * 1024 + 1024));
static inline unsigned row_grid(int rows, int T) { return (unsigned)((size_t)rows * ((T + 1023) / 1024)); }

// fused_add_tanh_sigmoid_multiply (utils/util.py:602-609) with the time-constant condition g_l:
// out[b,c,t] = tanh(a[b,c,t] + g[b,c]) * sigmoid(a[b,c+H,t] + g[b,c+H])     modules/flow/modules.py:141
__global__ __launch_bounds__(256) void wn_gate_kernel(const float* __restrict__ a, const float* __restrict__ cond,
                                                      long long cond_bs, float* __restrict__ out, int H, int T) {
    AMP_ROW_LOOP(B * H, T)
    const int b = row / H, c = row - b * H;
    const float gt = cond ? cond[(size_t)b * cond_bs + c] : 0.f;
    const float gs = cond ? cond[(size_t)b * cond_bs + c + H] : 0.f;
    const float* at = a + ((size_t)b * 2 * H + c) * T;
    const float* as = at + (size_t)H * T;
    float* o = out + (size_t)row * T;
    for (int t = tstart; t < tend; t += 256) {
        const float tv = tanhf(at[t] + gt);
        const float sv = 1.0f / (1.0f + expf(-(as[t] + gs)));
        o[t] = tv * sv;
    }
}

// x = (x + rs[:, :H]) * mask ; out (+)= rs[:, H:]      (last layer: out (+)= rs)   modules.py:146-151
__global__ __launch_bounds__(256) void wn_accumulate_kernel(float* __restrict__ x, float* __restrict__ out,
                                                            const float* __restrict__ rs,
                                                            const int* __restrict__ lens, int H, int T, int last,
                                                            int first) {
    AMP_ROW_LOOP(B * H, T)
    const int b = row / H, c = row - b * H;
    const int len = lens ? lens[b] : T;
    float* xr = x + (size_t)row * T;
    float* orow = out + (size_t)row * T;
    if (last) {
        const float* r = rs + ((size_t)b * H + c) * T;
        for (int t = tstart; t < tend; t += 256) orow[t] = (first ? 0.f : orow[t]) + r[t];
    } else {
        const float* r0 = rs + ((size_t)b * 2 * H + c) * T;
        const float* r1 = r0 + (size_t)H * T;
        for (int t = tstart; t < tend; t += 256) {
            xr[t] = (xr[t] + r0[t]) * (t < len ? 1.f : 0.f);
            orow[t] = (first ? 0.f : orow[t]) + r1[t];
        }
    }
}

// x *= sequence_mask(lens)                                  utils/util.py:618-622
__global__ __launch_bounds__(256) void mask_kernel(float* __restrict__ x, const int* __restrict__ lens, int C, int T) {
    AMP_ROW_LOOP(B * C, T)
    const int len = lens[row / C];
    float* xr = x + (size_t)row * T;
    for (int t = tstart; t < tend; t += 256)
        if (t >= len) xr[t] = 0.f;
        else xr[t] = xr[t] * 1.f;
}

// mean-only coupling (modules/flow/modules.py:379-397): forward x1 = m + x1*mask, reverse x1 = (x1 - m)*mask
__global__ __launch_bounds__(256) void coupling_kernel(float* __restrict__ x, const float* __restrict__ m,
                                                       const int* __restrict__ lens, int h, int T, int reverse) {
    AMP_ROW_LOOP(B * h, T)
    const int b = row / h, c = row - b * h;
    const int len = lens ? lens[b] : T;
    float* x1 = x + ((size_t)b * 2 * h + h + c) * T;
    const float* mr = m + (size_t)row * T;
    for (int t = tstart; t < tend; t += 256) {
        const float mk = t < len ? 1.f : 0.f;
        x1[t] = reverse ? (x1[t] - mr[t]) * mk : mr[t] + x1[t] * mk;
    }
}

// Flip: torch.flip(x, [1])                                   modules/flow/modules.py:314-321
__global__ __launch_bounds__(256) void flip_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T) {
    AMP_ROW_LOOP(B * C, T)
    const int b = row / C, c = row - b * C;
    const float* xr = x + ((size_t)b * C + (C - 1 - c)) * T;
    float* yr = y + (size_t)row * T;
    for (int t = tstart; t < tend; t += 256) yr[t] = xr[t];
}

// z = (m + eps * exp(logs)) * mask,  stats = [m ; logs]       models/tts/vits/vits.py:150-151
__global__ __launch_bounds__(256) void posterior_sample_kernel(const float* __restrict__ stats,
                                                               const float* __restrict__ eps,
                                                               const int* __restrict__ lens, float* __restrict__ z,
                                                               int C, int T) {
    AMP_ROW_LOOP(B * C, T)
    const int b = row / C, c = row - b * C;
    const int len = lens ? lens[b] : T;
    const float* mr = stats + ((size_t)b * 2 * C + c) * T;
    const float* lr = mr + (size_t)C * T;
    const float* er = eps + (size_t)row * T;
    float* zr = z + (size_t)row * T;
    for (int t = tstart; t < tend; t += 256) zr[t] = (mr[t] + er[t] * expf(lr[t])) * (t < len ? 1.f : 0.f);
}

hipError_t launch_wn_gate(const float* a, const float* cond, long long cond_bs, float* out, int B, int H, int T,
                          hipStream_t stream) {
    hipLaunchKernelGGL(wn_gate_kernel, dim3(row_grid(B * H, T)), dim3(256), 0, stream, a, cond, cond_bs, out, H, T);
    return hipGetLastError();
}
hipError_t launch_wn_accumulate(float* x, float* out, const float* rs, const int* lens, int B, int H, int T, int last,
                                int first, hipStream_t stream) {
    hipLaunchKernelGGL(wn_accumulate_kernel, dim3(row_grid(B * H, T)), dim3(256), 0, stream, x, out, rs, lens, H, T,
                       last, first);
    return hipGetLastError();
}
hipError_t launch_mask(float* x, const int* lens, int B, int C, int T, hipStream_t stream) {
    hipLaunchKernelGGL(mask_kernel, dim3(row_grid(B * C, T)), dim3(256), 0, stream, x, lens, C, T);
    return hipGetLastError();
}
hipError_t launch_coupling(float* x, const float* m, const int* lens, int B, int h, int T, int reverse,
                           hipStream_t stream) {
    hipLaunchKernelGGL(coupling_kernel, dim3(row_grid(B * h, T)), dim3(256), 0, stream, x, m, lens, h, T, reverse);
    return hipGetLastError();
}
hipError_t launch_flip_channels(const float* x, float* y, int B, int C, int T, hipStream_t stream) {
    hipLaunchKernelGGL(flip_kernel, dim3(row_grid(B * C, T)), dim3(256), 0, stream, x, y, C, T);
    return hipGetLastError();
}
hipError_t launch_posterior_sample(const float* stats, const float* eps, const int* lens, float* z, int B, int C, int T,
                                   hipStream_t stream) {
    hipLaunchKernelGGL(posterior_sample_kernel, dim3(row_grid(B * C, T)), dim3(256), 0, stream, stats, eps, lens, z, C, T);
    return hipGetLastError();
}

}  // namespace amp
