// Bandwidth-bound kernels of the generator that are not GEMM-shaped (gfx950).
//   conv_post   : Conv1d(C -> 1, k7) + tanh            hifigan.py:215-217, bigvgan.py:328-329
//   act1d       : Activation1d(Snake|SnakeBeta)         modules/anti_aliasing/act.py:31-36
//   add_channel_bias : x + cond(g) for a length-1 g      hifigan.py:426-427
#include "amp_internal.h"
#include "act1d_math.h"
#include <stdlib.h>

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
    if (t0 >= Tv) return;   // the whole block lies beyond the utterance (block-uniform, before any barrier)
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

// Streaming form for rows that are 16-B aligned (T % 4 == 0: every tensor of a generator forward): no LDS, no barrier.
// A lane owns 4 consecutive outputs and fetches, per input channel, the three aligned float4 around them (the
// neighbouring lanes' copies come out of L1); 8 channels = 24 loads are in flight before the first is used.  Per
// output the channels and taps are accumulated in the order of conv_post_kernel, so both give the same bits.  The
// weights are wave-uniform (scalar loads).
template <int K>
__global__ __launch_bounds__(256) void conv_post_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int Cin, int T, float slope_in, int apply_tanh,
                                                                const int* __restrict__ lens, int len_mul) {
    constexpr int PAD = (K - 1) / 2;
    const int b = blockIdx.y;
    const int t = blockIdx.x * CP_TT + 4 * (int)threadIdx.x;      // first of this lane's 4 outputs
    if (t >= T) return;
    int Tv = T;
    if (lens) { const int l = lens[b] * len_mul; Tv = l < Tv ? l : Tv; }
    if (t >= Tv) return;                                          // beyond the utterance: unspecified samples (no barrier in this kernel)
    const float* xb = x + (size_t)b * Cin * T;
    const float b0 = bias ? bias[0] : 0.f;
    float acc[4] = {b0, b0, b0, b0};
    // clamped (always in-row) addresses of the left / right neighbours; their values are zeroed below when out of range
    const int tl = t >= 4 ? t - 4 : t;
    const int tr = t + 4 < T ? t + 4 : t;
    const bool lok = t >= 4, rok = t + 4 < T;
    const bool inner = lok && (t + 8 <= Tv);                      // all 12 positions valid: no per-element masks
    for (int c0 = 0; c0 < Cin; c0 += 8) {
        float4 vl[8], vc[8], vr[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ch = c0 + r < Cin ? c0 + r : Cin - 1;       // scalar clamp: loads stay unconditional
            const float* row = xb + (size_t)ch * T;
            vl[r] = *reinterpret_cast<const float4*>(row + tl);
            vc[r] = *reinterpret_cast<const float4*>(row + t);
            vr[r] = *reinterpret_cast<const float4*>(row + tr);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (c0 + r < Cin) {                                   // wave-uniform
                float v[12] = {vl[r].x, vl[r].y, vl[r].z, vl[r].w, vc[r].x, vc[r].y, vc[r].z, vc[r].w,
                               vr[r].x, vr[r].y, vr[r].z, vr[r].w};
                if (!inner) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int ti = t - 4 + i;
                        const bool ok = (i >= 4 || lok) && (i < 8 || rok) && ti < Tv;
                        v[i] = ok ? v[i] : 0.f;
                    }
                }
#pragma unroll
                for (int i = 4 - PAD; i < 8 + PAD; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope_in;
                const float* wr = w + (size_t)(c0 + r) * K;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float wj = wr[j];
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = fmaf(wj, v[4 - PAD + o + j], acc[o]);
                }
            }
        }
    }
    float4 o4;
    o4.x = apply_tanh ? tanhf(acc[0]) : acc[0];
    o4.y = apply_tanh ? tanhf(acc[1]) : acc[1];
    o4.z = apply_tanh ? tanhf(acc[2]) : acc[2];
    o4.w = apply_tanh ? tanhf(acc[3]) : acc[3];
    *reinterpret_cast<float4*>(y + (size_t)b * T + t) = o4;
}

hipError_t launch_conv_post(const float* x, const float* w_dev, const float* bias_dev, float* y, int B, int Cin,
                            int T, int K, float slope_in, int apply_tanh, const int* lens, int len_mul,
                            hipStream_t stream) {
    if (K > 9 || K < 1 || (K & 1) == 0) return hipErrorInvalidValue;
    const bool aligned = (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (aligned && (K == 7 || K == 3 || K == 5)) {
        dim3 grid((unsigned)((T + CP_TT - 1) / CP_TT), (unsigned)B);
        note_kernel("conv_post_stream_kernel", K);
        note_work((unsigned long long)grid.x * grid.y, 2.0 * Cin * K * (double)T * B / 1e9, 4.0 * B * (double)T * (Cin + 1) / 1e6,
                  "conv_post %d->1 k=%d T=%d B=%d%s", Cin, K, T, B, apply_tanh ? " + tanh" : "");
        if (K == 7) hipLaunchKernelGGL(conv_post_stream_kernel<7>, grid, dim3(256), 0, stream, x, w_dev, bias_dev, y, Cin, T, slope_in, apply_tanh, lens, len_mul);
        else if (K == 5) hipLaunchKernelGGL(conv_post_stream_kernel<5>, grid, dim3(256), 0, stream, x, w_dev, bias_dev, y, Cin, T, slope_in, apply_tanh, lens, len_mul);
        else hipLaunchKernelGGL(conv_post_stream_kernel<3>, grid, dim3(256), 0, stream, x, w_dev, bias_dev, y, Cin, T, slope_in, apply_tanh, lens, len_mul);
        return hipGetLastError();
    }
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
constexpr int A1_TT = 1024;   // outputs per workgroup tile (grid granularity) ...
constexpr int A1_WT = 256;    // ... of which every WAVE owns a quarter, with a window and Snake buffer of its own (round 3)

// NTILE consecutive 1024-output tiles per workgroup.  Round 3: the four waves of a workgroup are INDEPENDENT -- wave w handles outputs
// [256 w, 256 w + 256) of every tile from its own LDS slices (window 256 + 16 floats, Snake values 512 + 16), so nothing in the tile
// loop is a workgroup barrier: LDS operations of one wave execute in order, a wave only needs its own earlier writes.  Rounds 1-2
// staged one 1024-output tile per workgroup behind three __syncthreads per tile and the SQ counters showed the waves parked 58 % of
// their life (profiles/r2_uv_act1d.txt); the price of independence is a 16-float window halo and 10 Snake values per 256 outputs
// instead of per 1024.  Per element the operation sequence is unchanged (act1d_math.h): same bits.
// When all tiles lie inside the row their windows are requested at kernel entry by straight-line code (NTILE <= 4); otherwise a rolled
// loop with the next tile's window in flight.  First / last / ragged wave tiles stage a clamped window and then run the same packed
// arithmetic, with the out-of-range Snake values replicated afterwards.
// (94 registers in the strip forms = 5 waves per SIMD.  A budget for 6 / 8 waves spills 5 / 14 registers and is slower: C3 26.5 ->
// 28.1 / 28.7 ms, profiles/r3_tu_act1d_waves.txt.)
template <int NTILE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void act1d_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T,
                                                    const float* __restrict__ a_dev,
                                                    const float* __restrict__ invb_dev, const float* __restrict__ fu,
                                                    const float* __restrict__ fd, const int* __restrict__ lens,
                                                    int len_mul, int rev) {
    // per wave: staged window xl[i] = x[clamp(t0w - 8 + i)], i < A1_WT + 16 (starts 8 before the wave tile: 16-B aligned rows)
    __shared__ __attribute__((aligned(16))) float xl_all[4][A1_WT + 16];
    __shared__ __attribute__((aligned(16))) float sl_all[4][2 * A1_WT + 16];   // swizzled (sl_pos): whole float4 pairs
    __shared__ float ful_all[4][12];               // up taps for the one-value-at-a-time paths: indexed by parity
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const xl = xl_all[wave];
    float* const sl = sl_all[wave];
    float* const ful = ful_all[wave];
    const int ntiles = (T + A1_TT - 1) / A1_TT;   // T = row stride (padded length)
    const int ngroups = (ntiles + NTILE - 1) / NTILE;
    // rev: descending workgroup order (this launch starts on what the previous one wrote last, ConvArgs::rev)
    const int blk = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int bc = blk / ngroups;
    const int c = bc % C;
    const int tile_first = (blk - bc * ngroups) * NTILE;
    // Tv = the utterance's own length: the replicate padding (resample.py:36-45, filter.py:92-99) clamps to
    // ITS last sample, so a padded batch equals the per-utterance results
    int Tv = T;
    if (lens) { const int l = lens[bc / C] * len_mul; Tv = l < Tv ? l : Tv; }
    const float a = a_dev[c], invb = invb_dev[c];
    const float* xr = x + (size_t)bc * T;
    float* yr = y + (size_t)bc * T;
    // the 12 + 12 taps are wave-uniform: scalar loads, no LDS reads in the packed path.  The up filter carries
    // the x2 gain of UpSample1d (resample.py:41; a power of two, so folding it is exact).
    float fu2[12], fdr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { fu2[k] = 2.f * fu[k]; fdr[k] = fd[k]; }
    if (lane < 12) ful[lane] = 2.f * fu[lane];
    const int twoT = 2 * Tv;
    const bool vec_rows = (T & 3) == 0;            // rows start 16-B aligned (x and y come from the workspace)
    const int woff = A1_WT * wave;                 // this wave's quarter of a workgroup tile
    // interior wave tile: no index is clamped anywhere in it, and its window can be loaded as aligned float4
    auto is_interior = [&](int t0) { return vec_rows && (t0 >= 8) && (t0 + A1_WT + 8 <= Tv); };
    auto tile_live = [&](int tl) { return tl < NTILE && tile_first + tl < ntiles && (tile_first + tl) * A1_TT + woff < Tv; };
    // one Snake value from the staged window: s-index i of the wave tile has its k = 0 tap at xl[top], parity par
    auto snake_scalar = [&](int i) {
        const int top = ((i + 10) >> 1) + 3, par = i & 1;
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) u = fmaf(xl[top - k], ful[par + 2 * k], u);
        return fmaf(invb, snake_sin2(u * a), u);
    };
    auto load_window = [&](int tl, float4& w0, float4& w1) {      // issue the global loads of an interior wave tile
        const int t0 = (tile_first + tl) * A1_TT + woff;
        if (tile_live(tl) && is_interior(t0)) {
            w0 = *reinterpret_cast<const float4*>(xr + t0 - 8 + 4 * lane);
            if (lane < 4) w1 = *reinterpret_cast<const float4*>(xr + t0 - 8 + A1_WT + 4 * lane);
        }
    };

    // ---- one wave tile, in two steps: the window into LDS (v0 / v1 = the window when the tile is interior) ... ----
    auto stage_window = [&](const int t0, const bool interior, const float4 v0, const float4 v1) __attribute__((always_inline)) {
        if (interior) {
            *reinterpret_cast<float4*>(&xl[4 * lane]) = v0;
            if (lane < 4) *reinterpret_cast<float4*>(&xl[A1_WT + 4 * lane]) = v1;
        } else {
            // the replicate-padded window itself: every tap below then reads x[clamp(.)] as resample.py:36 does
            for (int i = lane; i < A1_WT + 16; i += 64) {
                int t = t0 - 8 + i;
                t = t < 0 ? 0 : (t > Tv - 1 ? Tv - 1 : t);
                xl[i] = xr[t];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // ---- ... and the arithmetic from there to the store ----
    auto compute_tile = [&](const int t0, const bool interior) __attribute__((always_inline)) {
        {
            // Snake values i = 8*lane + e, e < 8 (n = 2*t0 + i - 5) need x[((n + 15) >> 1) - k - 5], k = 0..5
            //   = xl[4*lane + ((e + 10) >> 1) + 3 - k]: a 12-float window at xl[4*lane].  Values e = 2q, 2q+1
            // share their six x taps and take the even / odd phase of the filter: one packed chain.
            const int xb = 4 * lane;
            const float4 w0 = *reinterpret_cast<const float4*>(&xl[xb]);
            const float4 w1 = *reinterpret_cast<const float4*>(&xl[xb + 4]);
            const float4 w2 = *reinterpret_cast<const float4*>(&xl[xb + 8]);
            const float xw[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
            f32x2 uv[4], xa[4], sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int top = q + 8;                        // ((2q + 10) >> 1) + 3
                f32x2 u = pk_splat(0.f);
#pragma unroll
                for (int k = 0; k < 6; ++k) u = pk_fma(pk_splat(xw[top - k]), (f32x2){fu2[2 * k], fu2[2 * k + 1]}, u);
                uv[q] = u;
                xa[q] = u * a;
            }
            snake_sin2_pk4(xa, sv);
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = pk_fma(pk_splat(invb), sv[q], uv[q]);
            *reinterpret_cast<float4*>(&sl[4 * sl_pos4(2 * lane)]) = make_float4(sv[0].x, sv[0].y, sv[1].x, sv[1].y);
            *reinterpret_cast<float4*>(&sl[4 * sl_pos4(2 * lane + 1)]) = make_float4(sv[2].x, sv[2].y, sv[3].x, sv[3].y);
            // the 10 values past the 512 (the down filter's right halo): one lane each
            if (lane >= 64 - 10) {
                const int i = 2 * A1_WT + (lane - (64 - 10));
                sl[sl_pos(i)] = snake_scalar(i);
            }
        }
        if (!interior) {
            // DownSample1d pads the SNAKE OUTPUT by replication (filter.py:92-99): values whose n = 2*t0 + i - 5
            // lies outside [0, 2*Tv - 1] are copies of the first / last valid one, not filter results
            __builtin_amdgcn_wave_barrier();
            const int ilo = 5 - 2 * t0;                       // i of n = 0
            const int ihi = twoT + 4 - 2 * t0;                // i of n = 2*Tv - 1  (>= 6: t0 < Tv)
            const float slo = sl[sl_pos(ilo > 0 ? ilo : 0)];
            const float shi = sl[sl_pos(ihi < 2 * A1_WT + 9 ? ihi : 2 * A1_WT + 9)];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = 8 * lane + e;
                if (i < ilo) sl[sl_pos(i)] = slo;
                else if (i > ihi) sl[sl_pos(i)] = shi;
            }
            if (lane >= 64 - 10) {
                const int i = 2 * A1_WT + (lane - (64 - 10));
                if (i > ihi) sl[sl_pos(i)] = shi;
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int k0 = 4 * lane;                          // outputs k0..k0+3 need sl[2*k0 .. 2*k0 + 17]
            f32x2 sw[10];
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float4 q = *reinterpret_cast<const float4*>(&sl[4 * sl_pos4(2 * lane + v)]);
                sw[2 * v] = (f32x2){q.x, q.y};
                sw[2 * v + 1] = (f32x2){q.z, q.w};
            }
            float4 o;
            float* op = &o.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 acc = pk_splat(0.f);                    // (even taps, odd taps): two FMA chains, added below
#pragma unroll
                for (int m = 0; m < 6; ++m) acc = pk_fma((f32x2){fdr[2 * m], fdr[2 * m + 1]}, sw[e + m], acc);
                op[e] = acc.x + acc.y;
            }
            if (interior || (vec_rows && t0 + k0 + 4 <= Tv)) {
                *reinterpret_cast<float4*>(yr + t0 + k0) = o; // vec_rows: 16-B aligned
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t0 + k0 + e < Tv) yr[t0 + k0 + e] = op[e];
            }
        }
        __builtin_amdgcn_wave_barrier();   // xl / sl are rewritten by this wave's next tile
    };
    auto process_tile = [&](const int t0, const bool interior, const float4 v0, const float4 v1) __attribute__((always_inline)) {
        stage_window(t0, interior, v0, v1);
        compute_tile(t0, interior);
    };

    // do all NTILE tiles of this workgroup exist and lie inside the row?  (per wave: its quarter of the first and the last tile)
    // NTILE >= 8: always the rolled loop -- a workgroup lives long enough (8+ tiles) for its start-up (scalar loads of
    // the taps, the first window's latency: 40 % of a 2-tile workgroup's life, visit U) to stop mattering, with the
    // next window always in flight
    const bool all_interior = NTILE <= 4 && tile_first + NTILE <= ntiles && is_interior(tile_first * A1_TT + woff) &&
                              is_interior((tile_first + NTILE - 1) * A1_TT + woff);
    if (all_interior) {
        // straight-line path: every window requested now, by every lane (lanes >= 4 repeat the halo address of
        // lane & 3, so no load is conditional and none has to be waited for before its tile comes up)
        float4 w0[NTILE], w1[NTILE];
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl) {
            const float* wp = xr + (tile_first + tl) * A1_TT + woff - 8;
            w0[tl] = *reinterpret_cast<const float4*>(wp + 4 * lane);
            w1[tl] = *reinterpret_cast<const float4*>(wp + A1_WT + 4 * (lane & 3));
            // keep the requests together at the top (the machine scheduler would otherwise interleave them with
            // address arithmetic).  hipcc still sinks tile 0's pair below the others, to its first use: harmless,
            // loads return in issue order and tile 0 has to wait one full latency either way.
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl) process_tile((tile_first + tl) * A1_TT + woff, true, w0[tl], w1[tl]);
    } else {
        // Rolled strip.  Runs of interior tiles go through a loop of their own whose body is: window registers -> LDS, the NEXT
        // window's loads into the same registers, arithmetic, ONE store.  On this hardware stores count in vmcnt like loads, so
        // the wait for a window at the top of the loop is "all but the one younger store" -- the previous form (next window
        // requested before the current one was staged, so the registers were copied at the loop's latch) waited for vmcnt(0)
        // there and parked every wave on the acknowledgement of its own store once per tile.  The row's first / last / ragged
        // tiles (clamped windows, conditional stores) stay outside that loop, one at a time.
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int tl = 0;
        while (tl < NTILE) {
            if (tile_first + tl >= ntiles) break;             // no tile left in the row (block-uniform)
            int t0 = (tile_first + tl) * A1_TT + woff;
            if (t0 >= Tv) break;                              // nothing valid for this wave from here on (wave-uniform)
            if (!is_interior(t0)) { process_tile(t0, false, z4, z4); ++tl; continue; }
            const float* wp = xr + t0 - 8;
            float4 p0 = *reinterpret_cast<const float4*>(wp + 4 * lane);
            float4 p1 = *reinterpret_cast<const float4*>(wp + A1_WT + 4 * (lane & 3));     // (every lane: no conditional load)
            // (the run's first tile is peeled: every way into the loop below then has the same memory operations behind it --
            //  two loads, one store -- and the compiler's wait counts come out exact instead of the conservative zero)
            auto next_is_interior = [&]() { return tl + 1 < NTILE && tile_first + tl + 1 < ntiles && is_interior(t0 + A1_TT); };
            auto request_next = [&]() {                       // in flight under this tile's arithmetic
                p0 = *reinterpret_cast<const float4*>(wp + A1_TT + 4 * lane);
                p1 = *reinterpret_cast<const float4*>(wp + A1_TT + A1_WT + 4 * (lane & 3));
            };
            stage_window(t0, true, p0, p1);
            bool more = next_is_interior();
            if (more) request_next();
            compute_tile(t0, true);
            ++tl;
            while (more) {
                t0 += A1_TT;
                wp += A1_TT;
                stage_window(t0, true, p0, p1);
                more = next_is_interior();
                if (more) request_next();
                compute_tile(t0, true);
                ++tl;
            }
        }
    }
}

template <int NTILE>
static hipError_t launch_act1d_n(const float* x, float* y, int B, int C, int T, const float* a_dev, const float* invb_dev,
                                 const float* fu, const float* fd, const int* lens, int len_mul, hipStream_t stream, int rev) {
    const int ntiles = (T + A1_TT - 1) / A1_TT;
    const int ngroups = (ntiles + NTILE - 1) / NTILE;
    dim3 grid((unsigned)((size_t)ngroups * (size_t)(B * C)));
    note_kernel("act1d_kernel", NTILE);
    note_work(grid.x, 0.0, 2 * 4.0 * B * (double)C * T / 1e6, "Activation1d C=%d T=%d B=%d", C, T, B);
    hipLaunchKernelGGL(act1d_kernel<NTILE>, grid, dim3(256), 0, stream, x, y, C, T, a_dev, invb_dev, fu, fd, lens,
                       len_mul, rev);
    return hipGetLastError();
}

hipError_t launch_act1d(const float* x, float* y, int B, int C, int T, const float* a_dev, const float* invb_dev,
                        const float* filt_up12, const float* filt_dn12, const int* lens, int len_mul,
                        hipStream_t stream, int rev) {
    // tiles per workgroup.  Measured on C3 (profiles/r2_uv_act1d.txt): 2 tiles (straight-line, both windows requested at
    // entry) 141 us per launch on average, rolled strips of 8 / 16 / 32 tiles 125.5 / 122.6 / 125.8 us: long-lived
    // workgroups stop paying their start-up 32 768 times per launch.  Strips need a grid that still fills the chip
    // (2 048 workgroup slots), so small launches (single utterances) keep short ones.
    const long long total = (long long)B * C * ((T + A1_TT - 1) / A1_TT);
    if (total >= 16 * 4096) return launch_act1d_n<16>(x, y, B, C, T, a_dev, invb_dev, filt_up12, filt_dn12, lens, len_mul, stream, rev);
    if (total >= 8 * 4096) return launch_act1d_n<8>(x, y, B, C, T, a_dev, invb_dev, filt_up12, filt_dn12, lens, len_mul, stream, rev);
    return launch_act1d_n<2>(x, y, B, C, T, a_dev, invb_dev, filt_up12, filt_dn12, lens, len_mul, stream, rev);
}

// ---------------------------------------------------------------------------------------------
// UpSample1d / LowPassFilter1d (DownSample1d) as stand-alone ops (resample.py:17-65, filter.py:64-99): the two
// halves of Activation1d for callers that use them on their own, any ratio / kernel size up to 64 taps.
// Depthwise FIR, one output per lane, rows of [B*C, T] coalesced along time; the taps travel as a kernel argument.
// ---------------------------------------------------------------------------------------------
struct FirTaps { float f[AMP_FIR_MAX_TAPS]; };

// y[n] = ratio * sum_m xp[m] * f[n + pad_left - m*ratio],  xp[m] = x[clamp(m - pad, 0, T-1)]   (resample.py:38-42)
__global__ __launch_bounds__(256) void fir_up_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                                     int K, int ratio, int pad, int pad_left, FirTaps taps) {
    const int Tout = ratio * T;
    const int tb = (Tout + 255) / 256;
    const size_t row = blockIdx.x / tb;
    const int n = (int)(blockIdx.x - row * tb) * 256 + threadIdx.x;
    if (n >= Tout) return;
    const float* xr = x + row * T;
    const int j = n + pad_left;
    float acc = 0.f;
    for (int m = j / ratio; m >= 0 && j - m * ratio < K; --m) {
        if (m >= T + 2 * pad) continue;
        int t = m - pad;
        t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
        acc = fmaf(xr[t], taps.f[j - m * ratio], acc);
    }
    y[row * Tout + n] = (float)ratio * acc;
}

// y[t] = sum_j f[j] * xp[t*stride + j],  xp = x padded by (pad_left, pad_right)            (filter.py:95-99)
//   mode 0 replicate, 1 zeros ("constant"), 2 reflect; without padding pad_left = pad_right = 0.
__global__ __launch_bounds__(256) void fir_filter_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                                         int Tout, int K, int stride, int pad_left, int mode,
                                                         FirTaps taps) {
    const int tb = (Tout + 255) / 256;
    const size_t row = blockIdx.x / tb;
    const int t = (int)(blockIdx.x - row * tb) * 256 + threadIdx.x;
    if (t >= Tout) return;
    const float* xr = x + row * T;
    float acc = 0.f;
    for (int j = 0; j < K; ++j) {
        int i = t * stride + j - pad_left;
        float v;
        if (i >= 0 && i < T) v = xr[i];
        else if (mode == 1) v = 0.f;
        else if (mode == 2) v = xr[i < 0 ? -i : 2 * (T - 1) - i];
        else v = xr[i < 0 ? 0 : T - 1];
        acc = fmaf(taps.f[j], v, acc);
    }
    y[row * Tout + t] = acc;
}

hipError_t launch_fir_up(const float* x, float* y, int rows, int T, const float* taps_host, int K, int ratio, int pad,
                         int pad_left, hipStream_t stream) {
    FirTaps taps;
    for (int i = 0; i < AMP_FIR_MAX_TAPS; ++i) taps.f[i] = i < K ? taps_host[i] : 0.f;
    const size_t blocks = (size_t)rows * (((size_t)ratio * T + 255) / 256);
    if (blocks > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fir_up_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, T, K, ratio, pad, pad_left, taps);
    return hipGetLastError();
}

hipError_t launch_fir_filter(const float* x, float* y, int rows, int T, int Tout, const float* taps_host, int K,
                             int stride, int pad_left, int mode, hipStream_t stream) {
    FirTaps taps;
    for (int i = 0; i < AMP_FIR_MAX_TAPS; ++i) taps.f[i] = i < K ? taps_host[i] : 0.f;
    const size_t blocks = (size_t)rows * (((size_t)Tout + 255) / 256);
    if (blocks > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fir_filter_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, T, Tout, K, stride,
                       pad_left, mode, taps);
    return hipGetLastError();
}

// Snake / SnakeBeta on their own (snake.py:51-61, 110-122): y = x + sin(a x)^2 / (b + 1e-9), a = alpha or
// exp(alpha), b = beta (SnakeBeta) or a.  One pass, rows of [B*C, T].
__global__ __launch_bounds__(256) void snake_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                    const float* __restrict__ alpha, const float* __restrict__ beta,
                                                    int logscale, int C, int T) {
    const int tb = (T + 1023) / 1024;
    const size_t row = blockIdx.x / tb;
    const int c = (int)(row % C);
    float a = alpha[c], b = beta ? beta[c] : a;
    if (logscale) { a = expf(a); b = beta ? expf(b) : a; }
    const float invb = 1.0f / (b + 0.000000001f);
    const int tbeg = (int)(blockIdx.x - row * tb) * 1024 + threadIdx.x;
    const int tend = min(T, (int)(blockIdx.x - row * tb) * 1024 + 1024);
    const float* xr = x + row * T;
    float* yr = y + row * T;
    for (int t = tbeg; t < tend; t += 256) {
        const float v = xr[t];
        yr[t] = fmaf(invb, snake_sin2(v * a), v);
    }
}

hipError_t launch_snake(const float* x, float* y, int B, int C, int T, const float* alpha, const float* beta,
                        int logscale, hipStream_t stream) {
    const size_t blocks = (size_t)B * C * ((T + 1023) / 1024);
    if (blocks > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(snake_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, alpha, beta, logscale, C, T);
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

// ---------------------------------------------------------------------------------------------
// fp32 waveform -> signed 16-bit PCM, the conversion the reference's save_audio performs on the host after the
// D2H copy (utils/io.py:68-76: torchaudio.save(..., encoding="PCM_S", bits_per_sample=16); torchaudio 2.0.2 /
// libsox 14.4.2, restated in oracle/pcm16.py).  Integer work, bit-exact:
//   d   = trunc(clamp(x * 2^31, INT32_MIN, INT32_MAX))                 (sox_sample_t; NaN -> INT32_MIN like x86)
//   pcm = d > INT32_MAX - 2^15 ? 32767 : (d + 2^15) >> 16              (SOX_SAMPLE_TO_SIGNED_16BIT: round half up)
// Samples at or past an item's length are written as 0.  8 samples per lane: 2 x 16-B loads, one 16-B store.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int pcm16_of(float x) {
    const float v = x * 2147483648.0f;                 // exact (power of two) or +-inf
    int d;
    if (v >= 2147483648.0f) d = 2147483647;
    else if (v > -2147483648.0f) d = (int)v;           // |v| < 2^31: truncation toward zero, exact
    else d = (-2147483647 - 1);                        // <= -2^31 and NaN
    return d > 2147483647 - 32768 ? 32767 : ((d + 32768) >> 16);
}

__global__ __launch_bounds__(256) void pcm16_kernel(const float* __restrict__ x, short* __restrict__ y, int L,
                                                    long long x_stride, long long y_stride,
                                                    const int* __restrict__ lens) {
    const int b = blockIdx.y;
    const int Lv = lens ? min(max(lens[b], 0), L) : L;
    const float* xr = x + (size_t)b * x_stride;
    short* yr = y + (size_t)b * y_stride;
    const bool vec = ((x_stride & 3) == 0) && ((y_stride & 7) == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    for (int t0 = (blockIdx.x * 256 + threadIdx.x) * 8; t0 < L; t0 += gridDim.x * 256 * 8) {
        if (vec && t0 + 8 <= Lv) {
            const float4 a = *reinterpret_cast<const float4*>(xr + t0);
            const float4 c = *reinterpret_cast<const float4*>(xr + t0 + 4);
            uint4 o;
            o.x = ((unsigned)pcm16_of(a.x) & 0xffffu) | ((unsigned)pcm16_of(a.y) << 16);
            o.y = ((unsigned)pcm16_of(a.z) & 0xffffu) | ((unsigned)pcm16_of(a.w) << 16);
            o.z = ((unsigned)pcm16_of(c.x) & 0xffffu) | ((unsigned)pcm16_of(c.y) << 16);
            o.w = ((unsigned)pcm16_of(c.z) & 0xffffu) | ((unsigned)pcm16_of(c.w) << 16);
            *reinterpret_cast<uint4*>(yr + t0) = o;
        } else {
            for (int e = 0; e < 8 && t0 + e < L; ++e) yr[t0 + e] = t0 + e < Lv ? (short)pcm16_of(xr[t0 + e]) : (short)0;
        }
    }
}

hipError_t launch_pcm16(const float* x, short* y, int B, int L, long long x_stride, long long y_stride,
                        const int* lens, hipStream_t stream) {
    unsigned gx = (unsigned)((L + 2047) / 2048);
    if (gx > 4096u) gx = 4096u;                         // grid-stride past 8 Mi samples per row
    hipLaunchKernelGGL(pcm16_kernel, dim3(gx, (unsigned)B), dim3(256), 0, stream, x, y, L, x_stride, y_stride, lens);
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

// MRF combination of resblocks that ran side by side and stored their results separately (generator.hip, concurrent mode, stages whose
// resblocks all end in a fused pair / whole-resblock kernel): y = ((y + p0) + p1 ...) / div.  Those kernels add the accumulated y to
// their finished, rounded result (`acc += y; acc /= div`), so this order and the true division give the bits of the sequential chain.
__global__ __launch_bounds__(256) void mrf_sum_kernel(MrfSumArgs a) {
    const size_t n4 = a.count >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(a.y)[i];
        float4 p[AMP_MRF_MAX_PARTS];
#pragma unroll
        for (int k = 0; k < AMP_MRF_MAX_PARTS; ++k)
            if (k < a.n) p[k] = reinterpret_cast<const float4*>(a.p[k])[i];      // uniform condition: all loads before the first add
#pragma unroll
        for (int k = 0; k < AMP_MRF_MAX_PARTS; ++k) {
            if (k < a.n) { v.x += p[k].x; v.y += p[k].y; v.z += p[k].z; v.w += p[k].w; }
        }
        v.x = v.x / a.div; v.y = v.y / a.div; v.z = v.z / a.div; v.w = v.w / a.div;
        reinterpret_cast<float4*>(a.y)[i] = v;
    }
    if (blockIdx.x == 0) {
        for (size_t i = (n4 << 2) + threadIdx.x; i < a.count; i += 256) {
            float v = a.y[i];
            for (int k = 0; k < a.n; ++k) v += a.p[k][i];
            a.y[i] = v / a.div;
        }
    }
}

hipError_t launch_mrf_sum(const MrfSumArgs& a, hipStream_t stream) {
    size_t blocks = ((a.count >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mrf_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
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
        // selects, not products with the mask: m may be UNSPECIFIED beyond the length (a ragged conv skips those tiles); for
        // t < len the bits are those of (x1 - m) * 1 and m + x1 * 1
        const float v = reverse ? x1[t] - mr[t] : mr[t] + x1[t];
        x1[t] = t < len ? v : 0.f;
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
