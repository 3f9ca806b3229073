// EXPERIMENTAL -- written after round 1's GPU budget was spent: compiled by build.py, launched only when
// AMP_FUSE_AMP=1 (default off).  NOT yet run on hardware.  Its seam indexing is checked against the oracle's
// Activation1d by a CPU emulation (tests/experiments/amp_seam_emulation.py); to verify on a GPU:
//     AMP_FUSE_AMP=1 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_full_size.py tests/test_gpu_inference_api.py
// (the BigVGAN generator tests then run every AMPBlock1 pair through this kernel).
//
// Fused AMPBlock1 pair of BigVGAN on the gfx950 f16 matrix cores (bigvgan.py:137-146, one iteration):
//
//     xt = a1(x)   (separate act1d launch, the input `xin` of this kernel)
//     y  = x + c2( a2( c1(xt) ) )              [+ the running MRF sum, / num_kernels]
//
// Same structure as pair_f16x3.hip (split-f16 operands, conv1 staged through LDS, conv2 from an LDS xt tile), with
// the anti-aliased activation a2 (Activation1d: up x2 -> Snake -> down x2, modules/anti_aliasing/act.py:31-36)
// applied AT THE SEAM instead of as its own pass over HBM:
//
//   phase 1   conv1 on N1 = 32*NI*WN columns = the NT outputs + conv2's halo (KT-1) + the activation's halo (2*5:
//             y[t] needs xt[t-5 .. t+5]).
//   seam      in four groups of 8 channels per 32-row block: fp32 conv1 output -> LDS scratch (aliasing the x staging
//             buffers, free after phase 1) -> every lane runs the activation along a stretch of one channel row, in
//             act1d_kernel's exact operation order (replicate clamps at the utterance's own ends included), zeroes
//             what lies outside the utterance (conv2's zero padding), scales x16, splits hi/lo and writes the f16
//             B-layout xt tile.
//   phase 2 / epilogue   as pair_f16x3.hip, the residual coming from `res` (the block's running x, not xin).
//
// Compiled once per tap count:  -DAMP_KT=<3|5|7|11>.
#include "amp_internal.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union FragA {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)

// sin(x)^2 as small_kernels.hip: snake_sin2 (kept identical; to be shared through a header once this kernel has run)
__device__ __forceinline__ float amp_snake_sin2(float x) {
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

constexpr int AH = 5;   // Activation1d needs xt[t-5 .. t+5] for y[t] (12-tap up and down filters at ratio 2)

template <int KT, int WM, int WN, int NI, int SX>
__global__ __launch_bounds__(256, 2) void amp_pair_f16x3_kernel(const AmpPairArgs a) {
    constexpr int N1 = 32 * NI * WN;          // conv1 output columns (tile-local 0 .. N1-1)
    constexpr int H2 = (KT - 1) / 2;
    constexpr int NV = N1 - 2 * AH;           // activated columns = xt-tile columns conv2 may read
    constexpr int NT = NV - 2 * H2;           // output columns per workgroup
    constexpr int XT = N1 + 12;               // xt row length (+ the read overrun of the unused tail columns)
    constexpr int NCH = 2 * WM;               // 16-channel chunks (C = 32 * WM)
    constexpr int XBUF = 4 * SX;              // uint4 per x staging buffer [plane][octet][SX]
    constexpr int XTCH = 4 * XT;              // uint4 per xt chunk       [plane][octet][XT]
    constexpr int NST = (4 * SX) / 256;       // staging items per thread
    constexpr int SROW = N1 + 4;              // fp32 scratch row stride (floats)
    constexpr int ROWS = 8 * WM;              // channel rows per seam group
    constexpr int LPR = 256 / ROWS;           // lanes per row in the activation
    constexpr int CPL = (NV + LPR - 1) / LPR; // activated columns per lane
    static_assert(SX % 64 == 0, "staging items must have a wave-uniform channel quad");
    static_assert(KT - 1 <= 12, "xt pad");
    static_assert((size_t)ROWS * SROW * sizeof(float) <= (size_t)2 * XBUF * sizeof(uint4), "seam scratch must fit the x staging buffers");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [2][XBUF] + [NCH][XTCH]
    uint4* const xt4 = smem4 + 2 * XBUF;
    float* const scr = reinterpret_cast<float*>(smem4);            // [ROWS][SROW], valid between phase 1 and phase 2

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    const int nbx = gridDim.x;  // XCD-contiguous tile runs, see conv_f16x3.hip
    const int bx = (nbx & 7) == 0 ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int q0 = tile * NT;                 // first output column
    const int C = 32 * WM;
    const int T = a.T;
    int Tv = T;                               // valid columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    const int dil = a.dil;
    const int h1 = H2 * dil;
    const int qa = q0 - H2 - AH;              // global column of conv1-output / scratch column 0

    const float* xb = a.xin + (size_t)item * C * T;
    const int tbase = qa - h1;                // global column of staged column 0

    // output positions of this lane (epilogue)
    const int colw = wn * (32 * NI) + l31;
    int qc[NI];
    bool okc[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int col = colw + 32 * t;
        const int q = q0 + col;
        okc[t] = (col < NT) && (q < T);
        qc[t] = q < T ? q : T - 1;
    }
    const float* xres = a.res + (size_t)item * C * T + (size_t)(32 * wm + 4 * hi) * T;

    // ---------------- phase 1: conv1 ----------------
    f32x16 acc[NI];
    {
        const float s1 = a.sc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = a.bias1[32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hi] * s1;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
        }
    }

    float xs[NST][4];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int ibase = wave * 64 + 256 * it;          // wave-uniform
            const int qd = ibase / SX;                       // channel quad 0..3
            const int col = ibase - qd * SX + lane;
            int t = tbase + col;
            t = t < 0 ? 0 : t;
            t = t > T - 1 ? T - 1 : t;
            const int ch0 = chunk * KC16 + 4 * qd;
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[it][e] = xb[(size_t)(ch0 + e) * T + t];
        }
    };
    auto stage_store = [&](int buf) {
        uint2* dst = reinterpret_cast<uint2*>(smem4 + buf * XBUF);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int ibase = wave * 64 + 256 * it;
            const int qd = ibase / SX;
            const int col = ibase - qd * SX + lane;
            const int t = tbase + col;
            const bool tok = (t >= 0) && (t < Tv);            // conv1 zero-pads a1(x) outside the utterance
            union { uint2 u; _Float16 h[4]; } fh, fl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = tok ? xs[it][e] * 16.f : 0.f;  // already activated: no leaky_relu on load
                split_f16(v, fh.h[e], fl.h[e]);
            }
            const int o2 = (((qd >> 1) * SX + col) << 1) + (qd & 1);
            dst[o2] = fh.u;
            dst[4 * SX + o2] = fl.u;
        }
    };

    const uint4* wa1 = static_cast<const uint4*>(a.wp1) + (size_t)wm * NCH * (KT * 128) + lane;
    const uint4* wa2 = static_cast<const uint4*>(a.wp2) + (size_t)wm * NCH * (KT * 128) + lane;
    FragA a_h[KT], a_l[KT];

    const int rd1 = hi * SX + colw;

    stage_load(0);
#pragma unroll
    for (int g = 0; g < KT; ++g) {
        a_h[g].u = wa1[g * 128];
        a_l[g].u = wa1[g * 128 + 64];
    }
    AMP_PIN_VMEM();
    stage_store(0);
    __syncthreads();

    for (int c = 0; c < NCH; ++c) {
        const bool more = (c + 1) < NCH;
        stage_load(more ? c + 1 : c);
        AMP_PIN_VMEM();
        // the reload during the LAST chunk re-reads that chunk: its values are dead (conv2's first chunk is fetched
        // after the seam, which needs the registers)
        const uint4* wan = wa1 + (size_t)(more ? c + 1 : c) * (KT * 128);
        const uint4* base = smem4 + (c & 1) * XBUF + rd1;
#pragma unroll
        for (int g = 0; g < KT; ++g) {
            const uint4* bg = base + g * dil;
            FragA bh[NI], bl[NI];
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                bh[t].u = bg[32 * t];
                bl[t].u = bg[2 * SX + 32 * t];
            }
#pragma unroll
            for (int t = 0; t < NI; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[g].h, bh[t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NI; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NI; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[g].h, bh[t].h, acc[t], 0, 0, 0);
            a_h[g].u = wan[g * 128];
            a_l[g].u = wan[g * 128 + 64];
            AMP_PIN_VMEM();
        }
        if (more) stage_store((c + 1) & 1);
        __syncthreads();
    }

    // ---------------- seam: y = a2(conv1 + bias) -> LDS xt tile, split-f16 B layout ----------------
    {
        const float i1 = a.isc1;
        // this lane's stretch of one channel row in the activation
        const int arow = tid / LPR;                              // 0 .. ROWS-1: 32-row block arow >> 3, row arow & 7
        const int oc0 = (tid - arow * LPR) * CPL;                // first activated column (xt-tile column)
        const int twoT = 2 * Tv;
        _Float16* const xth = reinterpret_cast<_Float16*>(xt4);
        for (int j = 0; j < 4; ++j) {
            // (a) conv1 output of channels 8j .. 8j+7 of every 32-row block, fp32, to the scratch rows
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const int col = colw + 32 * t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // acc register 4*j + i holds row 8*j + i + 4*hi of the block; the register index must be a
                    // compile-time constant, so select over j
                    float v = acc[t][i];
                    if (j == 1) v = acc[t][4 + i];
                    if (j == 2) v = acc[t][8 + i];
                    if (j == 3) v = acc[t][12 + i];
                    scr[(8 * wm + 4 * hi + i) * SROW + col] = v * i1;
                }
            }
            __syncthreads();
            // (b) Activation1d along the row, act1d_kernel's operation order (small_kernels.hip):
            //   u[n] = sum_k xt[clamp(((n + 15) >> 1) - k - 5)] * fu2[((n + 15) & 1) + 2k],  k = 0..5, from 0
            //   s[n] = u + invb * sin(a u)^2
            //   y[q] = (sum_m fd[2m] s[c(2q + 2m - 5)]) + (sum_m fd[2m+1] s[c(2q + 2m - 4)]),  c = clamp to [0, 2Tv)
            if (oc0 < NV) {
                const int ch = 32 * (arow >> 3) + 8 * j + (arow & 7);
                const float aa = a.act_a[ch], invb = a.act_invb[ch];
                const float* srow = scr + arow * SROW;
                const int qf = qa + AH + oc0;                    // global column of this lane's first output
                // one Snake value s[n] (n clamped here: DownSample1d's replicate padding of the Snake output)
                auto sval = [&](int n) {
                    n = n < 0 ? 0 : (n > twoT - 1 ? twoT - 1 : n);
                    const int np = n + 15;
                    const int mmax = np >> 1;
                    const bool odd = (np & 1) != 0;
                    float u = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        int xi = mmax - k - 5;                   // global xt column before clamping
                        xi = xi < 0 ? 0 : (xi > Tv - 1 ? Tv - 1 : xi);
                        int ci = xi - qa;                        // scratch column
                        ci = ci < 0 ? 0 : (ci > N1 - 1 ? N1 - 1 : ci);
                        u = fmaf(srow[ci], odd ? a.fu2[2 * k + 1] : a.fu2[2 * k], u);
                    }
                    return fmaf(invb, amp_snake_sin2(u * aa), u);
                };
                // sliding window of the 12 Snake values under the down filter (rolled loops: the fully unrolled
                // version spilled ~290 registers)
                float ring[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) ring[i] = 0.f;
                auto push = [&](float v) {
#pragma unroll
                    for (int i = 0; i < 11; ++i) ring[i] = ring[i + 1];
                    ring[11] = v;
                };
                int n = 2 * qf - 5;
#pragma nounroll
                for (int e = 0; e < 10; ++e) { push(sval(n)); ++n; }
                const int chunk = ch >> 4, octet = (ch >> 3) & 1, e8 = ch & 7;
#pragma nounroll
                for (int o = 0; o < CPL; ++o) {
                    push(sval(n)); ++n;
                    push(sval(n)); ++n;                          // ring = s[2q - 5 .. 2q + 6]
                    float ae = 0.f, ao = 0.f;
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        ae = fmaf(a.fd[2 * m], ring[2 * m], ae);
                        ao = fmaf(a.fd[2 * m + 1], ring[2 * m + 1], ao);
                    }
                    const int oc = oc0 + o;
                    const int q = qf + o;
                    const float v = (q >= 0 && q < Tv) ? (ae + ao) * 16.f : 0.f;   // conv2 zero-pads outside the utterance
                    _Float16 vh, vl;
                    split_f16(v, vh, vl);
                    if (oc < NV) {
                        const int o4 = chunk * XTCH + octet * XT + oc;
                        xth[o4 * 8 + e8] = vh;
                        xth[(o4 + 2 * XT) * 8 + e8] = vl;
                    }
                }
            }
            __syncthreads();   // the scratch rows are rewritten by the next group
        }
    }
    {
        const float s2 = a.sc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = a.bias2[32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hi] * s2;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
        }
    }
#pragma unroll
    for (int g = 0; g < KT; ++g) {
        a_h[g].u = wa2[g * 128];
        a_l[g].u = wa2[g * 128 + 64];
    }
    AMP_PIN_VMEM();
    // (the last group's barrier already ordered every xt-tile write before the reads below)

    // ---------------- phase 2: conv2 over the xt tile ----------------
    {
        const int rd2 = hi * XT + colw;
        for (int c = 0; c < NCH; ++c) {
            const uint4* wan = wa2 + (size_t)(c + 1) * (KT * 128);   // last: next mb block / pad
            const uint4* base = xt4 + c * XTCH + rd2;
#pragma unroll
            for (int g = 0; g < KT; ++g) {
                const uint4* bg = base + g;
                FragA bh[NI], bl[NI];
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    bh[t].u = bg[32 * t];
                    bl[t].u = bg[2 * XT + 32 * t];
                }
#pragma unroll
                for (int t = 0; t < NI; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[g].h, bh[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[g].h, bh[t].h, acc[t], 0, 0, 0);
                a_h[g].u = wan[g * 128];
                a_l[g].u = wan[g * 128 + 64];
                AMP_PIN_VMEM();
            }
        }
    }

    // ---------------- epilogue: + residual, MRF accumulate, store ----------------
    {
        const float i2 = a.isc2;
        const int mode = a.mode;
        float* yr = a.y + (size_t)item * C * T + (size_t)(32 * wm + 4 * hi) * T;
        f32x16 rv[NI];
#pragma unroll
        for (int t = 0; t < NI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[t][r] = xres[(size_t)((r & 3) + 8 * (r >> 2)) * T + qc[t]];
#pragma unroll
        for (int t = 0; t < NI; ++t) acc[t] = acc[t] * i2 + rv[t];
        if (mode != 0) {   // wave-uniform
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[t][r] = yr[(size_t)((r & 3) + 8 * (r >> 2)) * T + qc[t]];
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] += rv[t];
            if (mode == 2) {
#pragma unroll
                for (int t = 0; t < NI; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] / a.div;
            }
        }
#pragma unroll
        for (int t = 0; t < NI; ++t)
            if (okc[t]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yr[(size_t)((r & 3) + 8 * (r >> 2)) * T + qc[t]] = acc[t][r];
            }
    }
}

template <int KT, int WM, int WN, int NI, int SX>
static hipError_t launch_amp_pair_one(const AmpPairArgs& a, hipStream_t stream) {
    constexpr int N1 = 32 * NI * WN;
    constexpr int XT = N1 + 12;
    const size_t lds = ((size_t)2 * 4 * SX + (size_t)2 * WM * 4 * XT) * sizeof(uint4);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&amp_pair_f16x3_kernel<KT, WM, WN, NI, SX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((unsigned)(a.B * a.tiles_per_item));
    hipLaunchKernelGGL((amp_pair_f16x3_kernel<KT, WM, WN, NI, SX>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// Output columns per workgroup for C channels, or 0 when (C, KT, dilation) is not covered.
int AMP_CAT(amp_pair_tile_kt, AMP_KT)(int C, int dil) {
    constexpr int KT = AMP_KT;
    const int span = (KT - 1) * dil;   // 2 * h1
    if (C == 128) return (96 + span <= 192) ? 96 - 2 * AH - (KT - 1) : 0;
    if (C == 64) return (128 + span <= 192) ? 128 - 2 * AH - (KT - 1) : 0;
    if (C == 32) return (256 + span <= 320) ? 256 - 2 * AH - (KT - 1) : 0;
    return 0;
}

hipError_t AMP_CAT(launch_amp_pair_kt, AMP_KT)(const AmpPairArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    if (a.C == 128) return launch_amp_pair_one<KT, 4, 1, 3, 192>(a, stream);
    if (a.C == 64) return launch_amp_pair_one<KT, 2, 2, 2, 192>(a, stream);
    if (a.C == 32) return launch_amp_pair_one<KT, 1, 4, 2, 320>(a, stream);
    return hipErrorInvalidValue;
}

}  // namespace amp
