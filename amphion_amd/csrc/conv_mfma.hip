// Implicit-GEMM 1-D convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the whole generator, the reference's F.conv1d / nn.ConvTranspose1d calls
// (hifigan.py:93-100,204,207,216; bigvgan.py:137-146,314-329) together with the element-wise ops
// around them (leaky_relu before/after, bias, residual add, MRF accumulate and 1/num_kernels).
//
//   GEMM view (per batch item):  Y'[M, Tq] = W'[M, K] * Xcol[K, Tq],   K = Cin * taps
//   - A operand (weights) is pre-packed on the host in MFMA fragment order and streamed from L2
//     straight into VGPRs (one float4 = the four k-steps of one tap of one 8-channel chunk).
//   - B operand (activations) is staged once per 8-channel chunk into LDS as [8][NT + HALO] rows
//     (time fastest, coalesced global reads, activation + zero padding applied while staging); every
//     tap is the same rows read at a shifted column, 32 consecutive lanes -> 32 consecutive banks.
//   - fp32 MFMA is bit-for-bit an fmaf chain, so parity with the fp32 reference is at rounding level.
//
// This file is compiled once per tap count:  -DAMP_KT=<1|2|3|5|7|11>.
#include "amp_internal.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KT, int WM, int WN, int NI, int HALO>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int NT = 32 * NI * WN;            // output columns per workgroup
    constexpr int S = NT + HALO;                // LDS row stride (floats)
    constexpr int NST = (KC * S + 255) / 256;   // staging elements per thread per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][KC][S]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    // Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own
    // L2: hand every XCD a contiguous run of tiles, so the halo columns two neighbouring tiles share
    // are fetched into ONE L2 instead of two.
    const int nbx = gridDim.x;
    // (ragged batches keep the dispatch order: with utterances of different lengths a contiguous run per XCD would hand
    // one XCD the long utterances and another only tiles that exit at once -- measured 43.8 vs 48.7 ms padded, visit AD)
    const int bx = ((nbx & 7) == 0 && !a.lens) ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int q0 = tile * NT;
    // ragged batch: a tile that lies entirely beyond this utterance's valid length produces only samples the contract
    // leaves unspecified (nothing downstream reads them: every layer takes its input as zero / replicated beyond the valid
    // length) -- skip it.  A batch of 60..400-frame utterances is 40 % such tiles.
    if (a.lens) {
        const long long lv = (long long)a.lens[item] * a.len_mul;                       // valid INPUT columns
        const long long first_out = (long long)q0 * a.up - a.up_pad;                    // first output sample of the tile
        if ((a.up > 1 || a.Tout == a.Tin) && first_out >= lv * a.up) return;            // block-uniform, before any barrier
    }
    const int mb = blockIdx.y * WM + wm;        // 32-row block of W'

    // Accumulators start from bias (+ residual) (+ the running MRF sum): the epilogue is then
    // store-only, and these loads overlap the first staging instead of serialising behind stores
    // (res / y may alias, so the compiler cannot batch epilogue loads across stores).
    const int up = a.up;
    const int qw = q0 + wn * (32 * NI) + l31;
    // wave-uniform: whole 32 x (32*NI) wave tile in range and no polyphase scatter -> straight-line
    // loads/stores with scalar row bases (no per-element predicates)
    const bool fast = (up == 1) && (mb * 32 + 32 <= a.M) && (q0 + wn * (32 * NI) + 32 * NI <= a.Tq);
    const size_t wave_base = ((size_t)item * a.Cout + (size_t)mb * 32) * a.Tout;  // uniform
    const int lane_off = (4 * hi) * a.Tout + qw;                                  // per lane
    f32x16 acc[NI];
    if (fast) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowc = (r & 3) + 8 * (r >> 2);
            const float bv = a.bias ? a.bias[mb * 32 + rowc + 4 * hi] : 0.f;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
        }
        if (a.res) {
            const float* rp = a.res + wave_base;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* rr_ = rp + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t][r] += rr_[lane_off + 32 * t];
            }
        }
        if (a.mode != 0) {
            const float* yp = a.y + wave_base;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* yr_ = yp + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t][r] += yr_[lane_off + 32 * t];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int m = mb * 32 + row;
            const bool mok = m < a.M;
            const int o = (up == 1) ? m : m / up;
            const int rr = m - o * up;
            const float bv = (mok && a.bias) ? a.bias[o] : 0.f;
            const size_t rowoff = ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const int q = qw + 32 * t;
                const int n = q * up + rr - a.up_pad;
                float v = bv;
                if (mok && q < a.Tq && n >= 0 && n < a.Tout) {
                    if (a.res) v += a.res[rowoff + n];
                    if (a.mode != 0) v += a.y[rowoff + n];
                }
                acc[t][r] = v;
            }
        }
    }

    const float* xb = a.x + (size_t)item * (size_t)a.xbs;
    const int tbase = q0 - a.halo_left;
    int Tv = a.Tin;  // valid input columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    const float slope_in = a.slope_in;

    // stage_load issues UNCONDITIONAL loads from clamped addresses (a predicated load makes hipcc branch
    // around it and lose its vmcnt bookkeeping: the loads of chunk c+1 were then drained before the MFMAs of
    // chunk c instead of flying under them); the zero padding is applied as a select in stage_store.
    float xs[NST];
    auto stage_index = [&](int it, int chunk, int& ch, int& tr, bool& ok) {
        const int idx = tid + 256 * it;
        const int row = idx / S;
        const int col = idx - row * S;
        ch = chunk * KC + row;
        tr = tbase + col;
        if (a.pad_reflect) {
            tr = tr < 0 ? -tr : tr;
            tr = tr > Tv - 1 ? 2 * (Tv - 1) - tr : tr;
        }
        ok = (idx < KC * S) && (col < a.wd) && (ch < a.Cin) && (tr >= 0) && (tr < Tv);
    };
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            int ch, tr;
            bool ok;
            stage_index(it, chunk, ch, tr, ok);
            ch = ch > a.Cin - 1 ? a.Cin - 1 : ch;
            tr = tr < 0 ? 0 : (tr > a.Tin - 1 ? a.Tin - 1 : tr);
            xs[it] = xb[(size_t)ch * a.Tin + tr];
        }
    };
    auto stage_store = [&](int chunk, int buf) {
        float* dst = smem + buf * (KC * S);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = tid + 256 * it;
            int ch, tr;
            bool ok;
            stage_index(it, chunk, ch, tr, ok);
            float v = ok ? xs[it] : 0.f;
            v = v > 0.f ? v : v * slope_in;
            if (idx < KC * S) dst[idx] = v;
        }
    };

    const float4* wa = static_cast<const float4*>(a.wp) + (size_t)mb * a.nchunks * (KT * 64) + lane;
    // One register set for the A fragments: a tap's float4 is re-loaded for the NEXT chunk right
    // after its last use, so the L2 latency hides under the remaining KT-1 taps of MFMAs.
    float4 a_cur[KT];

    const int rd0 = hi * S + wn * (32 * NI) + l31 + a.halo_left + a.off0;
    const int dstep = a.dstep;

    stage_load(0);
#pragma unroll
    for (int g = 0; g < KT; ++g) a_cur[g] = wa[(size_t)g * 64];
    stage_store(0, 0);
    __syncthreads();

    const int nchunks = a.nchunks;
    for (int c = 0; c < nchunks; ++c) {
        const bool more = (c + 1) < nchunks;
        stage_load(more ? c + 1 : c);   // unconditional (a branch around loads would blur the vmcnt counts)
        const float4* wan = wa + (size_t)(c + 1) * (KT * 64);
        const float* base = smem + (c & 1) * (KC * S) + rd0;
#pragma unroll
        for (int g = 0; g < KT; ++g) {
            const float* bg = base + g * dstep;
            const float av[4] = {a_cur[g].x, a_cur[g].y, a_cur[g].z, a_cur[g].w};
            a_cur[g] = wan[(size_t)g * 64];   // past the last chunk: next mb block / allocation pad
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    const float bv = bg[p * 2 * S + 32 * t];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bv, acc[t], 0, 0, 0);
                }
            }
        }
        if (more) stage_store(c + 1, (c + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: MRF mean, activation-on-store, (polyphase) scatter ----
    const float slope_out = a.slope_out;
    if (fast) {
        const float scale = a.mode == 2 ? 1.f / a.div : 1.f;
        float* yp = a.y + wave_base;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* yr_ = yp + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                float v = acc[t][r];
                if (a.mode == 2) v = v / a.div;
                (void)scale;
                v = v > 0.f ? v : v * slope_out;
                if (a.tanh_out) v = tanhf(v);
                yr_[lane_off + 32 * t] = v;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int m = mb * 32 + row;
            if (m < a.M) {
                const int o = (up == 1) ? m : m / up;
                const int rr = m - o * up;
                const size_t rowoff = ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    const int q = qw + 32 * t;
                    const int n = q * up + rr - a.up_pad;
                    if (q < a.Tq && n >= 0 && n < a.Tout) {
                        float v = acc[t][r];
                        if (a.mode == 2) v = v / a.div;
                        v = v > 0.f ? v : v * slope_out;
                        if (a.tanh_out) v = tanhf(v);
                        a.y[rowoff + n] = v;
                    }
                }
            }
        }
    }
}

template <int KT, int WM, int WN, int NI, int HALO>
static hipError_t launch_one(const ConvArgs& a, hipStream_t stream) {
    constexpr int NT = 32 * NI * WN;
    constexpr int S = NT + HALO;
    const size_t lds = (size_t)2 * KC * S * sizeof(float);
    dim3 grid((unsigned)(a.B * a.tiles_per_item), (unsigned)((a.M + 32 * WM - 1) / (32 * WM)));
    note_kernel("conv_mfma_kernel", KT, WM, WN, NI, HALO);
    note_conv_work(a, KT, grid);
    hipLaunchKernelGGL((conv_mfma_kernel<KT, WM, WN, NI, HALO>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

hipError_t AMP_CAT(launch_conv_kt, AMP_KT)(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    if (p.HALO == 64) {
        if (p.WM == 4) return launch_one<KT, 4, 1, 8, 64>(a, stream);
        if (p.WM == 2) return launch_one<KT, 2, 2, 8, 64>(a, stream);
        return launch_one<KT, 1, 4, 4, 64>(a, stream);
    } else {
        if (p.WM == 4) return launch_one<KT, 4, 1, 8, 128>(a, stream);
        if (p.WM == 2) return launch_one<KT, 2, 2, 8, 128>(a, stream);
        return launch_one<KT, 1, 4, 4, 128>(a, stream);
    }
}

}  // namespace amp
