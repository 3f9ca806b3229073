// Implicit-GEMM 1-D convolution on the gfx950 f16 matrix cores with SPLIT operands
// (3 x v_mfma_f32_32x32x16_f16 per product term, f32 accumulate) -- "f16x3".
//
// Same contract as conv_mfma.hip (the exact-f32 kernel): replaces the reference's F.conv1d /
// nn.ConvTranspose1d calls (hifigan.py:93-100,204,207,216; bigvgan.py:137-146,314-329) with the
// leaky_relu / bias / residual / MRF-accumulate ops around them fused in.  What differs is the
// arithmetic of the contraction: f32 MFMA runs at the f32 vector rate (157 TFLOP/s), f16 MFMA at 16x
// that, so each f32 operand is split as  v = hi + lo  with hi = f16(v), lo = f16(v - hi)  (22 mantissa
// bits together) and the product is formed as  Whi*Xhi + Whi*Xlo + Wlo*Xhi  in the f32 accumulator
// (the dropped Wlo*Xlo term is 2^-22 relative).  End to end through HiFi-GAN V1 this is as close to an
// fp64 run of the reference as the reference's own fp32 run is (tests/experiments/numerics_f16x3.py, tests).
//
//   scaling (all exact powers of two, undone in the epilogue): activations x16 while staging, so that
//   `lo` of anything >= 2^-7 is a normal f16 and |x| up to 4094 is representable; weights by the
//   per-conv 2^s that puts max|w| in [2^12, 2^13] (host, conv_build()).
//
//   GEMM view (per batch item):  Y'[M, Tq] = W'[M, K] * Xcol[K, Tq],   K = Cin * taps
//   - A operand: packed on the host in MFMA fragment order as hi/lo f16 planes and streamed from L2
//     into VGPRs: one 16-B load per lane = the 8 channels that lane group owns, for one tap.
//   - B operand: a 16-channel chunk is staged global -> VGPR -> LDS once per chunk as
//     [plane hi|lo][channel octet h][column][8 x f16]: a lane's B fragment for ANY tap is ONE
//     ds_read_b128 at column + tap*dilation (16-B aligned, conflict-free: 16 consecutive lanes cover
//     all 64 banks).  leaky_relu-on-load, zero padding and the f32 -> hi/lo split happen while staging.
//
// This file is compiled once per tap count:  -DAMP_KT=<1|2|3|5|7|11>.
#include "amp_internal.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

union Frag {
    uint4 u;
    f16x8 h;
};

// VMEM and MFMA may not cross (VALU, SALU, DS may): pins where the global loads are issued relative to
// the MFMA blocks and their issue ORDER, on which the
// counted s_waitcnt vmcnt(N) the compiler derives depends (loads return in order).
#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)

template <int KT, int WM, int WN, int NI, int HALO>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(const ConvArgs a) {
    constexpr int NT = 32 * NI * WN;           // output columns per workgroup
    constexpr int S = NT + HALO;               // staged columns
    constexpr int NST = (4 * S) / 256;         // staging items (column x channel quad) per thread
    constexpr int BUF = 4 * S;                 // uint4 per LDS buffer: [plane hi|lo][octet h][S]
    static_assert(S % 64 == 0, "the channel quad of a staging item must be wave-uniform");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [2][BUF]

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
    int bx = ((nbx & 7) == 0 && !a.lens) ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (a.rev) bx = nbx - 1 - bx;   // descending tile order: start where the previous launch stopped writing (ConvArgs::rev)
    int rg = blockIdx.y;         // group of 32 * WM rows
    if (a.row_groups > 0) { rg = bx % a.row_groups; bx /= a.row_groups; }   // row group fastest, see ConvArgs
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
    const int mb = rg * WM + wm;               // 32-row block of W'

    // accumulators start from (bias + residual + running MRF sum) * acc_scale, see conv_mfma.hip
    const int up = a.up;
    const int qw = q0 + wn * (32 * NI) + l31;
    const bool fast = (up == 1) && (mb * 32 + 32 <= a.M) && (q0 + wn * (32 * NI) + 32 * NI <= a.Tq);
    const size_t wave_base = ((size_t)item * a.Cout + (size_t)mb * 32) * a.Tout;  // uniform
    const int lane_off = (4 * hi) * a.Tout + qw;                                  // per lane
    const float asc = a.acc_scale;
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
    } else if (!a.res && a.mode == 0) {   // transposed convs / ragged tiles without residual: bias only
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int o = (up == 1) ? m : m / up;
            const float bv = (m < a.M && a.bias) ? a.bias[o] : 0.f;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
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
#pragma unroll
    for (int t = 0; t < NI; ++t) acc[t] *= asc;

    const float* xb = a.x + (size_t)item * (size_t)a.xbs;
    const int tbase = q0 - a.halo_left;
    int Tv = a.Tin;                                      // valid input columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    const float kpos = 16.f, kneg = 16.f * a.slope_in;  // x16 and leaky_relu-on-load in one multiply

    // Staging item = (column, channel quad): 4 global dword loads (lanes = consecutive columns, so
    // every load instruction is one coalesced 256-B row segment) -> two 8-B LDS writes (hi, lo).
    // Split in two so that the loads of chunk c+1 fly under the MFMAs of chunk c: stage_load issues
    // UNCONDITIONAL loads from clamped addresses (a predicated load makes hipcc branch around it and
    // lose its vmcnt bookkeeping), stage_store applies the zero padding as a select, then leaky_relu,
    // the x16 scaling and the hi/lo split.  4*S is a multiple of 256 and S of 64: every thread has
    // exactly NST items and an item's quad is wave-uniform.
    float range_max = 0.f;       // largest |staged operand| (x16 applied): beyond 65504 it left the f16 range (a.range_flag)
    float xs[NST][4];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int ibase = wave * 64 + 256 * it;          // wave-uniform
            const int qd = ibase / S;                        // channel quad 0..3
            const int col = ibase - qd * S + lane;
            int t = tbase + col;
            if (a.pad_reflect) {                             // wave-uniform: mirror without repeating the edge
                t = t < 0 ? -t : t;
                t = t > Tv - 1 ? 2 * (Tv - 1) - t : t;
            }
            t = t < 0 ? 0 : t;
            t = t > a.Tin - 1 ? a.Tin - 1 : t;
            const int ch0 = chunk * KC16 + 4 * qd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int ch = ch0 + e;
                ch = ch > a.Cin - 1 ? a.Cin - 1 : ch;        // scalar clamp
                xs[it][e] = xb[(size_t)ch * a.Tin + t];
            }
        }
    };
    auto stage_store = [&](int chunk, int buf) {
        uint2* dst = reinterpret_cast<uint2*>(smem4 + buf * BUF);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int ibase = wave * 64 + 256 * it;
            const int qd = ibase / S;
            const int col = ibase - qd * S + lane;
            const int t = tbase + col;
            const bool tok = (col < a.wd) && (a.pad_reflect || ((t >= 0) && (t < Tv)));
            const int ch0 = chunk * KC16 + 4 * qd;
            struct { uint2 u; } fh, fl;
            stage4_f16((tok && (ch0 + 0) < a.Cin) ? xs[it][0] : 0.f, (tok && (ch0 + 1) < a.Cin) ? xs[it][1] : 0.f, (tok && (ch0 + 2) < a.Cin) ? xs[it][2] : 0.f, (tok && (ch0 + 3) < a.Cin) ? xs[it][3] : 0.f,
                       kpos, kneg, range_max, fh.u, fl.u);
            // uint2 index inside a plane: ((octet * S + col) * 2 + half)
            const int o2 = (((qd >> 1) * S + col) << 1) + (qd & 1);
            dst[o2] = fh.u;
            dst[4 * S + o2] = fl.u;
        }
    };

    // A fragments: [mb][chunk][tap][plane][lane] x uint4.  ONE register set for a whole chunk: tap g's
    // pair is re-loaded for the NEXT chunk right after its last use, so during the MFMAs of a chunk every
    // wait is for a load issued one chunk earlier -- older than the staging loads in flight (loads
    // return in order: a wait for a younger load would drain them).  The reload after the last chunk
    // reads the next mb block / the allocation pad (conv_build()).
    const uint4* wa = static_cast<const uint4*>(a.wp) + (size_t)mb * a.nchunks * (KT * 128) + lane;
    Frag a_h[KT], a_l[KT];

    const int rd0 = hi * S + wn * (32 * NI) + l31 + a.halo_left + a.off0;
    const int dstep = a.dstep;

    stage_load(0);
#pragma unroll
    for (int g = 0; g < KT; ++g) {
        a_h[g].u = wa[g * 128];
        a_l[g].u = wa[g * 128 + 64];
    }
    AMP_PIN_VMEM();
    stage_store(0, 0);
    __syncthreads();

    // One chunk of the contraction; `more` (a literal at both call sites): the next chunk's loads fly under this chunk's
    // MFMAs.  The last chunk is peeled so that it issues none (round 2 re-loaded the last chunk there to keep the loop
    // uniform: for C = 32, two chunks, a third of the kernel's loads were thrown away).
    const int nchunks = a.nchunks;
    auto chunk = [&](const int c, const bool more) __attribute__((always_inline)) {
        if (more) {
            stage_load(c + 1);
            AMP_PIN_VMEM();
        }
        wa += KT * 128;
        const uint4* base = smem4 + (c & 1) * BUF + rd0;
#pragma unroll
        for (int g = 0; g < KT; ++g) {
            const uint4* bg = base + g * dstep;
            Frag bh[NI], bl[NI];
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                bh[t].u = bg[32 * t];
                bl[t].u = bg[2 * S + 32 * t];
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
            if (more) {
                a_h[g].u = wa[g * 128];
                a_l[g].u = wa[g * 128 + 64];
                AMP_PIN_VMEM();
            }
        }
        if (more) stage_store(c + 1, (c + 1) & 1);
        __syncthreads();
    };
    for (int c = 0; c + 1 < nchunks; ++c) chunk(c, true);
    chunk(nchunks - 1, false);

    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);

    // ---- epilogue: undo the operand scaling, MRF mean, activation-on-store, (polyphase) scatter ----
    const float slope_out = a.slope_out;
    const bool exact_div = a.mode == 2;  // x = xs / num_kernels is a true division (hifigan.py:214)
    if (fast) {
        float* yp = a.y + wave_base;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* yr_ = yp + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                float v = acc[t][r] * a.inv_scale;
                if (exact_div) v = v / a.div;
                v = v > 0.f ? v : v * slope_out;
                if (a.tanh_out) v = tanhf(v);
                yr_[lane_off + 32 * t] = v;
            }
        }
    } else if (up > 1 && (up & 3) == 0 && (a.up_pad & 3) == 0 && (mb * 32 + 32 <= a.M) && a.mode == 0) {
        // Polyphase scatter, stride % 4 == 0 (the 8x stages): a lane's 4 rows (r & 3) of one register quad
        // are 4 CONSECUTIVE phases of one output channel = 4 consecutive samples n0..n0+3, 16-B aligned
        // (n0 = q*up + phase0 - pad with up, phase0, pad all multiples of 4): one float4 store, and the two
        // lane halves (hi) interleave to fully coalesced rows.  (Scalar stores at stride `up` touched one
        // 32-B sector per lane.)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m0 = mb * 32 + 8 * j + 4 * hi;          // GEMM row of (r & 3) == 0
            const int o = m0 / up;
            const int ph = m0 - o * up;                       // phase of that row, multiple of 4
            float* yrow = a.y + ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const int q = qw + 32 * t;
                const int n0 = q * up + ph - a.up_pad;
                if (q < a.Tq && n0 >= 0 && n0 + 3 < a.Tout) {
                    float4 v;
                    v.x = acc[t][4 * j + 0] * a.inv_scale;
                    v.y = acc[t][4 * j + 1] * a.inv_scale;
                    v.z = acc[t][4 * j + 2] * a.inv_scale;
                    v.w = acc[t][4 * j + 3] * a.inv_scale;
                    v.x = v.x > 0.f ? v.x : v.x * slope_out;
                    v.y = v.y > 0.f ? v.y : v.y * slope_out;
                    v.z = v.z > 0.f ? v.z : v.z * slope_out;
                    v.w = v.w > 0.f ? v.w : v.w * slope_out;
                    *reinterpret_cast<float4*>(yrow + n0) = v;
                }
            }
        }
    } else if (up == 2 && (mb * 32 + 32 <= a.M) && a.mode == 0) {
        // stride 2: rows (r & 3) = {0,1} and {2,3} are the two phases of two consecutive output channels:
        // two consecutive samples each -> 8-B stores (4-B aligned: pad is odd), lanes contiguous
        struct __attribute__((packed, aligned(4))) F2 { float a, b; };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int m0 = mb * 32 + 8 * j + 4 * hi + 2 * h2;   // even row: phase 0
                const int o = m0 >> 1;
                float* yrow = a.y + ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    const int q = qw + 32 * t;
                    const int n0 = 2 * q - a.up_pad;
                    float v0 = acc[t][4 * j + 2 * h2] * a.inv_scale;
                    float v1 = acc[t][4 * j + 2 * h2 + 1] * a.inv_scale;
                    v0 = v0 > 0.f ? v0 : v0 * slope_out;
                    v1 = v1 > 0.f ? v1 : v1 * slope_out;
                    if (q < a.Tq) {
                        if (n0 >= 0 && n0 + 1 < a.Tout) {
                            F2 v{v0, v1};
                            *reinterpret_cast<F2*>(yrow + n0) = v;
                        } else {
                            if (n0 >= 0 && n0 < a.Tout) yrow[n0] = v0;
                            if (n0 + 1 >= 0 && n0 + 1 < a.Tout) yrow[n0 + 1] = v1;
                        }
                    }
                }
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
                        float v = acc[t][r] * a.inv_scale;
                        if (exact_div) v = v / a.div;
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
static hipError_t launch_one_h(const ConvArgs& a, hipStream_t stream) {
    constexpr int NT = 32 * NI * WN;
    constexpr int S = NT + HALO;
    const size_t lds = (size_t)2 * 4 * S * sizeof(uint4);
    if (hipError_t e = ensure_dynamic_lds<&conv_f16x3_kernel<KT, WM, WN, NI, HALO>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.tiles_per_item), (unsigned)((a.M + 32 * WM - 1) / (32 * WM)));
    if (a.row_groups > 0) grid = dim3((unsigned)(a.B * a.tiles_per_item * a.row_groups), 1u);   // conv_run: row_groups == grid.y
    note_kernel("conv_f16x3_kernel", KT, WM, WN, NI, HALO);
    note_conv_work(a, KT, grid);
    hipLaunchKernelGGL((conv_f16x3_kernel<KT, WM, WN, NI, HALO>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

hipError_t AMP_CAT(launch_conv_h_kt, AMP_KT)(const ConvPlan& p, const ConvArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    // 4 accumulator tiles (64 VGPRs) per wave: the register budget goes to the per-chunk A-fragment set
    // (8 * KT VGPRs) instead, see the kernel.  NI = 2 halves the tile width for grids that would leave most of
    // the 256 CUs idle (single utterances, frame-rate convs): twice the workgroups, half the work in each.
#define AMP_LAUNCH_NI(NI_)                                                          \
    if (p.HALO == 64) {                                                             \
        if (p.WM == 4) return launch_one_h<KT, 4, 1, NI_, 64>(a, stream);           \
        if (p.WM == 2) return launch_one_h<KT, 2, 2, NI_, 64>(a, stream);           \
        return launch_one_h<KT, 1, 4, NI_, 64>(a, stream);                          \
    } else {                                                                        \
        if (p.WM == 4) return launch_one_h<KT, 4, 1, NI_, 128>(a, stream);          \
        if (p.WM == 2) return launch_one_h<KT, 2, 2, NI_, 128>(a, stream);          \
        return launch_one_h<KT, 1, 4, NI_, 128>(a, stream);                         \
    }
    if (p.NI == 2) { AMP_LAUNCH_NI(2) }
    AMP_LAUNCH_NI(4)
#undef AMP_LAUNCH_NI
}

}  // namespace amp
