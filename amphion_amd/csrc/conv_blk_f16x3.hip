// Row-blocked variant of the f16x3 implicit-GEMM conv (conv_f16x3.hip) for convs with 256+ GEMM rows: the transposed convs
// (hifigan.py:204-207: polyphase rows, 2 taps per 16-channel chunk) and the convs of the C = 256 stage (hifigan.py:93-100 unfused;
// k = 3 with a whole-chunk A-fragment set, k = 7 / 11 with a ring of four taps, see RING below).
//
// In conv_f16x3.hip a wave owns 32 GEMM rows x 128 columns: per tap it fetches 2 A fragments (hi, lo) from L2 and 8 B
// fragments from LDS for 12 MFMAs, and a 16-channel chunk of x is staged (global -> split -> LDS) once per 128-row
// workgroup -- with 11 taps that staging is shared by 132 MFMAs per wave, with the 2 taps of a ConvTranspose1d by 24,
// and the 2 048 polyphase rows of the first up-sampling layer re-stage every x tile sixteen times.  Here a wave owns
// MI = 2 row blocks (64 rows; a workgroup 256 rows): the B fragments of a tap feed both (operand fragments per MFMA
// 0.83 -> 0.5 / 0.56), x is staged half as often, and CM 16-channel chunks share one staging round / barrier (CM * KT
// taps per round instead of KT), so that a round's MFMAs cover the latency of the next round's loads.
//
// Per output element the order of operations (accumulator start, chunks, taps, the three MFMAs of a term, epilogue) is
// that of conv_f16x3.hip: the results are bit-identical (tests/test_gpu_f16x3_kernels.py).
//
// Compiled once per tap count:  -DAMP_KT=<2|3|7|11>.
#include "amp_internal.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union FragB {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)

// 4 / WN waves along M, WN along the columns, MI = 2 row blocks per wave: 256 / WN rows x WN * 32 * NI columns per workgroup.
// WN = 1: 256-row groups (the C = 256 stage, the transposed convs' polyphase rows); round 4: WN = 2 for convs with 128 rows -- BigVGAN's
// AMPBlock convs at C = 128 cannot be paired (an activation sits between them) and ran on conv_f16x3.hip's 32-row wave tiles.  Measured
// inside the C3 forward (profiles/r4_t_narrow_blocked_conv.txt): k = 7 / 11 resblocks 2.98 / 3.77 -> 2.91 / 3.72 ms, k = 3 1.86 -> 1.92 (the
// policy keeps k = 3 on the pipelined kernel); a WN = 4 form for 64-row convs (576 staged columns: 11 spilled registers) was +5 % / +-0 at
// k = 7 / 11 and is not built.
// RING = 0: one A-fragment register set per staging round (CM * KT taps), every entry re-loaded for the next round right after
// its use.  RING = D > 0 (long tap loops, CM = 1): a ring of D taps -- two row blocks' whole-chunk sets (16 * KT registers)
// plus 128 accumulators do not fit two waves per SIMD at KT = 7 / 11.  Tap g lives in slot g % D and the slot is re-loaded right
// after tap g with the next tap that will use it: g + D of this chunk, or -- from the last D taps -- tap (g % D) of the NEXT
// chunk, whose first D taps restart at slot 0.  Loads run 2-4 taps (3 000-6 000 matrix-pipe cycles) ahead of their use.
template <int KT, int NI, int HALO, int CM, int RING = 0, int WN = 1>
__global__ __launch_bounds__(256, 2) void conv_blk_kernel(const ConvArgs a) {
    constexpr int MI = 2;
    static_assert(RING == 0 || (CM == 1 && RING <= KT), "the A ring serves one chunk per round");
    static_assert(WN == 1 || WN == 2, "4 waves: 4 x 1 or 2 x 2");
    constexpr int WMW = 4 / WN;                // waves along M
    constexpr int NT = 32 * NI;                // output columns per WAVE
    constexpr int NTW = NT * WN;               // ... per workgroup
    constexpr int S = NTW + HALO;              // staged columns
    constexpr int NST = (4 * S) / 256;         // staging items (column x channel quad) per thread and chunk
    constexpr int BUF = 4 * S;                 // uint4 per chunk: [plane hi|lo][octet h][S]
    constexpr int VT = CM * KT;                // taps per staging round
    static_assert(S % 64 == 0, "the channel quad of a staging item must be wave-uniform");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [2][CM][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int nbx = gridDim.x;   // XCD-contiguous tile runs, see conv_f16x3.hip
    int bx = ((nbx & 7) == 0 && !a.lens) ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (a.rev) bx = nbx - 1 - bx;   // descending tile order: start where the previous launch stopped writing (ConvArgs::rev)
    int rg = blockIdx.y;         // 256-row group
    if (a.row_groups > 0) { rg = bx % a.row_groups; bx /= a.row_groups; }   // row group fastest, see ConvArgs
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int q0 = tile * NTW;
    const int wm = wave / WN, wnc = (wave % WN) * NT;     // this wave's row-block pair and first column inside the workgroup's tile
    if (a.lens) {   // ragged batch: tiles beyond the utterance's valid length, see conv_f16x3.hip
        const long long lv = (long long)a.lens[item] * a.len_mul;
        const long long first_out = (long long)q0 * a.up - a.up_pad;
        if ((a.up > 1 || a.Tout == a.Tin) && first_out >= lv * a.up) return;
    }
    const int mb0 = (rg * WMW + wm) * MI;   // first 32-row block of this wave (host: M % (256 / WN) == 0)

    const int up = a.up;
    const int qw = q0 + wnc + l31;
    const bool fast = (up == 1) && (q0 + NTW <= a.Tq);
    const int lane_off = (4 * hi) * a.Tout + qw;
    const float asc = a.acc_scale;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int mb = mb0 + mi;
        const size_t wave_base = ((size_t)item * a.Cout + (size_t)mb * 32) * a.Tout;
        if (fast) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = a.bias ? a.bias[mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] : 0.f;
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[mi][t][r] = bv;
            }
            if (a.res) {
                const float* rp = a.res + wave_base;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* rr_ = rp + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
                    for (int t = 0; t < NI; ++t) acc[mi][t][r] += rr_[lane_off + 32 * t];
                }
            }
            if (a.mode != 0) {
                const float* yp = a.y + wave_base;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* yr_ = yp + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
                    for (int t = 0; t < NI; ++t) acc[mi][t][r] += yr_[lane_off + 32 * t];
                }
            }
        } else if (!a.res && a.mode == 0) {   // transposed convs / ragged last tiles without residual: bias only
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int o = (up == 1) ? m : m / up;
                const float bv = a.bias ? a.bias[o] : 0.f;
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[mi][t][r] = bv;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int o = (up == 1) ? m : m / up;
                const int rr = m - o * up;
                const float bv = a.bias ? a.bias[o] : 0.f;
                const size_t rowoff = ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    const int q = qw + 32 * t;
                    const int n = q * up + rr - a.up_pad;
                    float v = bv;
                    if (q < a.Tq && n >= 0 && n < a.Tout) {
                        if (a.res) v += a.res[rowoff + n];
                        if (a.mode != 0) v += a.y[rowoff + n];
                    }
                    acc[mi][t][r] = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NI; ++t) acc[mi][t] *= asc;
    }

    const float* xb = a.x + (size_t)item * (size_t)a.xbs;
    const int tbase = q0 - a.halo_left;
    int Tv = a.Tin;
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    const float kpos = 16.f, kneg = 16.f * a.slope_in;

    float range_max = 0.f;
    float xs[CM][NST][4];
    auto stage_load = [&](int round) {   // the CM chunks of a round; unconditional loads from clamped addresses
#pragma unroll
        for (int cc = 0; cc < CM; ++cc)
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int ibase = wave * 64 + 256 * it;          // wave-uniform
                const int qd = ibase / S;                        // channel quad 0..3
                const int col = ibase - qd * S + lane;
                int t = tbase + col;
                if (a.pad_reflect) {
                    t = t < 0 ? -t : t;
                    t = t > Tv - 1 ? 2 * (Tv - 1) - t : t;
                }
                t = t < 0 ? 0 : t;
                t = t > a.Tin - 1 ? a.Tin - 1 : t;
                const int ch0 = (round * CM + cc) * KC16 + 4 * qd;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int ch = ch0 + e;
                    ch = ch > a.Cin - 1 ? a.Cin - 1 : ch;
                    xs[cc][it][e] = xb[(size_t)ch * a.Tin + t];
                }
            }
    };
    auto stage_store = [&](int round, int buf) {
#pragma unroll
        for (int cc = 0; cc < CM; ++cc) {
            uint2* dst = reinterpret_cast<uint2*>(smem4 + (buf * CM + cc) * BUF);
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int ibase = wave * 64 + 256 * it;
                const int qd = ibase / S;
                const int col = ibase - qd * S + lane;
                const int t = tbase + col;
                const bool tok = (col < a.wd) && (a.pad_reflect || ((t >= 0) && (t < Tv)));
                const int ch0 = (round * CM + cc) * KC16 + 4 * qd;
                struct { uint2 u; } fh, fl;
                stage4_f16((tok && (ch0 + 0) < a.Cin) ? xs[cc][it][0] : 0.f, (tok && (ch0 + 1) < a.Cin) ? xs[cc][it][1] : 0.f,
                           (tok && (ch0 + 2) < a.Cin) ? xs[cc][it][2] : 0.f, (tok && (ch0 + 3) < a.Cin) ? xs[cc][it][3] : 0.f,
                           kpos, kneg, range_max, fh.u, fl.u);
                const int o2 = (((qd >> 1) * S + col) << 1) + (qd & 1);
                dst[o2] = fh.u;
                dst[4 * S + o2] = fl.u;
            }
        }
    };

    // A fragments [mb][chunk][tap][plane][lane] x uint4: one register set per row block for a whole round, entry
    // (cc, g) re-loaded for the NEXT round right after its last use (conv_f16x3.hip); the reload after the last round
    // reads the next row block / the allocation pad (conv_build()).
    constexpr size_t kRound = (size_t)VT * 128;
    const size_t mbs = (size_t)a.nchunks * (KT * 128);   // uint4 per row block
    const uint4* wa = static_cast<const uint4*>(a.wp) + (size_t)mb0 * mbs + lane;
    constexpr int NA = RING > 0 ? RING : VT;             // A-fragment register sets held per row block
    FragB a_h[MI][NA], a_l[MI][NA];

    const int rd0 = hi * S + wnc + l31 + a.halo_left + a.off0;
    const int dstep = a.dstep;

    stage_load(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int v = 0; v < NA; ++v) {
            a_h[mi][v].u = wa[mi * mbs + v * 128];
            a_l[mi][v].u = wa[mi * mbs + v * 128 + 64];
        }
    AMP_PIN_VMEM();
    stage_store(0, 0);
    __syncthreads();

    const int nrounds = a.nchunks / CM;   // host: nchunks % CM == 0
    for (int c = 0; c < nrounds; ++c) {
        const bool more = (c + 1) < nrounds;
        stage_load(more ? c + 1 : c);
        AMP_PIN_VMEM();
        if constexpr (RING == 0) wa += kRound;
        const uint4* base = smem4 + (c & 1) * (CM * BUF) + rd0;
#pragma unroll
        for (int cc = 0; cc < CM; ++cc)
#pragma unroll
            for (int g = 0; g < KT; ++g) {
                const int v = RING > 0 ? g % NA : cc * KT + g;          // register set of this tap
                // RING: the tap the slot serves next -- g + RING of this chunk, or tap g % RING of the next chunk
                const int vn = RING > 0 ? (g + RING < KT ? g + RING : KT + g % NA) : v;
                const uint4* bg = base + cc * BUF + g * dstep;
                // BH = 2 (ring form): the tap's B fragments in two halves of NI / 2 column tiles -- 16 registers instead of 32
                // (what lets a ring of 4 taps fit); every accumulator still sees hh, hl, lh of this tap in that order
                constexpr int BH = ((RING > 3 || WN > 1) && NI % 2 == 0) ? 2 : 1;   // (WN > 1: the wider staging holds 8-24 more registers)
                constexpr int NB = NI / BH;
#pragma unroll
                for (int th = 0; th < BH; ++th) {
                    FragB bh[NB], bl[NB];
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        bh[t].u = bg[32 * (th * NB + t)];
                        bl[t].u = bg[2 * S + 32 * (th * NB + t)];
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v].h, bh[t].h, acc[mi][th * NB + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v].h, bl[t].h, acc[mi][th * NB + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[mi][v].h, bh[t].h, acc[mi][th * NB + t], 0, 0, 0);
                        if (th == BH - 1) {
                            a_h[mi][v].u = wa[mi * mbs + vn * 128];
                            a_l[mi][v].u = wa[mi * mbs + vn * 128 + 64];
                        }
                    }
                }
                AMP_PIN_VMEM();
            }
        if constexpr (RING > 0) wa += kRound;   // the ring's offsets are relative to the chunk being computed
        if (more) stage_store(c + 1, (c + 1) & 1);
        __syncthreads();
    }

    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);

    // ---- epilogue: undo the operand scaling, MRF mean, activation-on-store, (polyphase) scatter: conv_f16x3.hip's ----
    const float slope_out = a.slope_out;
    const bool exact_div = a.mode == 2;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int mb = mb0 + mi;
        if (fast) {
            float* yp = a.y + ((size_t)item * a.Cout + (size_t)mb * 32) * a.Tout;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* yr_ = yp + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    float v = acc[mi][t][r] * a.inv_scale;
                    if (exact_div) v = v / a.div;
                    v = v > 0.f ? v : v * slope_out;
                    yr_[lane_off + 32 * t] = v;
                }
            }
        } else if (up > 1 && (up & 3) == 0 && (a.up_pad & 3) == 0 && a.mode == 0) {
            // polyphase scatter, stride % 4 == 0: one float4 per lane and register quad (conv_f16x3.hip)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m0 = mb * 32 + 8 * j + 4 * hi;
                const int o = m0 / up;
                const int ph = m0 - o * up;
                float* yrow = a.y + ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    const int q = qw + 32 * t;
                    const int n0 = q * up + ph - a.up_pad;
                    if (q < a.Tq && n0 >= 0 && n0 + 3 < a.Tout) {
                        float4 v;
                        v.x = acc[mi][t][4 * j + 0] * a.inv_scale;
                        v.y = acc[mi][t][4 * j + 1] * a.inv_scale;
                        v.z = acc[mi][t][4 * j + 2] * a.inv_scale;
                        v.w = acc[mi][t][4 * j + 3] * a.inv_scale;
                        v.x = v.x > 0.f ? v.x : v.x * slope_out;
                        v.y = v.y > 0.f ? v.y : v.y * slope_out;
                        v.z = v.z > 0.f ? v.z : v.z * slope_out;
                        v.w = v.w > 0.f ? v.w : v.w * slope_out;
                        *reinterpret_cast<float4*>(yrow + n0) = v;
                    }
                }
            }
        } else if (up == 2 && a.mode == 0) {
            struct __attribute__((packed, aligned(4))) F2 { float a, b; };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int m0 = mb * 32 + 8 * j + 4 * hi + 2 * h2;
                    const int o = m0 >> 1;
                    float* yrow = a.y + ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
                    for (int t = 0; t < NI; ++t) {
                        const int q = qw + 32 * t;
                        const int n0 = 2 * q - a.up_pad;
                        float v0 = acc[mi][t][4 * j + 2 * h2] * a.inv_scale;
                        float v1 = acc[mi][t][4 * j + 2 * h2 + 1] * a.inv_scale;
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
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int o = (up == 1) ? m : m / up;
                const int rr = m - o * up;
                const size_t rowoff = ((size_t)item * a.Cout + o) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    const int q = qw + 32 * t;
                    const int n = q * up + rr - a.up_pad;
                    if (q < a.Tq && n >= 0 && n < a.Tout) {
                        float v = acc[mi][t][r] * a.inv_scale;
                        if (exact_div) v = v / a.div;
                        v = v > 0.f ? v : v * slope_out;
                        a.y[rowoff + n] = v;
                    }
                }
            }
        }
    }
}

template <int KT, int NI, int HALO, int CM, int RING = 0, int WN = 1>
static hipError_t launch_blk_one(const ConvArgs& a, hipStream_t stream) {
    constexpr int S = 32 * NI * WN + HALO;
    const size_t lds = (size_t)2 * CM * 4 * S * sizeof(uint4);
    if (hipError_t e = ensure_dynamic_lds<&conv_blk_kernel<KT, NI, HALO, CM, RING, WN>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.tiles_per_item), (unsigned)(a.M / (256 / WN)));
    if (a.row_groups > 0) grid = dim3((unsigned)(a.B * a.tiles_per_item * a.row_groups), 1u);
    note_kernel("conv_blk_kernel", KT, NI, HALO, CM, RING, WN);
    note_conv_work(a, KT, grid);
    hipLaunchKernelGGL((conv_blk_kernel<KT, NI, HALO, CM, RING, WN>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// Tile width (columns) of the blocked kernel for this tap count, chunks per round and staged halo, 0 = not covered.
//   KT = 2 (transposed convs, 1 halo column): 96-column tiles -- Tq = T + 1 columns cut into 128s leaves the first
//   up-sampling layer (T = 256) a third tile with ONE column; one or two chunks per round (180 / 227 VGPRs);
//   KT = 3: 128-column tiles, one chunk per round (244 VGPRs; two chunks' A fragments for two row blocks are 96
//   registers more than the 256 of two waves per SIMD hold: 96-column tiles still spill 17).
//   KT = 7 / 11: 128-column tiles, one chunk per round, A-fragment ring of AMP_BLK_RING taps (4, with the B fragments of a tap
//   in two halves: 253 VGPRs; 3 with whole-tap B fragments 252 and 3 % slower, 4 with whole-tap B fragments spills 23).
#ifndef AMP_BLK_RING
#define AMP_BLK_RING 4
#endif
int AMP_CAT(conv_blk_nt_kt, AMP_KT)(int cm, int halo_total) {
    constexpr int KT = AMP_KT;
    if (KT == 2) return ((cm == 1 || cm == 2) && halo_total <= 32) ? 96 : 0;
    return (cm == 1 && halo_total <= 64) ? 128 : 0;
}

// cm = chunks per staging round, wn = waves along the columns (1 | 2: 256- | 128-row groups); the caller guarantees
// M % (256 / wn) == 0, nchunks % cm == 0, tanh_out == 0 and tiles of wn * conv_blk_nt_kt*(cm, halo) columns
hipError_t AMP_CAT(launch_conv_blk_kt, AMP_KT)(int cm, int wn, const ConvArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    if constexpr (KT == 2) {
        if (wn != 1) return hipErrorInvalidValue;
        if (cm == 2) return launch_blk_one<KT, 3, 32, 2>(a, stream);
        return launch_blk_one<KT, 3, 32, 1>(a, stream);
    } else if constexpr (KT == 3) {
        if (cm != 1) return hipErrorInvalidValue;
        if (wn == 2) return launch_blk_one<KT, 4, 64, 1, 0, 2>(a, stream);
        if (wn != 1) return hipErrorInvalidValue;
        return launch_blk_one<KT, 4, 64, 1>(a, stream);
    } else {
        if (cm != 1) return hipErrorInvalidValue;
        // (two waves along the columns stage 320 columns per chunk -- 8 more staging registers than the 192 of the 256-row form: a
        //  ring of 3 taps keeps the kernel free of scratch)
        if (wn == 2) return launch_blk_one<KT, 4, 64, 1, 3, 2>(a, stream);
        if (wn != 1) return hipErrorInvalidValue;
        return launch_blk_one<KT, 4, 64, 1, AMP_BLK_RING>(a, stream);
    }
}

}  // namespace amp
