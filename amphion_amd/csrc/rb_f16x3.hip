// A whole ResBlock1 in one launch on the gfx950 f16 matrix cores (split-f16 operands, see conv_f16x3.hip) -- round 3:
//
//     for p in 0 .. n-1:   x = x + c2_p( lrelu( c1_p( lrelu(x) ) ) )        [last pair: + the running MRF sum, / num_kernels]
//
// i.e. ResBlock1.forward (hifigan.py:93-100) with its three (dilation d_p, dilation 1) pairs, for the narrow late stages
// (C = 32 / 64) where a fused PAIR (pair_f16x3.hip) is bounded by per-element work, not by the matrix pipe: 787 other VALU
// instructions beside 72 MFMAs per wave at C = 32, k = 3 (profiles/r2_ap_pair_counters_by_width.txt) -- global loads with
// 64-bit addresses, the staging split, the residual fetch, the MRF read, the store and the x round trip through HBM are paid
// per pair.  Here they are paid once per RESBLOCK (SURVEY.md §7 "hard parts" / App. B: "time-major tiles, whole ResBlock
// fused in LDS"):
//
//   workgroup = all C channels x W columns of one batch item, W = NT + 2 * RH with RH = sum_p (k-1)/2 * (d_p + 1) the
//               one-sided receptive field of the block (12 / 36 / 60 columns for k = 3 / 7 / 11 with dilations 1, 3, 5):
//               every conv is evaluated on all W columns, and the columns whose inputs lay outside the tile (RH from either
//               edge after the sixth conv) are simply never stored -- the neighbouring tile computes them.
//   registers   the wave's x tile in the MFMA C layout (fp32, exact): it is the residual of the pair, and it becomes the
//               next pair's x in place -- a lane's column never moves, so the residual never leaves the lane.
//   LDS         ONE tile [16-channel chunk][plane hi|lo][octet][G + W + G columns][8 x f16] in B-fragment layout (the layout
//               of pair_f16x3.hip's xt tile): each conv reads it (all waves), barrier, and its output -- leaky ReLU, the
//               conv's zero padding outside the utterance, x16, hi/lo split, all in registers -- is written over it in
//               place (every wave over its own columns), barrier.  The G guard columns either side are zero and are never
//               written: they stand for the tile's unknown outside and keep every read in bounds.
//   HBM         x is read once, y written once per resblock: 1/3 of the fused pairs' traffic, 1/15 of the unfused convs'.
//
// Per output element the operation order (bias, chunks, taps, the three MFMAs of a term, the fma of the residual, the MRF
// add and divide) is that of pair_f16x3.hip / conv_f16x3.hip, so the result is bit-identical to three fused pairs for
// every tiling (tests/test_gpu_resblock.py).
//
// Round 4: PERMUTED A ROWS.  In the MFMA C layout a lane holds rows 8 a + 4 hi + j (a, j < 4) of its column: four 4-channel
// groups 8 apart, so the seam wrote 8-B halves of a 16-B octet at a 16-B lane stride -- SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS
// 61-90 % at C = 32 (profiles/r3_sq_counters_in_forward.txt).  Which weight row an A-fragment lane fetches is only an address:
// lane (l & 31) = 8 a + 4 b + j now fetches the packed fragment of row 16 b + 4 a + j, and the lane's sixteen accumulators are the
// sixteen CONSECUTIVE channels 32 wm + 16 hi + r of its column = one 16-channel chunk = two whole octets: the seam is two
// conflict-free ds_write_b128 per plane instead of four ds_write_b64, with no instruction added anywhere (every output element is
// the same dot product, computed in another row of the tile: same bits).
//
// Compiled once per tap count:  -DAMP_KT=<3|5|7|11>.
#include "amp_internal.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union FragR {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)

// RB_G guard columns either side of the LDS tile (>= (k-1)/2 * max dilation): 32, or 16 where the tile of all channels would not
// fit otherwise (C = 128: 8 chunks x 64 B x (256 + 2 x 16) columns = 147 KB)

// RING = D > 0: the A fragments as a ring of D taps (conv_blk_f16x3.hip: tap g lives in slot g % D and the slot is re-loaded right after
// its use with tap g + D of this chunk or tap g % D of the next one) -- 8 * D registers instead of 8 * KT: the k = 7 form spilled 14
// registers with a whole chunk's set beside the 128 accumulator / residual registers.
template <int KT, int WM, int WN, int NI, int RING = 0, int RB_G = 32>
__global__ __launch_bounds__(64 * WM * WN, 2) void rb_f16x3_kernel(const RbArgs a) {
    static_assert(RING == 0 || RING < KT, "a ring shorter than one chunk");
    constexpr int NTHR = 64 * WM * WN;
    constexpr int W = 32 * NI * WN;           // columns per tile (every conv is evaluated on all of them)
    constexpr int WL = W + 2 * RB_G;          // LDS row length
    constexpr int H2 = (KT - 1) / 2;
    constexpr int NCH = 2 * WM;               // 16-channel chunks (C = 32 * WM)
    constexpr int CHS = 4 * WL;               // uint4 per chunk [plane][octet][WL]
    constexpr int C = 32 * WM;
    static_assert(NI % 2 == 0, "B fragments are read in halves");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];   // [NCH][plane][octet][WL]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    const int nbx = gridDim.x;  // XCD-contiguous tile runs, see conv_f16x3.hip (ragged batches keep the dispatch order)
    int bx = ((nbx & 7) == 0 && !a.lens) ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (a.rev) bx = nbx - 1 - bx;   // descending tile order, PairArgs::rev
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int T = a.T;
    const int RH = a.rh;
    const int NT = W - 2 * RH;                // output columns per tile
    const int O0 = tile * NT;                 // first output column
    int Tv = T;                               // valid columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    if (O0 >= Tv) return;                     // tile beyond the utterance: unspecified by contract (workgroup-uniform)
    const int q0 = O0 - RH;                   // global column of tile column 0

    // zero guards
    for (int i = tid; i < NCH * 4 * 2 * RB_G; i += NTHR) {
        const int row = i / (2 * RB_G), g = i - row * (2 * RB_G);
        smem4[row * WL + (g < RB_G ? g : W + g)] = make_uint4(0u, 0u, 0u, 0u);
    }

    // this lane's columns: tile column colw + 32 t, global column q0 + that
    const int colw = wn * (32 * NI) + l31;
    int qcl[NI];                              // clamped global column (addresses)
    bool qok[NI];                             // inside the utterance: what every conv sees as its input there, else zero
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int q = q0 + colw + 32 * t;
        qok[t] = (q >= 0) && (q < Tv);
        qcl[t] = q < 0 ? 0 : (q < T ? q : T - 1);
    }
    // this lane's rows: channels 32 * wm + 16 * hi + r (permuted A rows, see the header)
    const float* xr = a.x + (size_t)item * C * T + (size_t)(32 * wm + 16 * hi) * T;
    f32x16 rv[NI];                            // x (then x + pair_0(x), ...) of this lane's rows / columns: the residual
#pragma unroll
    for (int t = 0; t < NI; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[t][r] = xr[(size_t)r * T + qcl[t]];

    // A fragments [mb][chunk][tap][plane][lane] x uint4 (conv_build): one register set, re-loaded one chunk ahead; the reload
    // during a conv's last chunk fetches the next conv's first chunk
    constexpr size_t MBS = (size_t)NCH * (KT * 128);
    constexpr int NA = RING > 0 ? RING : KT;  // A-fragment register sets
    FragR a_h[NA], a_l[NA];
    // the packed fragment this lane fetches: A row (lane & 31) = 8 a + 4 b + j holds weight row 16 b + 4 a + j
    const int wlane = (lane & 32) | (16 * ((lane >> 2) & 1) + 4 * ((lane >> 3) & 3) + (lane & 3));
    {
        const uint4* w0 = static_cast<const uint4*>(a.wp1[0]) + (size_t)wm * MBS + wlane;
#pragma unroll
        for (int g = 0; g < NA; ++g) {
            a_h[g].u = w0[g * 128];
            a_l[g].u = w0[g * 128 + 64];
        }
    }
    AMP_PIN_VMEM();

    float range_max = 0.f;       // largest |staged operand| (x16 applied): beyond 65504 it left the f16 range (a.range_flag)
    const float kpos = 16.f, kneg = 16.f * a.slope;
    // x (registers) -> lrelu, the conv's zero padding, x16, hi / lo -> the LDS tile; what stage_store of pair_f16x3.hip does
    auto stage_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int col = RB_G + colw + 32 * t;
            const bool ok = qok[t];        // (a select, not a zero factor: beyond a ragged utterance's end the rows hold whatever the
                                           //  workspace held -- possibly NaN -- and NaN * 0 would reach the valid columns through the taps)
            // channels 32 * wm + 16 * hi + r  ->  chunk 2 * wm + hi, octet r >> 3: two whole 16-B units per plane
            const int o4 = (2 * wm + hi) * CHS + col;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                struct { uint2 u; } fh0, fl0, fh1, fl1;
                stage4_f16(ok ? rv[t][8 * o + 0] : 0.f, ok ? rv[t][8 * o + 1] : 0.f, ok ? rv[t][8 * o + 2] : 0.f,
                           ok ? rv[t][8 * o + 3] : 0.f, kpos, kneg, range_max, fh0.u, fl0.u);
                stage4_f16(ok ? rv[t][8 * o + 4] : 0.f, ok ? rv[t][8 * o + 5] : 0.f, ok ? rv[t][8 * o + 6] : 0.f,
                           ok ? rv[t][8 * o + 7] : 0.f, kpos, kneg, range_max, fh1.u, fl1.u);
                smem4[o4 + o * WL] = make_uint4(fh0.u.x, fh0.u.y, fh1.u.x, fh1.u.y);
                smem4[o4 + o * WL + 2 * WL] = make_uint4(fl0.u.x, fl0.u.y, fl1.u.x, fl1.u.y);
            }
        }
    };

    f32x16 acc[NI];
    // one conv over the LDS tile: acc = bias * sc + sum over chunks, taps of W' x tile[col + (g - H2) * d]
    auto conv = [&](const void* wp, const float* bias, float sc, int d, const void* wp_next) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bias[32 * wm + 16 * hi + r] * sc;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
        }
        const uint4* wa = static_cast<const uint4*>(wp) + (size_t)wm * MBS + wlane;
        const uint4* wn0 = static_cast<const uint4*>(wp_next) + (size_t)wm * MBS + wlane;
        const int rd = hi * WL + RB_G + colw - H2 * d;
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            const uint4* wcur = wa + (size_t)c * (KT * 128);
            const uint4* wan = (c + 1) < NCH ? wa + (size_t)(c + 1) * (KT * 128) : wn0;
            const uint4* base = smem4 + c * CHS + rd;
#pragma unroll
            for (int g = 0; g < KT; ++g) {
                const int v = RING > 0 ? g % NA : g;
                const uint4* bg = base + g * d;
                constexpr int NB = NI / 2;      // the tap's B fragments in two halves (16 registers instead of 32)
#pragma unroll
                for (int th = 0; th < 2; ++th) {
                    FragR bh[NB], bl[NB];
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        bh[t].u = bg[32 * (th * NB + t)];
                        bl[t].u = bg[2 * WL + 32 * (th * NB + t)];
                    }
#pragma unroll
                    for (int t = 0; t < NB; ++t)
                        acc[th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[v].h, bh[t].h, acc[th * NB + t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NB; ++t)
                        acc[th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[v].h, bl[t].h, acc[th * NB + t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NB; ++t)
                        acc[th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[v].h, bh[t].h, acc[th * NB + t], 0, 0, 0);
                }
                {   // this tap's register set is free: fetch the tap it serves next
                    const bool same = RING > 0 && g + RING < KT;
                    const uint4* src = same ? wcur + (size_t)(g + RING) * 128 : wan + (size_t)v * 128;
                    a_h[v].u = src[0];
                    a_l[v].u = src[64];
                }
                AMP_PIN_VMEM();
            }
        }
    };

    stage_x();
    __syncthreads();

    const int np = a.np;
    // ROLLED over the pairs (the arguments of pair p are picked out of the kernel-argument arrays with scalar loads): one copy of
    // the pair's code instead of three -- the unrolled k = 3 kernel is 43 KB of instructions against a 64-KB instruction cache
    // shared by two CUs whose waves sit in different phases of it
#pragma unroll 1
    for (int p = 0; p < AMP_RB_MAX_PAIRS; ++p) {
        if (p < np) {   // workgroup-uniform
            const bool last = (p + 1 == np);
            // ---- c1: kernel KT, dilation d_p ----
            conv(a.wp1[p], a.bias1[p], a.sc1[p], a.dil[p], a.wp2[p]);
            __syncthreads();                      // every wave has read the tile
            {   // seam: xt = lrelu(c1(.)) -> the tile, in place (pair_f16x3.hip's seam)
                const float i1 = a.isc1[p];
                const float slope = a.slope;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    const int o4 = (2 * wm + hi) * CHS + RB_G + colw + 32 * t;
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        struct { uint2 u; } fh0, fl0, fh1, fl1;
                        seam4_f16(acc[t][8 * o + 0], acc[t][8 * o + 1], acc[t][8 * o + 2], acc[t][8 * o + 3], i1, slope, qok[t], range_max, fh0.u, fl0.u);
                        seam4_f16(acc[t][8 * o + 4], acc[t][8 * o + 5], acc[t][8 * o + 6], acc[t][8 * o + 7], i1, slope, qok[t], range_max, fh1.u, fl1.u);
                        smem4[o4 + o * WL] = make_uint4(fh0.u.x, fh0.u.y, fh1.u.x, fh1.u.y);
                        smem4[o4 + o * WL + 2 * WL] = make_uint4(fl0.u.x, fl0.u.y, fl1.u.x, fl1.u.y);
                    }
                }
            }
            __syncthreads();
            // ---- c2: kernel KT, dilation 1 ----
            conv(a.wp2[p], a.bias2[p], a.sc2[p], 1, last ? a.wp2[p] : a.wp1[p + 1 < AMP_RB_MAX_PAIRS ? p + 1 : p]);
            {   // x = x + c2(.)   (one fma per element, as the pairs' epilogue)
                const float i2 = a.isc2[p];
#pragma unroll
                for (int t = 0; t < NI; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[t][r] = __builtin_fmaf(acc[t][r], i2, rv[t][r]);
            }
            if (!last) {
                __syncthreads();                  // every wave has read the tile
                stage_x();
                __syncthreads();
            }
        }
    }

    // ---------------- epilogue: MRF accumulate, store the NT output columns of the tile ----------------
    {
        const int mode = a.mode;
        float* yr = a.y + (size_t)item * C * T + (size_t)(32 * wm + 16 * hi) * T;
        if (mode != 0) {   // workgroup-uniform
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = yr[(size_t)r * T + qcl[t]];
#pragma unroll
            for (int t = 0; t < NI; ++t) rv[t] += acc[t];
            if (mode == 2) {
#pragma unroll
                for (int t = 0; t < NI; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[t][r] = rv[t][r] / a.div;
            }
        }
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int col = colw + 32 * t;
            const int q = q0 + col;
            if (col >= RH && col < W - RH && q < T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yr[(size_t)r * T + qcl[t]] = rv[t][r];
            }
        }
    }
    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);
}

template <int KT, int WM, int WN, int NI, int RING = 0, int RB_G = 32>
static hipError_t launch_rb_one(const RbArgs& a, hipStream_t stream) {
    constexpr int W = 32 * NI * WN;
    const size_t lds = (size_t)2 * WM * 4 * (W + 2 * RB_G) * sizeof(uint4);
    if (hipError_t e = ensure_dynamic_lds<&rb_f16x3_kernel<KT, WM, WN, NI, RING, RB_G>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.tiles_per_item));
    note_kernel("rb_f16x3_kernel", KT, WM, WN, NI, RING, RB_G);
    note_work(grid.x, a.np * 2 * 2.0 * a.C * a.C * KT * (double)a.T * a.B / 1e9, 4.0 * a.B * (double)a.C * a.T * (2 + (a.mode ? 1 : 0)) / 1e6,
              "whole ResBlock C=%d k=%d T=%d B=%d: %d convs%s", a.C, KT, a.T, a.B, 2 * a.np, a.mode ? " +sum" : "");
    hipLaunchKernelGGL((rb_f16x3_kernel<KT, WM, WN, NI, RING, RB_G>), grid, dim3(64 * WM * WN), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// Tile width W (columns evaluated per workgroup; outputs per tile = W - 2 * rh) for C channels in form `wide`
// (1: eight waves, one workgroup per CU; 0: four waves, two per CU -- C = 32, and C = 64 at k <= 5 since round 6), or 0 when not covered.
int AMP_CAT(rb_tile_kt, AMP_KT)(int C, int max_dil, int wide) {
    constexpr int KT = AMP_KT;
    const int reach = (KT - 1) / 2 * max_dil;
    if (reach > 32) return 0;
    if (C == 32) return wide ? 1024 : 512;
    if (C == 64) return wide ? 512 : ((KT <= 5 && reach <= 16) ? 256 : 0);   // four waves x 256 columns, 16 guard columns: 74 KB, two workgroups per CU
    if (C == 128) return (wide && KT <= 5 && reach <= 16) ? 256 : 0;
    return 0;
}

hipError_t AMP_CAT(launch_rb_kt, AMP_KT)(const RbArgs& a, int wide, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    constexpr int RING = KT >= 7 ? 4 : 0;
    if (a.C == 32) return wide ? launch_rb_one<KT, 1, 8, 4, RING>(a, stream) : launch_rb_one<KT, 1, 4, 4, RING>(a, stream);
    if (a.C == 64 && wide) return launch_rb_one<KT, 2, 4, 4, RING>(a, stream);
    if constexpr (KT <= 5) {
        if (a.C == 64 && !wide) return launch_rb_one<KT, 2, 2, 4, 0, 16>(a, stream);
    }
    if constexpr (KT <= 5) {
        if (a.C == 128 && wide) return launch_rb_one<KT, 4, 2, 4, 0, 16>(a, stream);
    }

    return hipErrorInvalidValue;
}

}  // namespace amp
