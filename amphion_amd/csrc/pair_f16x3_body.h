// The body of the fused ResBlock pair kernel (pair_f16x3.hip: see there for the algorithm), as a device function of (arguments, index of
// this workgroup, number of workgroups): pair_f16x3_kernel runs it over a grid of its own; pair3_kernel (pair3_f16x3.hip) runs the bodies
// of the three resblocks of a generator stage side by side in ONE grid.
#pragma once
#include "amp_internal.h"

namespace amp {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union Frag {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)

template <int KT, int WM, int WN, int NI, int SX>
__device__ __forceinline__ void pair_f16x3_body(const PairArgs& a, const int bid, const int nbx) {
    constexpr int N1 = 32 * NI * WN;          // conv1 output columns = xt columns conv2 reads
    constexpr int H2 = (KT - 1) / 2;
    constexpr int NT = N1 - 2 * H2;           // output columns per workgroup
    constexpr int XT = N1 + 12;               // xt row length: + the read overrun of the unused tail columns
    constexpr int NCH = 2 * WM;               // 16-channel chunks (C = 32 * WM)
    constexpr int XBUF = 4 * SX;              // uint4 per x staging buffer [plane][octet][SX]
    constexpr int XTCH = 4 * XT;              // uint4 per xt chunk       [plane][octet][XT]
    constexpr int NST = (4 * SX) / 256;       // staging items (column x channel quad) per thread
    static_assert(SX % 64 == 0, "staging items must have a wave-uniform channel quad");
    static_assert(KT - 1 <= 12, "xt pad");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [2][XBUF] + [NCH][XTCH]
    uint4* const xt4 = smem4 + 2 * XBUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    // nbx workgroups run this body, `bid` is this one's index among them: XCD-contiguous tile runs, see conv_f16x3.hip
    // (ragged batches keep the dispatch order: with utterances of different lengths a contiguous run per XCD would hand
    // one XCD the long utterances and another only tiles that exit at once -- measured 43.8 vs 48.7 ms padded, visit AD)
    int bx = ((nbx & 7) == 0 && !a.lens) ? (bid & 7) * (nbx >> 3) + (bid >> 3) : bid;
    if (a.rev) bx = nbx - 1 - bx;   // descending tile order: start where the previous launch stopped writing (PairArgs::rev)
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int q0 = tile * NT;                 // first output column
    const int C = 32 * WM;
    const int T = a.T;
    int Tv = T;                               // valid columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    // ragged batch: a tile that lies entirely beyond this utterance's valid length produces only samples the contract
    // leaves unspecified (nothing downstream reads them: every layer takes its input as zero / replicated beyond the valid
    // length) -- skip it.  A batch of 60..400-frame utterances is 40 % such tiles.
    if (q0 >= Tv) return;                     // block-uniform, before any barrier
    const int dil = a.dil;
    const int h1 = H2 * dil;

    const float* xb = a.x + (size_t)item * C * T;
    const int tbase = q0 - H2 - h1;           // global column of staged column 0
    const float kpos = 16.f, kneg = 16.f * a.slope;

    // Residual x at this lane's OUTPUT positions.  For C <= 64 (HBM-bound pairs) it is fetched here, next to the
    // staging loads of the same cache lines, and carried in registers: fetched again after phase 2 those lines
    // have left L2 and the residual costs a second HBM read of the tensor.  (C = 128 has no registers to spare
    // and is MFMA-bound; it re-reads in the epilogue.)
    constexpr bool RES_EARLY = WM < 4;
    const int colw0 = wn * (32 * NI) + l31;
    int qc[NI];
    bool okc[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int col = colw0 + 32 * t;
        const int q = q0 + col;
        okc[t] = (col < NT) && (q < T);
        qc[t] = q < T ? q : T - 1;
    }
    // this lane's rows: channels 32 * wm + 16 * hi + r (permuted A rows, round 4 -- see rb_f16x3.hip's header: a lane's accumulators
    // are one 16-channel chunk, the seam two conflict-free ds_write_b128 per plane; the same dot products, the same bits)
    const float* xres = a.x + (size_t)item * C * T + (size_t)(32 * wm + 16 * hi) * T;
    f32x16 rv[NI];
    if (RES_EARLY) {
#pragma unroll
        for (int t = 0; t < NI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[t][r] = xres[(size_t)r * T + qc[t]];
    }

    // ---------------- phase 1: conv1 ----------------
    f32x16 acc[NI];
    {
        const float s1 = a.sc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = a.bias1[32 * wm + 16 * hi + r] * s1;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
        }
    }

    float range_max = 0.f;       // largest |staged operand| (x16 applied): beyond 65504 it left the f16 range (a.range_flag)
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
            const bool tok = (t >= 0) && (t < Tv);
            struct { uint2 u; } fh, fl;
            stage4_f16(tok ? xs[it][0] : 0.f, tok ? xs[it][1] : 0.f, tok ? xs[it][2] : 0.f, tok ? xs[it][3] : 0.f,
                       kpos, kneg, range_max, fh.u, fl.u);
            const int o2 = (((qd >> 1) * SX + col) << 1) + (qd & 1);
            dst[o2] = fh.u;
            dst[4 * SX + o2] = fl.u;
        }
    };

    // A fragments [mb][chunk][tap][plane][lane] x uint4, one register set, reloaded one chunk ahead
    // (conv_f16x3.hip); the reload during conv1's LAST chunk fetches conv2's first chunk.
    const int wlane = (lane & 32) | (16 * ((lane >> 2) & 1) + 4 * ((lane >> 3) & 3) + (lane & 3));   // A row 8a + 4b + j <- weight row 16b + 4a + j
    const uint4* wa1 = static_cast<const uint4*>(a.wp1) + (size_t)wm * NCH * (KT * 128) + wlane;
    const uint4* wa2 = static_cast<const uint4*>(a.wp2) + (size_t)wm * NCH * (KT * 128) + wlane;
    Frag a_h[KT], a_l[KT];

    const int colw = wn * (32 * NI) + l31;    // this lane's column inside the tile (n-tile 0)
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
        const uint4* wan = more ? wa1 + (size_t)(c + 1) * (KT * 128) : wa2;
        const uint4* base = smem4 + (c & 1) * XBUF + rd1;
#pragma unroll
        for (int g = 0; g < KT; ++g) {
            const uint4* bg = base + g * dil;
            Frag bh[NI], bl[NI];
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

    // ---------------- seam: xt = lrelu(conv1) -> LDS, split-f16 B layout ----------------
    {
        const float i1 = a.isc1;
        const float slope = a.slope;
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int col = colw + 32 * t;                   // xt column (tile-local)
            const int q = q0 - H2 + col;                     // its global column
            const bool qok = (q >= 0) && (q < Tv);           // conv2 zero-pads xt outside the utterance
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                struct { uint2 u; } fh0, fl0, fh1, fl1;
                seam4_f16(acc[t][8 * o + 0], acc[t][8 * o + 1], acc[t][8 * o + 2], acc[t][8 * o + 3], i1, slope, qok, range_max, fh0.u, fl0.u);
                seam4_f16(acc[t][8 * o + 4], acc[t][8 * o + 5], acc[t][8 * o + 6], acc[t][8 * o + 7], i1, slope, qok, range_max, fh1.u, fl1.u);
                // channels 32*wm + 16*hi + 8*o + i  ->  chunk 2*wm + hi, octet o: a whole 16-B unit
                const int o4 = (2 * wm + hi) * XTCH + o * XT + col;
                xt4[o4] = make_uint4(fh0.u.x, fh0.u.y, fh1.u.x, fh1.u.y);
                xt4[o4 + 2 * XT] = make_uint4(fl0.u.x, fl0.u.y, fl1.u.x, fl1.u.y);
            }
        }
    }
    {
        const float s2 = a.sc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = a.bias2[32 * wm + 16 * hi + r] * s2;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
        }
    }
    __syncthreads();

    // ---------------- phase 2: conv2 over the xt tile ----------------
    {
        const int rd2 = hi * XT + colw;
        for (int c = 0; c < NCH; ++c) {
            const uint4* wan = wa2 + (size_t)(c + 1) * (KT * 128);   // last: next mb block / pad
            const uint4* base = xt4 + c * XTCH + rd2;
#pragma unroll
            for (int g = 0; g < KT; ++g) {
                const uint4* bg = base + g;
                Frag bh[NI], bl[NI];
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
    // loads are unconditional from clamped addresses (batched, one wait), stores are predicated
    {
        const float i2 = a.isc2;
        const int mode = a.mode;
        float* yr = a.y + (size_t)item * C * T + (size_t)(32 * wm + 16 * hi) * T;
        if (!RES_EARLY) {
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[t][r] = xres[(size_t)r * T + qc[t]];
        }
#pragma unroll
        for (int t = 0; t < NI; ++t) acc[t] = acc[t] * i2 + rv[t];
        if (mode != 0) {   // wave-uniform
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[t][r] = yr[(size_t)r * T + qc[t]];
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
                for (int r = 0; r < 16; ++r) yr[(size_t)r * T + qc[t]] = acc[t][r];
            }
    }
    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);
}


}  // namespace amp
