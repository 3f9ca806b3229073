// Timing-only ablation of pair_f16x3_kernel (a generated copy of amphion_amd/csrc/pair_f16x3.hip's kernel with
// knock-out switches): which part of the fused ResBlock pair costs what on MI355X.  Results are WRONG under any bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iamphion_amd/csrc -DAMP_KT=11 -DAMP_ABL=<mask> pair_ablate.hip -o pair_ablate_<mask>
//   bits: 2 no chunk barriers | 4 no A-fragment reloads | 8 no staging loads | 16 no residual loads | 32 no conv1 MFMAs
//         64 no conv2 MFMAs | 128 B fragments always tap 0 (LDS broadcast-free same address) | 256 seam without split
// Regenerate with tests/experiments/make_pair_ablate.py after editing the product kernel.
#include "amp_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef AMP_ABL
#define AMP_ABL 0
#endif
#include <type_traits>
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
typedef float f32x16_ __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16_ mf_keep(f16x8_ a, f16x8_ b, f32x16_ c) { asm volatile("" ::"v"(a), "v"(b)); return c; }
#if AMP_ABL & 32
#define MF1(a, b, c, x, y, z) mf_keep(a, b, c)
#else
#define MF1(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#endif
#if AMP_ABL & 64
#define MF2(a, b, c, x, y, z) mf_keep(a, b, c)
#else
#define MF2(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union Frag {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)

template <int KT, int WM, int WN, int NI, int SX>
__global__ __launch_bounds__(256, 2) void pair_f16x3_kernel(const PairArgs a) {
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
    const float* xres = a.x + (size_t)item * C * T + (size_t)(32 * wm + 4 * hi) * T;
    f32x16 rv[NI];
    if (RES_EARLY) {
#pragma unroll
        for (int t = 0; t < NI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[t][r] = xres[(size_t)((r & 3) + 8 * (r >> 2)) * T + qc[t]];
    }

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
            const bool tok = (t >= 0) && (t < Tv);
            union { uint2 u; _Float16 h[4]; } fh, fl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = tok ? xs[it][e] : 0.f;
                v = v * (v > 0.f ? kpos : kneg);
                split_f16(v, fh.h[e], fl.h[e]);
            }
            const int o2 = (((qd >> 1) * SX + col) << 1) + (qd & 1);
            dst[o2] = fh.u;
            dst[4 * SX + o2] = fl.u;
        }
    };

    // A fragments [mb][chunk][tap][plane][lane] x uint4, one register set, reloaded one chunk ahead
    // (conv_f16x3.hip); the reload during conv1's LAST chunk fetches conv2's first chunk.
    const uint4* wa1 = static_cast<const uint4*>(a.wp1) + (size_t)wm * NCH * (KT * 128) + lane;
    const uint4* wa2 = static_cast<const uint4*>(a.wp2) + (size_t)wm * NCH * (KT * 128) + lane;
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
#if !(AMP_ABL & 8)
        stage_load(more ? c + 1 : c);
#endif
        AMP_PIN_VMEM();
        const uint4* wan = more ? wa1 + (size_t)(c + 1) * (KT * 128) : wa2;
        const uint4* base = smem4 + (c & 1) * XBUF + rd1;
#if AMP_ABL & 1024
        // B fragments one tap ahead: bh in two register sets (+4*NI VGPRs), bl in one; MFMA order [ah.bl][ah.bh][al.bh]:
        // bl(g+1) is fetched after the 3rd MFMA of tap g (6 MFMAs of cover), bh(g+2) after its 9th (a whole tap of cover)
        Frag bhs[2][NI], bl[NI];
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            bhs[0][t].u = base[32 * t];
            bl[t].u = base[2 * SX + 32 * t];
            if (KT > 1) bhs[1][t].u = base[dil + 32 * t];
        }
        static_for<KT>([&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int cur = g & 1;
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] = MF1(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
            if constexpr (g + 1 < KT) {
#pragma unroll
                for (int t = 0; t < NI; ++t) bl[t].u = base[(g + 1) * dil + 2 * SX + 32 * t];
            }
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] = MF1(a_h[g].h, bhs[cur][t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] = MF1(a_l[g].h, bhs[cur][t].h, acc[t], 0, 0, 0);
            if constexpr (g + 2 < KT) {
#pragma unroll
                for (int t = 0; t < NI; ++t) bhs[cur][t].u = base[(g + 2) * dil + 32 * t];
            }
            a_h[g].u = wan[g * 128];
            a_l[g].u = wan[g * 128 + 64];
            __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
            if constexpr (g + 1 < KT) __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
            if constexpr (g + 2 < KT) __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        });
#elif AMP_ABL & 512
        // software-pipelined B fragments without a second register set: MFMA order [ah.bh][al.bh][ah.bl]; the next tap's
        // bh is fetched as soon as the 6th MFMA has read it, its bl after the 9th
        Frag bh[NI], bl[NI];
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            bh[t].u = base[32 * t];
            bl[t].u = base[2 * SX + 32 * t];
        }
        static_for<KT>([&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr bool nxt = g + 1 < KT;
            const uint4* bn = base + (nxt ? (g + 1) * dil : g * dil);
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] = MF1(a_h[g].h, bh[t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] = MF1(a_l[g].h, bh[t].h, acc[t], 0, 0, 0);
            if constexpr (nxt) {
#pragma unroll
                for (int t = 0; t < NI; ++t) bh[t].u = bn[32 * t];
            }
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t] = MF1(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
            if constexpr (nxt) {
#pragma unroll
                for (int t = 0; t < NI; ++t) bl[t].u = bn[2 * SX + 32 * t];
            }
            a_h[g].u = wan[g * 128];
            a_l[g].u = wan[g * 128 + 64];
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
            if constexpr (nxt) __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
            if constexpr (nxt) __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        });
#else
#pragma unroll
        for (int g = 0; g < KT; ++g) {
            const uint4* bg = base + ((AMP_ABL & 128) ? 0 : g * dil);
            Frag bh[NI], bl[NI];
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                bh[t].u = bg[32 * t];
                bl[t].u = bg[2 * SX + 32 * t];
            }
#pragma unroll
            for (int t = 0; t < NI; ++t)
                acc[t] = MF1(a_h[g].h, bh[t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NI; ++t)
                acc[t] = MF1(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NI; ++t)
                acc[t] = MF1(a_l[g].h, bh[t].h, acc[t], 0, 0, 0);
#if !(AMP_ABL & 4)
            a_h[g].u = wan[g * 128];
            a_l[g].u = wan[g * 128 + 64];
#endif
            AMP_PIN_VMEM();
        }
#endif
        if (more) stage_store((c + 1) & 1);
#if !(AMP_ABL & 2)
        __syncthreads();
#endif
    }

    // ---------------- seam: xt = lrelu(conv1) -> LDS, split-f16 B layout ----------------
    {
        const float i1 = a.isc1;
        const float slope = a.slope;
        uint2* xt2 = reinterpret_cast<uint2*>(xt4);
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int col = colw + 32 * t;                   // xt column (tile-local)
            const int q = q0 - H2 + col;                     // its global column
            const bool qok = (q >= 0) && (q < Tv);           // conv2 zero-pads xt outside the utterance
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                union { uint2 u; _Float16 h[4]; } fh, fl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[t][4 * j + i] * i1;
                    v = v > 0.f ? v : v * slope;
                    v = qok ? v * 16.f : 0.f;
#if AMP_ABL & 256
                    fh.h[i] = (_Float16)v; fl.h[i] = (_Float16)0.f;
#else
                    split_f16(v, fh.h[i], fl.h[i]);
#endif
                }
                // channels 32*wm + 8*j + 4*hi + i  ->  chunk 2*wm + (j >> 1), octet j & 1, half hi
                const int o4 = (2 * wm + (j >> 1)) * XTCH + (j & 1) * XT + col;
                xt2[(o4 << 1) + hi] = fh.u;
                xt2[((o4 + 2 * XT) << 1) + hi] = fl.u;
            }
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
    __syncthreads();

    // ---------------- phase 2: conv2 over the xt tile ----------------
    {
        const int rd2 = hi * XT + colw;
#if AMP_ABL & 1024
        static_assert(NCH % 2 == 0, "chunk pairs");
        Frag bhs[2][NI], bl[NI];
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            bhs[0][t].u = xt4[rd2 + 32 * t];
            bl[t].u = xt4[rd2 + 2 * XT + 32 * t];
            bhs[1][t].u = xt4[rd2 + (KT > 1 ? 1 : XTCH) + 32 * t];
        }
        for (int c = 0; c < NCH; c += 2) {
            static_for<2 * KT>([&](auto N) {
                constexpr int n = decltype(N)::value;
                constexpr int co = n / KT, g = n % KT, cur = n & 1;
                const int cc = c + co;
                const uint4* wan = wa2 + (size_t)(cc + 1) * (KT * 128);   // last: next mb block / pad
                const uint4* base = xt4 + cc * XTCH + rd2;
                const uint4* basen = cc + 1 < NCH ? base + XTCH : base;   // the last chunk re-reads itself
                const uint4* b1 = g + 1 < KT ? base + (g + 1) : basen + (g + 1 - KT);
                const uint4* b2 = g + 2 < KT ? base + (g + 2) : basen + (g + 2 - KT);
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t] = MF2(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t) bl[t].u = b1[2 * XT + 32 * t];
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t] = MF2(a_h[g].h, bhs[cur][t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t] = MF2(a_l[g].h, bhs[cur][t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t) bhs[cur][t].u = b2[32 * t];
                a_h[g].u = wan[g * 128];
                a_l[g].u = wan[g * 128 + 64];
                __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            });
        }
    }
#elif AMP_ABL & 512
        Frag bh[NI], bl[NI];
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            bh[t].u = xt4[rd2 + 32 * t];
            bl[t].u = xt4[rd2 + 2 * XT + 32 * t];
        }
        for (int c = 0; c < NCH; ++c) {
            const uint4* wan = wa2 + (size_t)(c + 1) * (KT * 128);   // last: next mb block / pad
            const uint4* base = xt4 + c * XTCH + rd2;
            const uint4* basen = c + 1 < NCH ? base + XTCH : base;   // next chunk (the last one re-reads itself)
#pragma unroll
            for (int g = 0; g < KT; ++g) {
                const uint4* bn = g + 1 < KT ? base + (g + 1) : basen;
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t] = MF2(a_h[g].h, bh[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t] = MF2(a_l[g].h, bh[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t) bh[t].u = bn[32 * t];
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t] = MF2(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t) bl[t].u = bn[2 * XT + 32 * t];
                a_h[g].u = wan[g * 128];
                a_l[g].u = wan[g * 128 + 64];
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
        }
    }
#else
        for (int c = 0; c < NCH; ++c) {
            const uint4* wan = wa2 + (size_t)(c + 1) * (KT * 128);   // last: next mb block / pad
            const uint4* base = xt4 + c * XTCH + rd2;
#pragma unroll
            for (int g = 0; g < KT; ++g) {
                const uint4* bg = base + ((AMP_ABL & 128) ? 0 : g);
                Frag bh[NI], bl[NI];
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    bh[t].u = bg[32 * t];
                    bl[t].u = bg[2 * XT + 32 * t];
                }
#pragma unroll
                for (int t = 0; t < NI; ++t)
                    acc[t] = MF2(a_h[g].h, bh[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t)
                    acc[t] = MF2(a_h[g].h, bl[t].h, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NI; ++t)
                    acc[t] = MF2(a_l[g].h, bh[t].h, acc[t], 0, 0, 0);
#if !(AMP_ABL & 4)
                a_h[g].u = wan[g * 128];
                a_l[g].u = wan[g * 128 + 64];
#endif
                AMP_PIN_VMEM();
            }
        }
    }

#endif
    // ---------------- epilogue: + residual, MRF accumulate, store ----------------
    // loads are unconditional from clamped addresses (batched, one wait), stores are predicated
    {
        const float i2 = a.isc2;
        const int mode = a.mode;
        float* yr = a.y + (size_t)item * C * T + (size_t)(32 * wm + 4 * hi) * T;
        if (AMP_ABL & 16) {
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[t][r] = 1.f;
        }
        if (!RES_EARLY && !(AMP_ABL & 16)) {
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[t][r] = xres[(size_t)((r & 3) + 8 * (r >> 2)) * T + qc[t]];
        }
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


}  // namespace amp

int main(int argc, char** argv) {
    using namespace amp;
    constexpr int KT = AMP_KT, WM = 4, WN = 1, NI = 3, SX = 192;
    const int B = 64, C = 128, T = 16384;
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    constexpr int N1 = 32 * NI * WN, XT = N1 + 12, NT = N1 - (KT - 1);
    const size_t n = (size_t)B * C * T, wbytes = (size_t)C * C * KT * 4 + 65536;
    float *x, *y, *b1, *b2; void *w1, *w2;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&b1, C * 4); hipMalloc(&b2, C * 4); hipMalloc(&w1, wbytes); hipMalloc(&w2, wbytes);
    {
        std::vector<float> h(n); srand(1);
        for (auto& v : h) v = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * 3.f;
        hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<_Float16> w(wbytes / 2);
        for (size_t i = 0; i < w.size(); ++i) { float r = (float)rand() / (float)RAND_MAX * 2.f - 1.f; w[i] = (_Float16)(((i / 512) & 1) ? r * 2.f : r * 4096.f); }
        hipMemcpy(w1, w.data(), wbytes, hipMemcpyHostToDevice); hipMemcpy(w2, w.data(), wbytes, hipMemcpyHostToDevice);
        std::vector<float> bb(C, 0.01f); hipMemcpy(b1, bb.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(b2, bb.data(), C * 4, hipMemcpyHostToDevice);
    }
    const size_t lds = ((size_t)2 * 4 * SX + (size_t)2 * WM * 4 * XT) * sizeof(uint4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_f16x3_kernel<KT, WM, WN, NI, SX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dil : {1, 5}) {
        PairArgs a{};
        a.x = x; a.y = y; a.wp1 = w1; a.wp2 = w2; a.bias1 = b1; a.bias2 = b2; a.B = B; a.C = C; a.T = T;
        a.tiles_per_item = (T + NT - 1) / NT; a.dil = dil; a.slope = 0.1f; a.sc1 = 16.f * 4096.f; a.isc1 = 1.f / a.sc1 / 128.f; a.sc2 = a.sc1; a.isc2 = a.isc1;
        a.mode = 0; a.div = 1.f; a.lens = nullptr; a.len_mul = 1;
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < reps + 1; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((pair_f16x3_kernel<KT, WM, WN, NI, SX>), dim3(B * a.tiles_per_item), dim3(256), lds, 0, a);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("abl %4d  k %d dil %d : mean %.3f ms  min %.3f ms  (%s)\n", AMP_ABL, KT, dil, sum / reps, best, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
