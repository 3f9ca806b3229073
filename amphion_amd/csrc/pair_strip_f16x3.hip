// Fused ResBlock pair, strip-mined along time (round 2; successor of the per-tile kernel in pair_f16x3.hip):
//
//     y = x + c2( lrelu( c1( lrelu(x) ) ) )         [+ the running MRF sum, / num_kernels]
//
// (ONE iteration of ResBlock1.forward, hifigan.py:93-100).  The per-tile kernel gave every workgroup N1 xt
// columns and kept NT = N1 - (k - 1) outputs: both convs paid the k - 1 seam columns on every tile (11.6 % of
// the MFMAs at k = 11, N1 = 96) and every tile re-staged its full dilated halo.  Here a workgroup walks a
// STRIP of one utterance in steps of N1 columns and carries conv2's halo in LDS:
//
//   step s    conv1 (kernel KT, dilation d) on the N1 xt columns [X, X + N1), K loop over 16-channel chunks of x
//             staged through the double-buffered LDS tile (as in conv_f16x3.hip);
//   seam      the last KT - 1 xt columns of the previous step move to the front of the xt tile (each lane moves
//             the values it wrote itself), then bias + leaky_relu + conv2's zero padding + x16 + hi/lo split of
//             the new columns -> xt tile [chunk][plane hi|lo][octet][KT - 1 old | N1 new][8 x f16];
//   conv2     (kernel KT, dilation 1) on the N1 outputs [X - H2, X + N1 - H2): K loop over the xt chunks in
//             LDS, no staging, no barriers; its last chunk fetches the next step's first A fragments;
//   epilogue  the staging loads of the NEXT step's first chunk are issued, then + bias, + residual x, MRF
//             accumulate, store.
//
// Only the first step of a strip computes KT - 1 columns it throws away.  Per output element the order of
// operations (bias, chunks, taps, the three MFMAs of a term) is that of the per-tile kernel and of
// conv_f16x3.hip, so the result has the same bits however the time axis is cut (tests/test_gpu_pair.py).
//
// Round 5 (clock stamps of every wave, profiles/r5_j_pair_strip_stamps.txt): with one wave per SIMD nothing hides an LDS round trip, and
// the B fragments of a half-tap were read 1-2 MFMAs before their first use -- the matrix pipe waited ~100 cycles 22 times per chunk.  They
// are now read one half-tap (12 MFMAs) ahead into a second register set (468 of 512 registers, no spill) and the next chunk's
// staging loads go out behind the first half-tap instead of in front of it: the MFMA loops issue at the rate of a register-resident
// MFMA loop (35.8 cycles per MFMA), a wave's life fell from 203.5 k to 190.1 k cycles -- and the launch by 1.8 %: the package sits at
// its power cap and gave the rest back as clock (1 809 -> 1 746 MHz).  Same bits (only load timing moved).
//
// WM x WN waves x MI row blocks per wave: C = 32 * WM * MI channels (C = 256 runs 8 waves, one workgroup per CU: the xt tile
// of all 256 channels is 108 KB); compiled once per tap count:  -DAMP_KT=<3|5|7|11>.
#include "amp_internal.h"

#include <stdlib.h>
#include <string.h>

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union FragS {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)
// LDS reads stay above, MFMAs below (VALU / SALU / VMEM / LDS writes may cross)
#define AMP_PIN_DSREAD() __builtin_amdgcn_sched_barrier(0x276)


// RING = D > 0 (round 2, third session; `wide` = 3): the A fragments as a ring of D taps per row block instead of a whole chunk's set
// (conv_blk_f16x3.hip: tap g in slot g % D, re-loaded right after its use with tap g + D of this chunk or tap g % D of the next
// one, across conv1 -> conv2 -> the next step's conv1) -- 16 * D registers instead of 16 * KT for two row blocks, which is what
// leaves room for NI = 4 column tiles per wave (64 rows x 128 columns: A-fragment traffic per MFMA -25 %, seam waste (k - 1) / 256
// instead of / 192); the B fragments of a tap are read in two halves.  SBUF = 1: ONE x staging buffer (the 136-KB xt tile of a
// 256-column step leaves 20 KB of the 160: the chunk loop waits for every wave's reads before the next chunk is written -- one
// more barrier per chunk).  Same bits as every other form; measured 1.86-1.90 ms against 2.23 for the per-tile kernel in the
// same process at k = 11 (the whole-chunk 2 x 2-blocked strips: 0.957 of the per-tile kernel), 1.27 against 1.43 at k = 7.
template <int KT, int WM, int WN, int NI, int SX, int MI = 1, int RING = 0, int SBUF = 2>
__global__ __launch_bounds__(64 * WM * WN, (MI > 1 ? 1 : 2)) void pair_strip_kernel(const PairArgs a) {
    static_assert(RING == 0 || RING <= KT, "ring of at most one chunk");
    static_assert(SBUF == 1 || SBUF == 2, "staging buffers");
    constexpr int NTHR = 64 * WM * WN;
    constexpr int N1 = 32 * NI * WN;          // xt columns per step = output columns per step
    constexpr int H2 = (KT - 1) / 2;
    constexpr int HB = KT - 1;                // xt columns carried from the previous step
    constexpr int XT = N1 + HB;               // xt row length
    constexpr int NCH = 2 * WM * MI;          // 16-channel chunks (C = 32 * WM * MI: a wave owns MI 32-row blocks)
    constexpr int XBUF = 4 * SX;              // uint4 per x staging buffer [plane][octet][SX]
    constexpr int XTCH = 4 * XT;              // uint4 per xt chunk       [plane][octet][XT]
    constexpr int NST = (4 * SX) / NTHR;      // staging items (column x channel quad) per thread
    // the tap's B fragments in BH halves (ring form with an even NI: 16 registers instead of 32); every accumulator still sees hh, hl, lh
    // of the tap in this order
    constexpr int BH = (RING > 0 && NI % 2 == 0) ? 2 : 1;
    constexpr int NB = NI / BH;
    static_assert(RING > 0 && BH == 2, "the one form left: A-fragment ring, B fragments in two halves, read one half-tap ahead");
    static_assert(SX % 64 == 0, "staging items must have a wave-uniform channel quad");
    static_assert((4 * SX) % NTHR == 0, "staging items must divide over the threads");
    static_assert(HB <= 32, "the carried columns must belong to the last n-tile");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [SBUF][XBUF] + [NCH][XTCH]
    uint4* const xt4 = smem4 + SBUF * XBUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5, l31 = lane & 31;
    const int nbx = gridDim.x;  // XCD-contiguous runs of strips, see conv_f16x3.hip
    // (ragged batches keep the dispatch order: with utterances of different lengths a contiguous run per XCD would hand
    // one XCD the long utterances and another only tiles that exit at once -- measured 43.8 vs 48.7 ms padded, visit AD)
    int bx = ((nbx & 7) == 0 && !a.lens) ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (a.rev) bx = nbx - 1 - bx;   // descending tile order: start where the previous launch stopped writing (PairArgs::rev)
    const int item = bx / a.strips_per_item;
    const int strip = bx - item * a.strips_per_item;
    constexpr int C = 32 * WM * MI;
    const int T = a.T;
    const int O = strip * a.strip_len;        // first output column of the strip
    int Tv = T;                               // valid columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    // the strip ends at the utterance's valid length: columns beyond it are unspecified by contract and nothing
    // downstream reads them (a batch of 60..400-frame utterances is 40 % such columns); the steps before keep their positions
    const int Oend = O + a.strip_len < Tv ? O + a.strip_len : Tv;
    if (O >= Oend) return;                    // workgroup-uniform
    const int nsteps = (Oend - O + HB + N1 - 1) / N1;
    const int dil = a.dil;
    const int h1 = H2 * dil;

    const float* xb = a.x + (size_t)item * C * T;
    const float kpos = 16.f, kneg = 16.f * a.slope;

    // Residual x at this lane's OUTPUT positions: for C <= 64 (HBM-bound pairs) fetched at the start of a step,
    // next to the staging loads of the same cache lines (pair_f16x3.hip); C >= 128 re-reads in the epilogue.
    constexpr bool RES_EARLY = WM * MI < 4;
    const int colw = wn * (32 * NI) + l31;    // this lane's column inside the step (n-tile 0)
    // this lane's rows of row block mi: channels 32 * (MI * wm + mi) + 16 * hi + r (PERMUTED A rows, round 4: A row (lane & 31) =
    // 8 a + 4 b + j fetches the packed fragment of weight row 16 b + 4 a + j, so a lane's sixteen accumulators are one 16-channel
    // chunk = two whole octets and the seam is two conflict-free ds_write_b128 per plane instead of four ds_write_b64 at a 16-B lane
    // stride -- rb_f16x3.hip's header has the measurement; the same dot products in other rows of the tile, the same bits)
    const float* xres = a.x + (size_t)item * C * T + (size_t)(32 * MI * wm + 16 * hi) * T;
    float* yr = a.y + (size_t)item * C * T + (size_t)(32 * MI * wm + 16 * hi) * T;

    // Step-local, opaque copies of T and the row bases (set at the top of every step): with the plain values hipcc
    // hoists every row offset r * T and the 48 residual / output addresses out of the step loop and holds them
    // across both convs (all 106 SGPRs + 256 VGPRs + spills); re-deriving them per step costs a few SALU ops.
    // The laundered pointers are cast back to the GLOBAL address space: through a generic pointer of unknown origin hipcc
    // emits flat_load / flat_store (1 056 + 480 in this file), which also count in lgkmcnt and return out of order with
    // LDS -- the next step's first ds_read then waits for every epilogue store of this one.
    typedef const float __attribute__((address_space(1)))* gcf_ptr;
    typedef float __attribute__((address_space(1)))* gf_ptr;
    int Ts = T;
    const float* xres_l = xres;
    float* yr_l = yr;
    gcf_ptr xres_s = (gcf_ptr)xres;
    gf_ptr yr_s = (gf_ptr)yr;
    float range_max = 0.f;       // largest |staged operand| (x16 applied): beyond 65504 it left the f16 range (a.range_flag)
    float xs[NST][4];
    auto stage_load = [&](int chunk, int tbase) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int ibase = wave * 64 + NTHR * it;         // wave-uniform
            const int qd = ibase / SX;                       // channel quad 0..3
            const int col = ibase - qd * SX + lane;
            int t = tbase + col;
            t = t < 0 ? 0 : t;
            t = t > Ts - 1 ? Ts - 1 : t;
            const int ch0 = chunk * KC16 + 4 * qd;
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[it][e] = xb[(size_t)(ch0 + e) * Ts + t];
        }
    };
    auto stage_store = [&](int buf, int tbase) {
        uint2* dst = reinterpret_cast<uint2*>(smem4 + buf * XBUF);
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int ibase = wave * 64 + NTHR * it;
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

    // A fragments [mb][chunk][tap][plane][lane] x uint4, one register set, reloaded one chunk ahead; the reload
    // during conv1's last chunk fetches conv2's first chunk, the one during conv2's last chunk the next step's.
    constexpr size_t MBS = (size_t)NCH * (KT * 128);   // uint4 per 32-row block of packed A fragments
    const int wlane = (lane & 32) | (16 * ((lane >> 2) & 1) + 4 * ((lane >> 3) & 3) + (lane & 3));   // the permuted fragment (above)
    const uint4* wa1 = static_cast<const uint4*>(a.wp1) + (size_t)(MI * wm) * MBS + wlane;
    const uint4* wa2 = static_cast<const uint4*>(a.wp2) + (size_t)(MI * wm) * MBS + wlane;
    constexpr int NA = RING > 0 ? RING : KT;          // A-fragment register sets per row block
    FragS a_h[MI][NA], a_l[MI][NA];
    // after tap g of the chunk at `wcur`: the register set of tap g is re-loaded with the tap it serves next -- the same tap of the
    // next chunk (`wnext`) without a ring; with a ring tap g + RING of this chunk, or tap g % RING of the next chunk
    auto reload = [&](int mi, int g, const uint4* wcur, const uint4* wnext) __attribute__((always_inline)) {
        const int v = RING > 0 ? g % NA : g;
        const bool same = RING > 0 && g + RING < KT;
        const uint4* src = same ? wcur + (size_t)(g + RING) * 128 : wnext + (size_t)(RING > 0 ? g % NA : g) * 128;
        a_h[mi][v].u = src[mi * MBS];
        a_l[mi][v].u = src[mi * MBS + 64];
    };
    const int rd1 = hi * SX + colw;
    const int rd2 = hi * XT + colw;

    int X = O - H2;                           // first xt column of the step
    stage_load(0, X - h1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < NA; ++g) {
            a_h[mi][g].u = wa1[mi * MBS + g * 128];
            a_l[mi][g].u = wa1[mi * MBS + g * 128 + 64];
        }
    AMP_PIN_VMEM();

#pragma clang loop unroll(disable)
    for (int s = 0; s < nsteps; ++s, X += N1) {
        // hipcc would hoist the step-invariant bias values and row addresses out of this loop (60+ registers held
        // across both convs -> spills); the opaque pointers make it re-derive them where they are used
        const float* bias1 = a.bias1;
        const float* bias2 = a.bias2;
        asm volatile("" : "+s"(bias1), "+s"(bias2));
        Ts = T; xres_l = xres; yr_l = yr;
        asm volatile("" : "+s"(Ts), "+v"(xres_l), "+v"(yr_l));
        xres_s = (gcf_ptr)xres_l; yr_s = (gf_ptr)yr_l;
        const int tbase = X - h1;             // global column of staged column 0
        // output columns of this step: q = X - H2 + col
        int qc[NI];
        bool okc[NI];
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int q = X - H2 + colw + 32 * t;
            okc[t] = (q >= O) && (q < Oend);
            qc[t] = q < 0 ? 0 : (q < Ts ? q : Ts - 1);
        }
        f32x16 rv[MI][NI];
        if (RES_EARLY) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int t = 0; t < NI; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[mi][t][r] = xres_s[(size_t)(32 * mi + r) * Ts + qc[t]];
        }

        // ---------------- conv1 ----------------
        f32x16 acc[MI][NI];
        {
            const float s1 = a.sc1;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias1[32 * (MI * wm + mi) + 16 * hi + r] * s1;
#pragma unroll
                    for (int t = 0; t < NI; ++t) acc[mi][t][r] = bv;
                }
        }
        stage_store(0, tbase);
        __syncthreads();

#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            const bool more = (c + 1) < NCH;
            const uint4* wcur = wa1 + (size_t)c * (KT * 128);
            const uint4* wan = more ? wa1 + (size_t)(c + 1) * (KT * 128) : wa2;
            const uint4* base = smem4 + (SBUF == 2 ? (c & 1) : 0) * XBUF + rd1;
            {
                // the B fragments of half-tap h + 1 are read while the MFMAs of half-tap h run (two register sets); the staging loads of the
                // next chunk are issued behind the first half-tap's MFMAs instead of in front of them (round 5: clock stamps showed the
                // matrix pipe waiting ~100 cycles for LDS at every half-tap and ~700 for the address arithmetic at every chunk)
                FragS bh[2][NB], bl[2][NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    bh[0][t].u = base[32 * t];
                    bl[0][t].u = base[2 * SX + 32 * t];
                }
#pragma unroll
                for (int h = 0; h < KT * BH; ++h) {
                    const int g = h / BH, th = h % BH, cur = h & 1;
                    const int v = g % NA;
                    if (h + 1 < KT * BH) {
                        const uint4* bn = base + ((h + 1) / BH) * dil;
                        const int tn = (h + 1) % BH;
#pragma unroll
                        for (int t = 0; t < NB; ++t) {
                            bh[cur ^ 1][t].u = bn[32 * (tn * NB + t)];
                            bl[cur ^ 1][t].u = bn[2 * SX + 32 * (tn * NB + t)];
                        }
                    }
                    AMP_PIN_DSREAD();
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v].h, bh[cur][t].h, acc[mi][th * NB + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v].h, bl[cur][t].h, acc[mi][th * NB + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[mi][v].h, bh[cur][t].h, acc[mi][th * NB + t], 0, 0, 0);
                        if (th == BH - 1) reload(mi, g, wcur, wan);
                    }
                    if (h == 0) {
                        AMP_PIN_VMEM();
                        stage_load(more ? c + 1 : c, tbase);
                    }
                    if (th == BH - 1 || h == 0) AMP_PIN_VMEM();
                }
            }
            if (SBUF == 1) __syncthreads();   // every wave has read the one staging buffer
            if (more) stage_store(SBUF == 2 ? (c + 1) & 1 : 0, tbase);
            __syncthreads();
        }

        // ---------------- seam: carry the halo, xt = lrelu(conv1) -> LDS, split-f16 B layout ----------------
        {
            // position of xt column `col` of this step = HB + col.  The lane that wrote column col >= N1 - HB in the
            // previous step moves it from HB + col to HB + col - N1 before overwriting it (own data, in order).
            if (s > 0 && wn == WN - 1) {
                const int col = colw + 32 * (NI - 1);
                if (col >= N1 - HB) {
#pragma unroll
                    for (int jm = 0; jm < 2 * MI; ++jm) {
                        const int mi = jm >> 1, o = jm & 1;
                        const int o4 = (2 * (MI * wm + mi) + hi) * XTCH + o * XT + HB + col;
                        const uint4 vh = xt4[o4];
                        const uint4 vl = xt4[o4 + 2 * XT];
                        xt4[o4 - N1] = vh;
                        xt4[o4 - N1 + 2 * XT] = vl;
                    }
                }
            }
            const float i1 = a.isc1;
            const float slope = a.slope;
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const int col = colw + 32 * t;                   // xt column of this step
                const int q = X + col;                           // its global column
                const bool qok = (q >= 0) && (q < Tv);           // conv2 zero-pads xt outside the utterance
#pragma unroll
                for (int jm = 0; jm < 2 * MI; ++jm) {
                    const int mi = jm >> 1, o = jm & 1;
                    struct { uint2 u; } fh0, fl0, fh1, fl1;
                    seam4_f16(acc[mi][t][8 * o + 0], acc[mi][t][8 * o + 1], acc[mi][t][8 * o + 2], acc[mi][t][8 * o + 3], i1, slope, qok, range_max, fh0.u, fl0.u);
                    seam4_f16(acc[mi][t][8 * o + 4], acc[mi][t][8 * o + 5], acc[mi][t][8 * o + 6], acc[mi][t][8 * o + 7], i1, slope, qok, range_max, fh1.u, fl1.u);
                    // channels 32*(MI*wm + mi) + 16*hi + 8*o + i  ->  chunk 2*(MI*wm + mi) + hi, octet o: a whole 16-B unit
                    const int o4 = (2 * (MI * wm + mi) + hi) * XTCH + o * XT + HB + col;
                    xt4[o4] = make_uint4(fh0.u.x, fh0.u.y, fh1.u.x, fh1.u.y);
                    xt4[o4 + 2 * XT] = make_uint4(fl0.u.x, fl0.u.y, fl1.u.x, fl1.u.y);
                }
            }
        }
        {
            const float s2 = a.sc2;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias2[32 * (MI * wm + mi) + 16 * hi + r] * s2;
#pragma unroll
                    for (int t = 0; t < NI; ++t) acc[mi][t][r] = bv;
                }
        }
        __syncthreads();

        // ---------------- conv2 over the xt tile ----------------
        {
            // as in conv1; there is no barrier between the chunks, so the last half-tap of a chunk reads the first of the next one
            FragS bh[2][NB], bl[2][NB];
            {
                const uint4* b0 = xt4 + rd2;
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    bh[0][t].u = b0[32 * t];
                    bl[0][t].u = b0[2 * XT + 32 * t];
                }
            }
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                const uint4* wcur = wa2 + (size_t)c * (KT * 128);
                const uint4* wan = (c + 1) < NCH ? wa2 + (size_t)(c + 1) * (KT * 128) : wa1;
                const uint4* base = xt4 + c * XTCH + rd2;
                const uint4* basen = xt4 + ((c + 1) < NCH ? c + 1 : c) * XTCH + rd2;      // last chunk: a read nobody uses
#pragma unroll
                for (int h = 0; h < KT * BH; ++h) {
                    const int g = h / BH, th = h % BH, cur = h & 1;
                    const int v = g % NA;
                    {
                        const bool wrap = h + 1 == KT * BH;
                        const uint4* bn = wrap ? basen : base + (h + 1) / BH;
                        const int tn = wrap ? 0 : (h + 1) % BH;
#pragma unroll
                        for (int t = 0; t < NB; ++t) {
                            bh[cur ^ 1][t].u = bn[32 * (tn * NB + t)];
                            bl[cur ^ 1][t].u = bn[2 * XT + 32 * (tn * NB + t)];
                        }
                    }
                    AMP_PIN_DSREAD();
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v].h, bh[cur][t].h, acc[mi][th * NB + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v].h, bl[cur][t].h, acc[mi][th * NB + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < NB; ++t)
                            acc[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[mi][v].h, bh[cur][t].h, acc[mi][th * NB + t], 0, 0, 0);
                        if (th == BH - 1) reload(mi, g, wcur, wan);
                    }
                    if (th == BH - 1) AMP_PIN_VMEM();
                }
            }
        }

        // ---------------- epilogue: + residual, MRF accumulate, store ----------------
        // loads are unconditional from clamped addresses (batched, one wait), stores are predicated.  The staging loads
        // of the next step's first chunk go first (their registers are free again; they land under the epilogue).
        stage_load(0, (s + 1) < nsteps ? tbase + N1 : tbase);
        AMP_PIN_VMEM();
        {
            const float i2 = a.isc2;
            const int mode = a.mode;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {     // one 32-row block at a time (48 residual registers, not 48 * MI)
                if (!RES_EARLY) {
#pragma unroll
                    for (int t = 0; t < NI; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) rv[mi][t][r] = xres_s[(size_t)(32 * mi + r) * Ts + qc[t]];
                }
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[mi][t] = acc[mi][t] * i2 + rv[mi][t];
                if (mode != 0) {   // wave-uniform
#pragma unroll
                    for (int t = 0; t < NI; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) rv[mi][t][r] = yr_s[(size_t)(32 * mi + r) * Ts + qc[t]];
#pragma unroll
                    for (int t = 0; t < NI; ++t) acc[mi][t] += rv[mi][t];
                    if (mode == 2) {
#pragma unroll
                        for (int t = 0; t < NI; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[mi][t][r] = acc[mi][t][r] / a.div;
                    }
                }
#pragma unroll
                for (int t = 0; t < NI; ++t)
                    if (okc[t]) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) yr_s[(size_t)(32 * mi + r) * Ts + qc[t]] = acc[mi][t][r];
                    }
            }
        }
    }
    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);
}


template <int KT, int WM, int WN, int NI, int SX, int MI = 1, int RING = 0, int SBUF = 2>
static hipError_t launch_strip_one(const PairArgs& a, hipStream_t stream) {
    constexpr int N1 = 32 * NI * WN;
    constexpr int XT = N1 + (KT - 1);
    const size_t lds = ((size_t)SBUF * 4 * SX + (size_t)2 * WM * MI * 4 * XT) * sizeof(uint4);
    if (hipError_t e = ensure_dynamic_lds<&pair_strip_kernel<KT, WM, WN, NI, SX, MI, RING, SBUF>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.strips_per_item));
    note_kernel("pair_strip_kernel", KT, WM, WN, NI, SX, MI, RING, SBUF);
    note_work(grid.x, 2 * 2.0 * a.C * a.C * KT * (double)a.T * a.B / 1e9, 2 * 4.0 * a.B * (double)a.C * a.T * (1.0 + (a.mode ? 0.5 : 0.0)) / 1e6,
              "fused pair C=%d k=%d d=%d T=%d B=%d%s", a.C, KT, a.dil, a.T, a.B, a.mode ? " +sum" : "");
    hipLaunchKernelGGL((pair_strip_kernel<KT, WM, WN, NI, SX, MI, RING, SBUF>), grid, dim3(64 * WM * WN), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// Step width (xt columns = output columns per step) for C channels, or 0 when (C, KT, dilation) is not covered;
// *wg_per_cu = resident workgroups per CU (LDS / register bound), used by the host to size the strips.
// ONE form is left (round 5): wide = 3, the 2 x 2-blocked strips with an A-fragment ring -- C = 128, k >= 7: 4 waves x (64 rows x 128 columns),
// 256-column steps, one workgroup per CU -- the form the launch policy picks.  The four-wave strips (C = 32 / 64 / 128, two workgroups per CU:
// +1.5 % at C = 128, k = 11, slower elsewhere) and the eight-wave C = 256 strips (overtaken by the row-blocked conv kernel) were reachable
// only through amp_set_pair_strips(1) and are gone, like round 2's double-width tiles, the whole-chunk 2 x 2 form and the C = 64 ring form.
int AMP_CAT(strip_step_kt, AMP_KT)(int C, int dil, int wide, int* wg_per_cu) {
    constexpr int KT = AMP_KT;
    const int span = (KT - 1) * dil;   // staged halo = 2 * h1
    if (wg_per_cu) *wg_per_cu = 1;
    if (wide == 3 && C == 128) return (KT >= 7 && 256 + span <= 320) ? 256 : 0;
    return 0;
}

hipError_t AMP_CAT(launch_strip_kt, AMP_KT)(const PairArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    if constexpr (KT >= 7) {
        if (a.wide == 3 && a.C == 128) return launch_strip_one<KT, 2, 2, 4, 320, 2, 4, 1>(a, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace amp
