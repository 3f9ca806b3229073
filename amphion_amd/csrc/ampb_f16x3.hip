// A whole AMPBlock1 of BigVGAN in one launch on the gfx950 f16 matrix cores (split-f16 operands, see conv_f16x3.hip) -- round 4:
//
//     for p in 0 .. n-1:   x = x + c2_p( a_{2p+1}( c1_p( a_{2p}(x) ) ) )      [last pair: + the running MRF sum, / num_kernels]
//
// i.e. AMPBlock1.forward (bigvgan.py:137-146) with a_i = Activation1d(Snake | SnakeBeta) = 12-tap 2x up-sampling FIR -> Snake
// -> 12-tap 2x down-sampling FIR, replicate-padded at the utterance's own ends (modules/anti_aliasing/act.py:31-36,
// resample.py:36-65, filter.py:92-99, modules/activation_functions/snake.py:51-61).  Unfused, the block is 6 conv launches
// (conv_f16x3.hip) and 6 act1d launches (small_kernels.hip): 27 passes of the [B, C, T] tensor through HBM with the convs'
// matrix pipe 9-49 % busy (profiles/r3_c3_kernel_stats.csv).  Here the tensor is read once and written once.
//
// What makes the activation fusable is the TRANSPOSED product.  rb_f16x3.hip (HiFi-GAN's whole ResBlock) keeps the wave's tile
// in the MFMA C layout "lane = column, registers = channels"; a FIR along time would cross lanes there.  This kernel issues
// every MFMA with its operands exchanged,  D^T[time, channel] = X^T[time, k] * W^T[k, channel]  -- the same fragments from the
// same LDS tile and the same packed weights (the A and B operand layouts of v_mfma_f32_32x32x16_f16 are mirror images), the
// same products summed in the same order, so the same bits -- and gets "lane = channel, registers = time":
//
//   A rows     row i = 8 a + 4 b + j of the MFMA tile t reads tile column 64 b + 16 t + 4 a + j (a lane picks its own LDS address,
//              so the row -> column map is free; this one keeps ds_read_b128's four 16-lane groups on 16 distinct bank quads)
//   P layout   hence acc[t][r] of lane (m = lane & 31, h = lane >> 5) = channel 32 * wm + m at wave column 64 h + 16 t + r: a lane
//              owns 64 CONSECUTIVE columns of its channel, in register order.  (The first version computed the plain C layout and
//              exchanged half-waves with 64 v_permlane32_swap per conv; permuting the rows costs nothing.)
//
// In the P layout Activation1d is register arithmetic: act1d_kernel's operation sequence (act1d_math.h) on a run of 64 columns,
// in place, with a sliding window of Snake pairs; only the 5 columns either side of a run come from elsewhere -- the other half
// wave (5 more swaps) or the neighbouring wave (a 1.25-KB LDS exchange per wave).  No LDS traffic, no bank conflicts, no
// redundant Snake evaluations beyond the 5 + 3 per 64 of the run's ends.  The activation's output then has to become the next
// conv's operand tile [16-channel chunk][plane hi|lo][octet][column][8 x f16] (rb_f16x3.hip's layout): the LDS does that transposition,
// every value is written to its own 2-byte slot (see the step loop; the first version's DPP lane transposes are in
// profiles/negative_kernels/ampb_transpose_write.hip.txt).
//
//   step s (6 per block):  [x or xt in registers, P layout] -> halo exchange -> Activation1d a_s -> x16, hi / lo -> LDS tile
//                          -> conv s (K loop over the tile, all waves) -> un-scale (+ residual: odd steps) -> P layout
//   tile      all C channels x W = 128 * WN columns of one item, every op evaluated on all of them; the `rh` columns either side
//             (the block's receptive field: 5 per activation + (k-1)/2 * d per conv) are never stored.
//   edges     the reference pads x AND the Snake output by replication at the utterance's ends; the conv pads with zeros.  Columns
//             outside [0, Tv) are zeroed when the tile is written; the 5 outputs next to either end are recomputed from scratch
//             (act_edge_value: act1d's edge semantics, the same operation order) by two lanes per channel from an fp32 copy of the
//             tile and patched into the operand tile -- in the (at most two) tiles per utterance that contain an end.
//
// Per output element the operation order of every conv (accumulator start (bias + residual + MRF sum) * scale, chunks, taps,
// the three MFMAs of a term, un-scale, divide) is conv_f16x3.hip's and that of every activation is act1d_kernel's, so the block is
// bit-identical to the 12 launches it replaces (tests/test_gpu_ampblock.py).
//
// Compiled once per tap count:  -DAMP_KT=<3|5|7|11>.
#include "act1d_math.h"
#include "amp_internal.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif
#include <type_traits>

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union FragQ {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)

// P layout: run column c (0 .. 63) of a lane lives in v[AMP_PT(c)][AMP_PR(c)]
#define AMP_PT(c) ((c) >> 4)
#define AMP_PR(c) ((c) & 15)

// v_permlane32_swap: lanes 32-63 of a swap with lanes 0-31 of b.  Issued through inline asm on scalar copies, for two reasons seen on
// hipcc 7.2: (i) __builtin_bit_cast applied directly to an element of an ext_vector_type lvalue reads element 0 and rewrites the whole
// vector; (ii) with the builtin on sub-registers of the 512-bit accumulator tuples the round trip C -> P -> C came back permuted
// (first hardware run: a lane's b = 1 column groups held its b = 0 groups), while the same builtin on plain scalars is correct
// (tests/experiments/ampb_primitives.hip).  The s_nop covers the "VALU write -> v_permlane read" hazard (2 wait states).
__device__ __forceinline__ void lane_half_swap(float a, float b, float& na, float& nb) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    na = a;
    nb = b;
}

template <typename T>
__device__ __forceinline__ T opaque(T v) {   // a value the optimiser cannot hoist out of the step loop (and then spill)
    asm volatile("" : "+v"(v));
    return v;
}

// Activation1d on a lane's run, in place: v (P layout) holds x[0 .. 63] of one channel, hl = x[-5 .. -1], hr = x[64 .. 68].
//   Snake pair P[Q] = (s[2Q - 5], s[2Q - 4]),  s[n] = u + invb * sin(a u)^2,  u = sum_k x[Q - k] * 2 f_up[2k (+1)]   (k = 0 .. 5)
//   y[t] = sum_m f_dn[2m] * P[t + m].x  +  sum_m f_dn[2m + 1] * P[t + m].y                                            (m = 0 .. 5)
// -- act1d_kernel's chains (small_kernels.hip), term for term.  Groups of four pairs; after group g the outputs 4g - 5 .. 4g - 2 are
// complete and overwrite inputs that no later pair reads (pair Q reads x[Q - 5 .. Q]).
__device__ __forceinline__ void act_run(f32x16 (&v)[4], const float (&hl)[5], const float (&hr)[5], const float a, const float invb,
                                        const float (&fu2)[12], const float (&fd)[12]) {
    f32x2 P[72];
#pragma unroll
    for (int g = 0; g < 18; ++g) {
        f32x2 uv[4], xa[4], sv[4];
        // the four chains step by step (dependent packed FMAs back to back cost a wait state each)
#pragma unroll
        for (int q = 0; q < 4; ++q) uv[q] = pk_splat(0.f);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * g + q - k;
                const float xv = c < 0 ? hl[c + 5 < 0 ? 0 : c + 5] : (c > 63 ? hr[c - 64 > 4 ? 4 : c - 64] : v[AMP_PT(c & 63)][AMP_PR(c & 63)]);
                uv[q] = pk_fma(pk_splat(xv), (f32x2){fu2[2 * k], fu2[2 * k + 1]}, uv[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = uv[q] * a;
        snake_sin2_pk4(xa, sv);
#pragma unroll
        for (int q = 0; q < 4; ++q) P[4 * g + q] = pk_fma(pk_splat(invb), sv[q], uv[q]);
        {
            f32x2 oa[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) oa[e] = pk_splat(0.f);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = 4 * g - 5 + e;
                    if (t >= 0 && t < 64) oa[e] = pk_fma((f32x2){fd[2 * m], fd[2 * m + 1]}, P[(t + m) < 72 ? (t + m) : 71], oa[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = 4 * g - 5 + e;
                if (t >= 0 && t < 64) v[AMP_PT(t & 63)][AMP_PR(t & 63)] = oa[e].x + oa[e].y;
            }
        }
        // One group at a time.  The 18 groups are independent until their outputs, and instruction selection linearises the block
        // with all of them interleaved (a sched_barrier only fences the machine scheduler: 200 spilled registers).  An empty asm that
        // "rewrites" the next group's first inputs together with this group's outputs is a true data dependence: group g + 1 cannot
        // start before group g's outputs exist.
        if (g >= 1 && g < 17) {
            float t0 = v[AMP_PT((4 * g + 4) & 63)][AMP_PR((4 * g + 4) & 63)], t1 = v[AMP_PT((4 * g + 5) & 63)][AMP_PR((4 * g + 5) & 63)];
            float t2 = v[AMP_PT((4 * g + 6) & 63)][AMP_PR((4 * g + 6) & 63)], t3 = v[AMP_PT((4 * g + 7) & 63)][AMP_PR((4 * g + 7) & 63)];
            float o0 = v[AMP_PT((4 * g - 5) & 63)][AMP_PR((4 * g - 5) & 63)], o1 = v[AMP_PT((4 * g - 4) & 63)][AMP_PR((4 * g - 4) & 63)];
            float o2 = v[AMP_PT((4 * g - 3) & 63)][AMP_PR((4 * g - 3) & 63)], o3 = v[AMP_PT((4 * g - 2) & 63)][AMP_PR((4 * g - 2) & 63)];
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
            v[AMP_PT((4 * g + 4) & 63)][AMP_PR((4 * g + 4) & 63)] = t0; v[AMP_PT((4 * g + 5) & 63)][AMP_PR((4 * g + 5) & 63)] = t1;
            v[AMP_PT((4 * g + 6) & 63)][AMP_PR((4 * g + 6) & 63)] = t2; v[AMP_PT((4 * g + 7) & 63)][AMP_PR((4 * g + 7) & 63)] = t3;
            v[AMP_PT((4 * g - 5) & 63)][AMP_PR((4 * g - 5) & 63)] = o0; v[AMP_PT((4 * g - 4) & 63)][AMP_PR((4 * g - 4) & 63)] = o1;
            v[AMP_PT((4 * g - 3) & 63)][AMP_PR((4 * g - 3) & 63)] = o2; v[AMP_PT((4 * g - 2) & 63)][AMP_PR((4 * g - 2) & 63)] = o3;
        }
    }
}

// One output of Activation1d next to an utterance end, from scratch: y[t] with x read from the fp32 row `frow` of the tile
// (tile column = global column - q0), x clamped to [0, Tv - 1] and Snake values to [0, 2 Tv - 1] (resample.py:36-45,
// filter.py:92-99) -- the values and the operation order of act1d_kernel's edge tiles.  Rolled loops, taps from memory: this runs in
// two lanes per channel of at most two tiles per utterance and must not cost the common path registers.
__device__ __forceinline__ float act_edge_value(const float* frow, const int q0, const int W, const int Tv, const int t, const float a,
                                               const float invb, const float* __restrict__ fu2p, const float* __restrict__ fdp) {
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 1
    for (int j = 0; j < 12; ++j) {            // tap j = 2 m + half of the down filter
        int n = 2 * t - 5 + j;
        n = n < 0 ? 0 : (n > 2 * Tv - 1 ? 2 * Tv - 1 : n);
        const int Q = (n + 5) >> 1;
        const int odd = (n + 5) & 1;
        float u = 0.f;
#pragma unroll 1
        for (int k = 0; k < 6; ++k) {
            int q = Q - k;
            q = q < 0 ? 0 : (q > Tv - 1 ? Tv - 1 : q);
            int col = q - q0;
            col = col < 0 ? 0 : (col > W - 1 ? W - 1 : col);
            u = fmaf(frow[col], fu2p[2 * k + odd], u);
        }
        const float s = fmaf(invb, snake_sin2(u * a), u);
        const float f = fdp[j];
        if (j & 1) acc1 = fmaf(f, s, acc1);
        else acc0 = fmaf(f, s, acc0);
    }
    return acc0 + acc1;
}

// RING = D > 0: the weight fragments as a ring of D taps (rb_f16x3.hip / conv_blk_f16x3.hip)
template <int KT, int WM, int WN, int RING, int G>
__global__ __launch_bounds__(64 * WM * WN, 2) void ampb_f16x3_kernel(const AmpbArgs a) {
    static_assert(RING == 0 || RING < KT, "a ring shorter than one chunk");
    constexpr int NTHR = 64 * WM * WN;
    constexpr int W = 128 * WN;               // columns per tile (every op is evaluated on all of them)
    constexpr int WL = W + 2 * G + 1;         // LDS row length: guards either side + 1 pad column (WL = 1 mod 8: see the scatter below)
    constexpr int H2 = (KT - 1) / 2;
    constexpr int NCH = 2 * WM;               // 16-channel chunks (C = 32 * WM)
    constexpr int CHS = 4 * WL;               // uint4 per chunk [plane][octet][WL]
    constexpr int C = 32 * WM;
    constexpr int FS = W + 4;                 // row stride of the fp32 copy of the tile (edge tiles; aliases the operand tile)
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];   // [NCH][plane][octet][WL] | exchange [wave][side][32][5] | fix [C][2][5]
    float* const xch = reinterpret_cast<float*>(smem4 + NCH * CHS);
    float* const fixb = xch + WM * WN * 320;
    float* const F = reinterpret_cast<float*>(smem4);
    static_assert(C * FS * 4 <= NCH * CHS * 16, "the fp32 copy must fit the operand tile");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, m = lane & 31;
    const int nbx = gridDim.x;  // XCD-contiguous tile runs, see conv_f16x3.hip (ragged batches keep the dispatch order)
    int bx = ((nbx & 7) == 0 && !a.lens) ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (a.rev) bx = nbx - 1 - bx;
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int T = a.T;
    const int RH = a.rh;
    const int NT = W - 2 * RH;                // output columns per tile
    const int O0 = tile * NT;                 // first output column
    int Tv = T;                               // valid columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    if (O0 >= Tv) return;                     // tile beyond the utterance: unspecified by contract (workgroup-uniform)
    const int q0 = O0 - RH;                   // global column of tile column 0 (a multiple of 4)
    const int qw = q0 + 128 * wn + 64 * h;    // global column of this lane's run
    const int ch = 32 * wm + m;               // this lane's channel
    // a tile that contains an end of the utterance: zero padding, replicate padding (workgroup-uniform)
    const bool has_lo = q0 <= 0, has_hi = Tv - 1 < q0 + W;
    const bool edge = has_lo || has_hi;

    auto zero_guards = [&]() {
        for (int i = tid; i < NCH * 4 * 2 * G; i += NTHR) {
            const int row = i / (2 * G), g = i - row * (2 * G);
            smem4[row * WL + (g < G ? g : W + g)] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    zero_guards();

    // a tensor row in the P layout: 16 aligned float4 per lane (T % 4 == 0, q0 % 4 == 0: a group lies inside [0, T) or outside)
    const size_t rowoff = ((size_t)item * C + ch) * T;
    // `lim`: columns from there on read as zero.  x: the utterance's own length -- beyond it a ragged row holds whatever the workspace
    // held (possibly NaN), and although those columns are zeroed again when the operand tile is written (a factor 0), NaN * 0 would
    // not be; the running MRF sum: the row length (its columns only ever meet the same columns of the output)
    auto load_rows = [&](const float* base, const int qrun, const int lim, f32x16 (&dst)[4]) __attribute__((always_inline)) {
        const float* row = base + rowoff;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qg = qrun + 4 * i;
            const bool inb = qg >= 0 && qg < T;                  // whole float4 inside the row or outside (T, q0 multiples of 4)
            const float4 f = *reinterpret_cast<const float4*>(row + (inb ? qg : 0));
            dst[AMP_PT(4 * i)][AMP_PR(4 * i) + 0] = (inb && qg + 0 < lim) ? f.x : 0.f;
            dst[AMP_PT(4 * i)][AMP_PR(4 * i) + 1] = (inb && qg + 1 < lim) ? f.y : 0.f;
            dst[AMP_PT(4 * i)][AMP_PR(4 * i) + 2] = (inb && qg + 2 < lim) ? f.z : 0.f;
            dst[AMP_PT(4 * i)][AMP_PR(4 * i) + 3] = (inb && qg + 3 < lim) ? f.w : 0.f;
        }
    };

    f32x16 xv[4];                             // x (then x + pair_0(x), ...): the residual, P layout
    f32x16 acc[4];                            // accumulators / the activation's operand and result (P layout)
    load_rows(a.x, qw, Tv, xv);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = xv[t];

    // weight fragments [mb][chunk][tap][plane][lane] x uint4 (conv_build); as the B operand of the transposed product a lane holds
    // exactly what it holds as the A operand of the plain one
    constexpr size_t MBS = (size_t)NCH * (KT * 128);
    constexpr int NA = RING > 0 ? RING : KT;
    FragQ w_h[NA], w_l[NA];

    float range_max = 0.f;       // largest |staged operand| (x16 applied): beyond 65504 it left the f16 range (a.range_flag)
    const int e8 = lane & 7, o4 = (lane >> 3) & 3;
    const bool b4 = (lane & 4) != 0, b2 = (lane & 2) != 0, b1 = (lane & 1) != 0;

    const int ns = a.ns;
#pragma unroll 1
    for (int s = 0; s < AMP_AMPB_MAX_STEPS; ++s) {
        if (s >= ns) break;                       // workgroup-uniform
        const bool odd = (s & 1) != 0;
        const bool last = s + 1 == ns;
        // ---------------- halo exchange: 5 columns either side of every run ----------------
        float hl[5], hr[5];
        {
            float* my = xch + ((wave * 2 + h) * 32 + m) * 5;      // side 0: my first five (h = 0), side 1: my last five (h = 1)
#pragma unroll
            for (int i = 0; i < 5; ++i) my[i] = h ? acc[AMP_PT(59 + i)][AMP_PR(59 + i)] : acc[AMP_PT(i)][AMP_PR(i)];
        }
        __syncthreads();                          // ... and every wave is through the previous conv: the tile may be overwritten
        {
            const int wl = wn > 0 ? wave - 1 : wave, wr = wn + 1 < WN ? wave + 1 : wave;   // (tile edges: finite filler)
            const float* nb = h ? xch + ((wr * 2 + 0) * 32 + m) * 5 : xch + ((wl * 2 + 1) * 32 + m) * 5;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                float from_lo, from_hi;
                lane_half_swap(acc[AMP_PT(i)][AMP_PR(i)], acc[AMP_PT(59 + i)][AMP_PR(59 + i)], from_lo, from_hi);
                // from_lo in lanes h = 1: the last five of lane (m, 0); from_hi in lanes h = 0: the first five of lane (m, 1)
                const float other = nb[i];
                hl[i] = h ? from_lo : other;
                hr[i] = h ? other : from_hi;
            }
        }
        // wave-uniform filter taps (scalar loads -> SGPR operands); the up taps arrive doubled (UpSample1d's gain, resample.py:41:
        // act1d_kernel folds the same exact x2 into them)
        float fu2[12], fd[12];
        {
            const float* fup = a.act_fu[s];
            const float* fdp = a.act_fd[s];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                fu2[k] = fup[k];
                fd[k] = fdp[k];
            }
        }
        const float aa = a.act_a[s][ch], invb = a.act_invb[s][ch];

        if (edge) {
            // fp32 copy of the tile (over the operand tile, which nobody reads now), then the outputs next to the utterance's ends
            // from it: lane (m, 0) of the wn = 0 waves the low end of channel m, lane (m, 1) the high end
            float* frow = F + (size_t)ch * FS + 128 * wn + 64 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                *reinterpret_cast<float4*>(frow + 4 * i) = make_float4(acc[AMP_PT(4 * i)][AMP_PR(4 * i)], acc[AMP_PT(4 * i)][AMP_PR(4 * i) + 1],
                                                                       acc[AMP_PT(4 * i)][AMP_PR(4 * i) + 2], acc[AMP_PT(4 * i)][AMP_PR(4 * i) + 3]);
            __syncthreads();
            if (wn == 0 && (h ? has_hi : has_lo)) {
                const int tb = h ? (Tv - 5 > 5 ? Tv - 5 : 5) : 0;          // low end: t = 0 .. 4, high end: the last five beyond those
                const int te = h ? Tv : (Tv < 5 ? Tv : 5);
                const float* fr = F + (size_t)ch * FS;
#pragma unroll 1
                for (int e = 0; e < 5; ++e) {
                    const int t = tb + e;
                    if (t < te) fixb[(ch * 2 + h) * 5 + e] = act_edge_value(fr, q0, W, Tv, t, aa, invb, a.act_fu[s], a.act_fd[s]);
                }
            }
            __syncthreads();
            zero_guards();                        // the copy ran over them
        }

        // ---------------- Activation1d, in place ----------------
        act_run(acc, hl, hr, aa, invb, fu2, fd);

        // the conv's first weight fragments: in flight under the transposes
        {
            const uint4* w0 = static_cast<const uint4*>(a.wp[s]) + (size_t)wm * MBS + lane;
#pragma unroll
            for (int g = 0; g < NA; ++g) {
                w_h[g].u = w0[g * 128];
                w_l[g].u = w0[g * 128 + 64];
            }
        }
        AMP_PIN_VMEM();

        // ---------------- x16, hi / lo, the conv's zero padding -> the operand tile ----------------
        // A lane holds ONE channel of 64 columns; the tile wants 8 channels of one column per 16-B unit.  The transposition is left to
        // the LDS: every value goes to its own 2-byte slot (ds_write_b16 / _d16_hi of the packed conversions, 128 per step).  The kernel is
        // VALU-bound with the LDS pipe ~10 % busy, and the register transposes of the first version (profiles/negative_kernels/ampb_transpose_write.hip.txt) were 384 of its ~2 500 vector
        // instructions per step.  Rows of WL * 16 B with WL = 1 (mod 8): the four octets of a half-wave (two per chunk, chunks 4 WL apart) land on disjoint banks.
        {
            _Float16* const dst = reinterpret_cast<_Float16*>(smem4 + (2 * wm + (o4 >> 1)) * CHS + (o4 & 1) * WL + G + 128 * wn + 64 * h) + e8;
            auto scatter = [&](auto masked) __attribute__((always_inline)) {
                const int qws = opaque(qw);
#pragma unroll
                for (int c = 0; c < 64; c += 4) {
                    float k[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) k[j] = (!decltype(masked)::value || (qws + c + j >= 0 && qws + c + j < Tv)) ? 16.f : 0.f;
                    const amp_f32x2 v01 = {acc[AMP_PT(c)][AMP_PR(c)] * k[0], acc[AMP_PT(c + 1)][AMP_PR(c + 1)] * k[1]};
                    const amp_f32x2 v23 = {acc[AMP_PT(c + 2)][AMP_PR(c + 2)] * k[2], acc[AMP_PT(c + 3)][AMP_PR(c + 3)] * k[3]};
                    range_max = __builtin_fmaxf(range_max, __builtin_fmaxf(__builtin_fabsf(v01.x), __builtin_fabsf(v01.y)));
                    range_max = __builtin_fmaxf(range_max, __builtin_fmaxf(__builtin_fabsf(v23.x), __builtin_fabsf(v23.y)));
                    uint2 hh, ll;
                    split4_f16(v01, v23, hh, ll);
                    const amp_f16x2 h01 = __builtin_bit_cast(amp_f16x2, hh.x), h23 = __builtin_bit_cast(amp_f16x2, hh.y);
                    const amp_f16x2 l01 = __builtin_bit_cast(amp_f16x2, ll.x), l23 = __builtin_bit_cast(amp_f16x2, ll.y);
                    dst[8 * (c + 0)] = h01.x; dst[8 * (c + 1)] = h01.y; dst[8 * (c + 2)] = h23.x; dst[8 * (c + 3)] = h23.y;
                    dst[8 * (c + 0 + 2 * WL)] = l01.x; dst[8 * (c + 1 + 2 * WL)] = l01.y; dst[8 * (c + 2 + 2 * WL)] = l23.x; dst[8 * (c + 3 + 2 * WL)] = l23.y;
                }
            };
            if (edge) scatter(std::true_type{});      // (workgroup-uniform: the selects only where an utterance ends)
            else scatter(std::false_type{});
        }
        if (edge) {
            __syncthreads();                      // the patched columns belong to other lanes' writes
            if (wn == 0 && (h ? has_hi : has_lo)) {
                const int tb = h ? (Tv - 5 > 5 ? Tv - 5 : 5) : 0;
                const int te = h ? Tv : (Tv < 5 ? Tv : 5);
                _Float16* const t16 = reinterpret_cast<_Float16*>(smem4);
                const size_t rowoff = ((size_t)((ch >> 4) * CHS + ((ch >> 3) & 1) * WL + G) << 3) + (ch & 7);
#pragma unroll 1
                for (int e = 0; e < 5; ++e) {
                    const int t = tb + e;
                    const int col = t - q0;
                    if (t < te && col >= 0 && col < W) {
                        const float v = fixb[(ch * 2 + h) * 5 + e] * 16.f;
                        range_max = __builtin_fmaxf(range_max, __builtin_fabsf(v));
                        _Float16 vh, vl;
                        split_f16(v, vh, vl);
                        t16[rowoff + ((size_t)col << 3)] = vh;
                        t16[rowoff + ((size_t)(col + 2 * WL) << 3)] = vl;
                    }
                }
            }
        }
        __syncthreads();

        // ---------------- conv s over the tile (transposed product): acc = init + sum over chunks, taps ----------------
        {
            const float bv = a.bias[s][ch];
            const float sc = a.sc[s];
            if (!odd) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = bv * sc;
            } else {
                // (bias + residual [+ running MRF sum]) * scale, conv_f16x3.hip's accumulator start
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = xv[t] + bv;
                if (last && a.mode != 0) {
                    // the running MRF sum, 16 columns (four float4) at a time with ALL FOUR loads issued before the first is consumed.
                    // (Round 5: as `load_rows` into a temporary f32x16[4] hipcc -- at 250 registers -- fused every load with its add and
                    // put s_waitcnt vmcnt(0) behind each: sixteen serial round trips per tile, 0.09-0.17 ms of the k = 7 / 11 launches with
                    // every other phase knocked out.)  Columns beyond the row read as zero; same sums in the same order.
                    const float* yrow = a.y + rowoff;
                    const int qrun = opaque(qw);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float4 f[4];
                        bool inb[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int qg = qrun + 16 * t + 4 * i;
                            inb[i] = qg >= 0 && qg < T;
                            f[i] = *reinterpret_cast<const float4*>(yrow + (inb[i] ? qg : 0));
                        }
                        __builtin_amdgcn_sched_barrier(0);      // nothing crosses: four loads in flight, then their uses
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[t][4 * i + 0] += inb[i] ? f[i].x : 0.f;
                            acc[t][4 * i + 1] += inb[i] ? f[i].y : 0.f;
                            acc[t][4 * i + 2] += inb[i] ? f[i].z : 0.f;
                            acc[t][4 * i + 3] += inb[i] ? f[i].w : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] *= sc;
            }
            const int d = a.dil[s];
            const uint4* wa = static_cast<const uint4*>(a.wp[s]) + (size_t)wm * MBS + lane;
            // A row m = 8 a + 4 b + j of tile t <- tile column 64 b + 16 t + 4 a + j (see the header): output register r of tile t is then
            // run column 16 t + r of this lane
            const int rd = h * WL + G + 128 * wn + 64 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3) - H2 * d;
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                const uint4* wcur = wa + (size_t)c * (KT * 128);
                const uint4* wan = (c + 1) < NCH ? wa + (size_t)(c + 1) * (KT * 128) : wa;   // (last chunk: a reload nobody uses)
                const uint4* base = smem4 + c * CHS + rd;
#pragma unroll
                for (int g = 0; g < KT; ++g) {
                    const int v = RING > 0 ? g % NA : g;
                    const uint4* bg = base + g * d;
#pragma unroll
                    for (int th = 0; th < 2; ++th) {      // the tap's x fragments in two halves (16 registers instead of 32)
                        FragQ xh[2], xl[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            xh[t].u = bg[16 * (th * 2 + t)];
                            xl[t].u = bg[2 * WL + 16 * (th * 2 + t)];
                        }
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[th * 2 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[t].h, w_h[v].h, acc[th * 2 + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[th * 2 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[t].h, w_h[v].h, acc[th * 2 + t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[th * 2 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[t].h, w_l[v].h, acc[th * 2 + t], 0, 0, 0);
                    }
                    {   // this tap's register set is free: fetch the tap it serves next
                        const bool same = RING > 0 && g + RING < KT;
                        const uint4* src = same ? wcur + (size_t)(g + RING) * 128 : wan + (size_t)v * 128;
                        w_h[v].u = src[0];
                        w_l[v].u = src[64];
                    }
                    AMP_PIN_VMEM();
                }
            }
            // un-scale (conv_f16x3.hip's epilogue)
            const float isc = a.isc[s];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] *= isc;
            if (odd && last && a.mode == 2) {
                const float dv = a.div;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = acc[t][r] / dv;
            }
            if (odd) {
#pragma unroll
                for (int t = 0; t < 4; ++t) xv[t] = acc[t];
            }
        }
    }

    // ---------------- store the NT output columns of the tile ----------------
    {
        float* row = a.y + rowoff;
        const int colr = opaque(128 * wn + 64 * h);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = colr + 4 * i;
            const int q = q0 + col;
            if (col >= RH && col < W - RH && q < T)
                *reinterpret_cast<float4*>(row + q) = make_float4(xv[AMP_PT(4 * i)][AMP_PR(4 * i)], xv[AMP_PT(4 * i)][AMP_PR(4 * i) + 1],
                                                                  xv[AMP_PT(4 * i)][AMP_PR(4 * i) + 2], xv[AMP_PT(4 * i)][AMP_PR(4 * i) + 3]);
        }
    }
    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);
}

template <int KT, int WM, int WN, int RING, int G>
static hipError_t launch_ampb_one(const AmpbArgs& a, hipStream_t stream) {
    constexpr int W = 128 * WN;
    const size_t lds = (size_t)2 * WM * 4 * (W + 2 * G + 1) * sizeof(uint4) + (size_t)WM * WN * 320 * sizeof(float) + (size_t)32 * WM * 10 * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds<&ampb_f16x3_kernel<KT, WM, WN, RING, G>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.tiles_per_item));
    note_kernel("ampb_f16x3_kernel", KT, WM, WN, RING, G);
    note_work(grid.x, a.ns * 2.0 * a.C * a.C * KT * (double)a.T * a.B / 1e9, 4.0 * a.B * (double)a.C * a.T * (2 + (a.mode ? 1 : 0)) / 1e6,
              "whole AMPBlock C=%d k=%d T=%d B=%d: %d convs + %d Activation1d%s", a.C, KT, a.T, a.B, a.ns, a.ns, a.mode ? " +sum" : "");
    hipLaunchKernelGGL((ampb_f16x3_kernel<KT, WM, WN, RING, G>), grid, dim3(64 * WM * WN), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// Tile width W (columns evaluated per workgroup; outputs per tile = W - 2 * rh) for C channels in form `wide`
// (1: eight waves, one workgroup per CU; 0: four waves, two per CU), or 0 when not covered.
int AMP_CAT(ampb_tile_kt, AMP_KT)(int C, int max_dil, int wide) {
    constexpr int KT = AMP_KT;
    const int reach = (KT - 1) / 2 * max_dil;
    if (reach > 32) return 0;
    if (C == 32) return wide ? 1024 : 512;
    if (C == 64) return wide ? 512 : 0;
    return 0;
}

hipError_t AMP_CAT(launch_ampb_kt, AMP_KT)(const AmpbArgs& a, int wide, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    constexpr int RING = KT >= 7 ? 4 : (KT == 5 ? 3 : 0);   // (k = 5: a ring of three taps instead of five resident ones leaves every form without spills)
    if (a.C == 32) return wide ? launch_ampb_one<KT, 1, 8, RING, 32>(a, stream) : launch_ampb_one<KT, 1, 4, RING, 32>(a, stream);
    if (a.C == 64 && wide) return launch_ampb_one<KT, 2, 4, RING, 32>(a, stream);
    return hipErrorInvalidValue;
}

}  // namespace amp
